#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
./sweep_probe.bin 0 0 3
./sweep_probe.bin 48 0 3
./sweep_probe.bin 256 0 3
./sweep_probe.bin 304 0 3
./sweep_probe.bin 257 0 2 8192 65536 $O/stamps12_1wg.bin
./sweep_probe.bin 305 0 2 8192 65536 $O/stamps12_1wg_nostage.bin
} 2>&1 | grep -v "^  launch" | tee $O/run3.log
