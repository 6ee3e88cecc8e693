#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
./sweep_probe.bin 0 0 5
./sweep_probe.bin 1 0 3 8192 65536 $O/stamps12_base.bin
./sweep_probe.bin 16 0 5
./sweep_probe.bin 32 0 5
./sweep_probe.bin 48 0 5
./sweep_probe.bin 64 0 5
./sweep_probe.bin 128 0 5
./sweep_probe.bin 17 0 3 8192 65536 $O/stamps12_nowrite.bin
./sweep_probe.bin 65 0 3 8192 65536 $O/stamps12_paced.bin
./sweep_probe.bin 0 0 5
} 2>&1 | grep -v "^  launch" | tee $O/run2.log
