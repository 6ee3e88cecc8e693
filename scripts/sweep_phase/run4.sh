#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
./sweep_probe.bin 0 0 3
./sweep_probe.bin 1 0 2 8192 65536 $O/stamps4w_base.bin
} 2>&1 | grep -v "^  launch" | tee $O/run4.log
