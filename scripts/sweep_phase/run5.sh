#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
for m in 0 512 1024 1536 2048 4096 0; do ./sweep_probe.bin $m 0 5; done
./sweep_probe.bin 513 0 2 8192 65536 $O/stamps4w_prio2.bin
./sweep_probe.bin 1025 0 2 8192 65536 $O/stamps4w_lateload.bin
} 2>&1 | grep -v "^  launch" | tee $O/run5.log
