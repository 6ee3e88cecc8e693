#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
for m in 0 512 8192 8704 10240 12288 8192 0; do ./sweep_probe.bin $m 0 5; done
./sweep_probe.bin 8193 0 2 8192 65536 $O/stamps4w_bufload.bin
} 2>&1 | grep -v "^  launch" | tee $O/run6.log
