#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
for m in 0 4235776 6332928 8430080 4203008 0; do ./sweep_probe.bin $m 0 5; done
} 2>&1 | grep -v "^  launch\|^mode" | tee $O/run10.log
