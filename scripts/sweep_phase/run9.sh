#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
for m in 0 40960 172032 303104 434176 696320 827392 958464 1089536 41472 43008 0; do ./sweep_probe.bin $m 0 5; done
} 2>&1 | grep -v "^  launch\|^mode\|checksum" | tee $O/run9.log
