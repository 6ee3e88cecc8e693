#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
for m in 0 41472 4235776 0; do ./sweep_probe.bin $m 0 5; done
} 2>&1 | grep -v "^  launch\|^mode\|checksum" | tee $O/run10.log
