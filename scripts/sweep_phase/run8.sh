#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
for m in 0 512 8192 32768 40960 65536 73728 0; do ./sweep_probe.bin $m 0 5; done
./sweep_probe.bin 40961 0 2 8192 65536 $O/stamps4w_bufload_band.bin
} 2>&1 | grep -v "^  launch\|^mode\|checksum" | tee $O/run8.log
