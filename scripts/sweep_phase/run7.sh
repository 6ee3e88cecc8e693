#!/bin/bash
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
./sweep_probe.bin 0 0 5
./sweep_probe.bin 8192 0 5
for s in 32 64 96 128 160 192 256; do ./sweep_probe.bin 8194 $s 5; done
for s in 64 128; do ./sweep_probe.bin 12290 $s 5; done
./sweep_probe.bin 8195 128 2 8192 65536 $O/stamps4w_bufload_deph.bin
./sweep_probe.bin 0 0 5
} 2>&1 | grep -v "^  launch\|^mode\|checksum" | tee $O/run7.log
