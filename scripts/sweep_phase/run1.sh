#!/bin/bash
# first look: baseline, stamps, de-phasing by pause, asymmetric priority
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
./sweep_probe.bin 0 0 5
./sweep_probe.bin 1 0 3 8192 65536 $O/stamps_base.bin
for s in 32 64 128 192 256; do ./sweep_probe.bin 2 $s 5; done
./sweep_probe.bin 4 0 5
./sweep_probe.bin 3 64 3 8192 65536 $O/stamps_deph64.bin
./sweep_probe.bin 5 0 3 8192 65536 $O/stamps_prioa.bin
./sweep_probe.bin 0 0 5
} 2>&1 | grep -v "^  launch" | tee $O/run1.log
