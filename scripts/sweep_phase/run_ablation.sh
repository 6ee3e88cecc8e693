#!/bin/bash
# ablation matrix of the sweep k-loop's ingredients + the stamp dumps behind profiles/r05_sweep_idle_attribution.txt
cd $(dirname $0)
O=../../gpurun_out/r05/sweep_phase; mkdir -p $O
{
echo "# round-4 kernel (flat loads, priority 1 in the matrix phase)"; ./sweep_probe.bin 0 0 5
echo "# + priority 2 while the loads are issued"; ./sweep_probe.bin 512 0 5
echo "# buffer loads"; ./sweep_probe.bin 8192 0 5
echo "# buffer loads + priority 2 at the load issue"; ./sweep_probe.bin 8704 0 5
echo "# banded priority (1 -> 2 after four MFMA groups)"; ./sweep_probe.bin 32768 0 5
echo "# buffer loads + banded priority"; ./sweep_probe.bin 40960 0 5
echo "# buffer loads + banded priority + priority 2 at the load issue"; ./sweep_probe.bin 41472 0 5
echo "# second fragment set"; ./sweep_probe.bin 4194304 0 5
echo "# buffer loads + second fragment set"; ./sweep_probe.bin 4202496 0 5
echo "# buffer loads + second fragment set + priority 2 at the load issue"; ./sweep_probe.bin 4203008 0 5
echo "# ALL (the library kernel, tile_order 27)"; ./sweep_probe.bin 4235776 0 5
echo "# no global loads, no LDS writes (the matrix loop alone, two workgroups per CU)"; ./sweep_probe.bin 48 0 5
echo "# the matrix loop alone, ONE workgroup per CU"; ./sweep_probe.bin 304 0 5
echo "# round-4 kernel (flat loads, priority 1 in the matrix phase)"; ./sweep_probe.bin 0 0 5
echo "# ALL (the library kernel, tile_order 27)"; ./sweep_probe.bin 4235776 0 5
echo "# stamps: round-4 kernel"; ./sweep_probe.bin 1 0 2 8192 65536 $O/stamps4w_r4.bin
echo "# stamps: library kernel"; ./sweep_probe.bin 4235777 0 2 8192 65536 $O/stamps4w_r5.bin
} 2>&1 | grep -v "^  launch\|^mode" | tee $O/ablation.log
