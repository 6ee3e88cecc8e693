#!/bin/bash
# MFMA-busy counters of one 65536-column sweep launch at N = 8192 for the round-4 (tile_order 23) and round-5 (27) k-loops
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/pmc_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 23 27; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/v$v -o p -- python $R/scripts/pmc_sweep.py $v < /dev/null > $O/v$v.log 2>&1
  f=$(find $O/v$v -name "*counter_collection.csv" | head -1)
  echo "## tile_order $v"; [ -n "$f" ] && python $R/scripts/pmc_parse.py $f | grep "k_sweep_trmm" ; [ -n "$f" ] && grep "k_sweep_trmm" $f > $O/r05_pmc_sq_v$v.csv
done 2>&1 | tee $O/summary.txt
rm -rf $O/v23 $O/v27
