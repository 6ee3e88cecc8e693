#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/pmc_rff; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/v${v}_$tag -o p -- python $R/scripts/rff_probe/rff_one.py 32 64 100 20 $v < /dev/null > $O/v${v}_$tag.log 2>&1
    f=$(find $O/v${v}_$tag -name "*counter_collection.csv" | head -1)
    echo "## x_rff=$v"; [ -n "$f" ] && python $R/scripts/pmc_parse.py $f | grep "k_rff_mfma"
    rm -rf $O/v${v}_$tag
  done
done 2>&1 | tee $O/summary.txt
