#!/bin/bash
# counter passes over k_rff_mfma: what bounds it?
export GPX_ROUND=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/scripts/pmc_cmd.sh rff1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" scripts/pmc_rff.py
bash $R/scripts/pmc_cmd.sh rff2 "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" scripts/pmc_rff.py
for t in rff1 rff2; do f=$(ls $R/gpurun_out/r03/pmc_$t/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/scripts/pmc_parse.py $f | grep -i "rff_mfma"; done > $R/gpurun_out/r03/pmc_rff_summary.txt
cat $R/gpurun_out/r03/pmc_rff_summary.txt
