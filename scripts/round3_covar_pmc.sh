#!/bin/bash
# counter passes over the covariance-evaluation kernels (k_cross_gram, k_sweep_rankq<1>): what bounds them?
export GPX_ROUND=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/scripts/pmc_cmd.sh cov1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" scripts/pmc_covar.py
bash $R/scripts/pmc_cmd.sh cov2 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" scripts/pmc_covar.py
bash $R/scripts/pmc_cmd.sh cov3 "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_WAVES SQ_IFETCH" scripts/pmc_covar.py
for t in cov1 cov2 cov3; do f=$(ls $R/gpurun_out/r03/pmc_$t/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/scripts/pmc_parse.py $f | grep -i "rankq\|cross_gram"; done > $R/gpurun_out/r03/pmc_covar_summary.txt
cat $R/gpurun_out/r03/pmc_covar_summary.txt
