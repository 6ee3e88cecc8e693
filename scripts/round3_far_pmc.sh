#!/bin/bash
# what the far trailing updates wait for: SQ counters of k_syrk_update_tri (N = 16384, far only) next to the sweep kernel's
export GPX_ROUND=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
C3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
bash $R/scripts/pmc_cmd.sh far1 "$C1" scripts/chol_one.py 16384 x_skip=6
bash $R/scripts/pmc_cmd.sh far2 "$C2" scripts/chol_one.py 16384 x_skip=6
bash $R/scripts/pmc_cmd.sh far3 "$C3" scripts/chol_one.py 16384 x_skip=6
bash $R/scripts/pmc_cmd.sh swp1 "$C1" scripts/pmc_sweep.py 23
bash $R/scripts/pmc_cmd.sh swp2 "$C2" scripts/pmc_sweep.py 23
bash $R/scripts/pmc_cmd.sh swp3 "$C3" scripts/pmc_sweep.py 23
for t in far1 far2 far3; do python $R/scripts/pmc_parse.py $R/gpurun_out/r03/pmc_$t/${t}_counter_collection.csv | grep -i "syrk_update_tri"; done
for t in swp1 swp2 swp3; do python $R/scripts/pmc_parse.py $R/gpurun_out/r03/pmc_$t/${t}_counter_collection.csv | grep -i "sweep_trmm"; done
