#!/bin/bash
# the double-buffered workers of the task-graph Cholesky (chol_tg_db) against the single-buffer ones, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for n in 2048 4096 8192; do
  timeout 300 python scripts/tg/tg_sweep.py $n chol_tg_db=0 chol_tg_db=1 chol_tg_db=0 chol_tg_db=1
done
for n in 12288 16384; do
  timeout 300 python scripts/tg/tg_sweep.py $n chol_tg_db=0 chol_tg_db=1
done
