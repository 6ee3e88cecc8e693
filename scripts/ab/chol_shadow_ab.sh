#!/bin/bash
# role S as one shadow workgroup (chol_tg_shadow=1) against the task version (two solve halves + six update pieces), same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for n in 2048 4096 8192; do
  timeout 300 python scripts/tg/tg_sweep.py $n chol_tg_shadow=0 chol_tg_shadow=1 chol_tg_shadow=0 chol_tg_shadow=1
done
for n in 5000 12288 16384; do
  timeout 300 python scripts/tg/tg_sweep.py $n chol_tg_shadow=0 chol_tg_shadow=1
done
