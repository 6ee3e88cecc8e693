#!/bin/bash
# The shadows of the diagonal factorisation (chol_tg_shadow=1: seven workgroups that follow role C 16 rows at a time) and the fused
# links of the column chains (chol_tg_fuse=1) against the task version (two solve halves + six update pieces on a critical list,
# every solve and every final chunk a worker task), same box, same build.  HISTORICAL: it ran on commit b6970cf (+ the early
# publication of the fused links), before the task version and the option chol_tg_shadow were removed; its log is
# profiles/r05_chol_shadow_ab.txt.  On later builds the chol_tg_shadow settings are rejected (unknown option).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for n in 512 1024 1536 2048 3001 4096 5000 8192; do
  timeout 300 python scripts/tg/tg_sweep.py $n chol_tg=0 "chol_tg_shadow=0,chol_tg_min=2" "chol_tg_shadow=1,chol_tg_fuse=0,chol_tg_min=2" "chol_tg_shadow=1,chol_tg_fuse=1,chol_tg_min=2" "chol_tg_shadow=0,chol_tg_min=2" "chol_tg_shadow=1,chol_tg_fuse=1,chol_tg_min=2"
done
for n in 12288 14336 16384; do
  timeout 300 python scripts/tg/tg_sweep.py $n chol_tg_shadow=0 "chol_tg_shadow=1,chol_tg_fuse=0" "chol_tg_shadow=1,chol_tg_fuse=1"
done
echo
echo "== the critical path, N = 2048, task version"
timeout 100 python scripts/tg/tg_trace.py 2048 chol_tg_shadow=0 2>&1 | head -22
echo
echo "== the critical path, N = 2048, shadows + fused links"
timeout 100 python scripts/tg/tg_trace.py 2048 2>&1 | head -24
