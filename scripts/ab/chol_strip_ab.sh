#!/bin/bash
# the stripped task-graph kernel (round 5) against the round-4 file with all its options, same box: Cholesky ms at five sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for n in 2048 4096 8192 12288 16384; do
  for lib in scripts/ab/libgpx_prev.so pybo_amd/csrc/libgpx.so; do
    echo -n "$lib  "
    GPX_LIB_PATH=$R/$lib timeout 300 python scripts/tg/tg_sweep.py $n chol_tg=1
  done
done
