#!/bin/bash
# A/B of two builds of libgpx.so on the fit stages: Cholesky (task graph / stream schedule) and triangular inverse
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for n in 2048 4096 8192 12288 16384; do
  for lib in scripts/ab/libgpx_prev.so pybo_amd/csrc/libgpx.so; do
    echo "== $lib"
    GPX_LIB_PATH=$R/$lib timeout 300 python scripts/tg/tg_sweep.py $n chol_tg=1 chol_tg=0
    GPX_LIB_PATH=$R/$lib timeout 300 python scripts/ab/trtri_time.py $n
  done
done
