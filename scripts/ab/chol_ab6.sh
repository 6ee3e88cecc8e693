#!/bin/bash
# round 6: the factorisation with the workers' k-loops by LDS-DMA (libgpx.so) against the same dispatcher with the register-staged
# loops (libgpx_ab_reg.so: -DGPX_TG_REGLOOPS) and against round 5's kernels_chol_tg.hip (libgpx_ab_r5.so); same box, same process order
cd "$(dirname "$0")/../.."
for n in 2048 4096 8192 12288 14336 16384; do
  for lib in libgpx_ab_r5.so libgpx_ab_reg.so libgpx.so; do
    [ -f pybo_amd/csrc/$lib ] || continue
    echo -n "$lib  "; GPX_LIB_PATH=$PWD/pybo_amd/csrc/$lib timeout 300 python scripts/tg/tg_sweep.py $n chol_tg=1
  done
done
