#!/bin/bash
# Build libgpx.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun snapshots).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-const-variable"
pids=()
for f in kernels_fit kernels_chol_tg kernels_sweep kernels_rff kernels_grad kernels_ens comm api; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ gemm_core.h -nt $f.o ] || [ gpx_internal.h -nt $f.o ] || [ gpx_math.h -nt $f.o ] || [ fit_tiles.h -nt $f.o ] || [ ../../include/gpx.h -nt $f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgpx.so kernels_fit.o kernels_chol_tg.o kernels_sweep.o kernels_rff.o kernels_grad.o kernels_ens.o comm.o api.o -ldl
echo "built $(pwd)/libgpx.so"
