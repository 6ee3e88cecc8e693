#!/bin/bash
# Build libgpx.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun snapshots).
#   build.sh          compile the objects that are older than their sources (incremental: development)
#   build.sh --force  compile EVERY object from source (what __graft_entry__.build() runs: the driver's build check must
#                     compile all eight translation units, also on a snapshot that shipped up-to-date .o files)
# A/B builds (never shipped): EXTRA="-DGPX_TG_REGLOOPS" compiles the factorisation's workers with the register-staged k-loops of rounds
# 2-5 instead of the LDS-DMA ones; EXTRA="-DGPX_PF16_COUNTED_WAIT" lets the diagonal role publish a step with its next tile's stores in flight
# (round 5's hand-counted s_waitcnt; 2-5 % faster up to N = 4096, see ADVICE round 5 for why it is off).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-const-variable $EXTRA"
SRCS="kernels_fit kernels_chol_tg kernels_sweep kernels_rff kernels_grad kernels_ens comm api"
FORCE=0
[ "$1" = "--force" ] && FORCE=1
pids=()
n=0
for f in $SRCS; do
  stale=$FORCE
  if [ ! -f $f.o ]; then stale=1; fi
  for dep in $f.hip gemm_core.h gpx_internal.h gpx_diag.h gpx_math.h fit_tiles.h ../../include/gpx.h; do
    [ $dep -nt $f.o ] && stale=1
  done
  if [ $stale = 1 ]; then
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    pids+=($!)
    n=$((n + 1))
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS $f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgpx.so $OBJS -ldl
# the diagnostics build: the same objects, api.hip compiled with -DGPX_DIAGNOSTICS (gpx_set_option accepts the diagnostic knobs of
# gpx_diag.h).  The test-suite drives this one (tests/conftest.py: GPX_DIAGNOSTICS=1); everything else loads libgpx.so.
stale=$FORCE
[ ! -f api_diag.o ] && stale=1
for dep in api.hip gpx_internal.h gpx_diag.h ../../include/gpx.h; do [ $dep -nt api_diag.o ] && stale=1; done
[ $stale = 1 ] && $HIPCC $FLAGS -DGPX_DIAGNOSTICS -c api.hip -o api_diag.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgpx_diag.so ${OBJS/api.o/api_diag.o} -ldl
echo "built $(pwd)/libgpx.so ($n of $(echo $SRCS | wc -w) objects compiled) and libgpx_diag.so"
