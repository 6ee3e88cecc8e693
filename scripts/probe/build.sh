#!/bin/bash
# the stand-alone A/B binary of the sweep schedules: the library's kernels_sweep.hip is compiled INTO it
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DGPX_SWEEP_PROBES -I../../include -o sweep_ab.bin sweep_ab.hip
