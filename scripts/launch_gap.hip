// Per-kernel cost of a chain of DEPENDENT launches in one stream (what a fused chain kernel could save at most):
// an empty kernel, a kernel that reads and writes one cache line per workgroup, with 1 and 64 workgroups.
// build + run (on the GPU box): hipcc --offload-arch=gfx950 -O3 scripts/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_touch(double* p) { if (threadIdx.x == 0) p[blockIdx.x * 16] += 1.0; }
int main() {
    double* d; hipMalloc((void**)&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, s);
            for (int i = 0; i < n; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s);
                else if (mode == 1) hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, s, d);
                else if (mode == 2) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, d);
                else hipLaunchKernelGGL(k_touch, dim3(512), dim3(512), 0, s, d);
            }
            hipEventRecord(b, s); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("mode %d: %.2f us per dependent launch\n", mode, ms * 1e3 / n);
        }
    }
    return 0;
}
