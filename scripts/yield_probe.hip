// Does a workgroup see, through a scalar load with glc, a word that ANOTHER workgroup on the same compute unit raised
// with an agent-scope atomic store -- and how soon?  (The mechanism behind option chol_yield.)
//   hipcc --offload-arch=gfx950 -O2 -I pybo_amd/csrc scripts/yield_probe.hip -o scripts/yield_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../pybo_amd/csrc/gemm_core.h"
using namespace gpx;

__global__ __launch_bounds__(256) void k_far_like(const int* pause, int* seen, long long* first_seen, int iters) {
    __shared__ double pad[72 * 128];
    const int* pw = pause + cu_key();
    int hits = 0;
    long long t_first = 0;
    for (int it = 0; it < iters; ++it) {
        int v = poll_issue(pw);
        __builtin_amdgcn_s_sleep(20);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v));
        if (v) { if (!hits) t_first = wall_clock64(); ++hits; }
    }
    if (threadIdx.x == 0) { seen[blockIdx.x] = hits; first_seen[blockIdx.x] = t_first; pad[0] = hits; }
    if (pad[threadIdx.x & 127] == 1234.5) seen[0] = -1;
}

__global__ __launch_bounds__(256) void k_pf_like(int* pause, long long* stamps, int key_out[1]) {
    int* pw = pause + cu_key();
    if (threadIdx.x == 0) {
        key_out[0] = cu_key();
        stamps[0] = wall_clock64();
        __hip_atomic_store(pw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = 0; i < 60; ++i) __builtin_amdgcn_s_sleep(127);       // ~200 us
    if (threadIdx.x == 0) {
        __hip_atomic_store(pw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamps[1] = wall_clock64();
    }
}

int main() {
    int *pause, *seen, *key;
    long long *first, *stamps;
    hipMalloc(&pause, CU_KEYS * 4); hipMemset(pause, 0, CU_KEYS * 4);
    hipMalloc(&seen, 512 * 4); hipMalloc(&first, 512 * 8); hipMalloc(&stamps, 16); hipMalloc(&key, 4);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipLaunchKernelGGL(k_far_like, dim3(510), dim3(256), 0, s1, pause, seen, first, 20000);   // ~ms, leaves two slots free
    hipLaunchKernelGGL(k_pf_like, dim3(1), dim3(256), 0, s2, pause, stamps, key);
    hipDeviceSynchronize();
    std::vector<int> h(512); std::vector<long long> f(512); long long st[2]; int k;
    hipMemcpy(h.data(), seen, 510 * 4, hipMemcpyDeviceToHost); hipMemcpy(f.data(), first, 510 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost); hipMemcpy(&k, key, 4, hipMemcpyDeviceToHost);
    int n = 0;
    for (int b = 0; b < 510; ++b)
        if (h[b] > 0) { ++n; printf("workgroup %d saw the word raised in %d polls, first %.2f us after it was raised\n", b, h[b], (f[b] - st[0]) / 100.0); }
    printf("the raising workgroup ran on CU key %d for %.1f us; %d of 510 polling workgroups saw its word\n", k, (st[1] - st[0]) / 100.0, n);
    return 0;
}
