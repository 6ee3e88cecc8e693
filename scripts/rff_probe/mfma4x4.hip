// layout + cost of v_mfma_f64_4x4x4_4b_f64 on gfx950 (the remainder columns of a Thompson draw: n = 100 features = 6 x 16 + 4)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_layout(double* out) {
    const int l = threadIdx.x;
    // one-hot probes: A = 1 at (lane la), B = 1 at (lane lb): which output lanes light up?
    for (int la = 0; la < 16; ++la)
        for (int lb = 0; lb < 16; ++lb) {
            double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 16 + lb) * 64 + l] = d;
        }
}
__global__ void k_time(double* out, int iters, double a, double b) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (16.0 * iters);
}
int main() {
    double* d; hipMalloc(&d, 16 * 16 * 64 * 8 + 1024);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    static double h[16 * 16 * 64];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // block 0 only (lanes 0..15): print for each (la, lb) the output lanes
    for (int la = 0; la < 16; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 16; ++lb) {
            int n = 0, first = -1;
            for (int l = 0; l < 64; ++l) if (h[(la * 16 + lb) * 64 + l] != 0) { if (first < 0) first = l; ++n; }
            if (n) printf(" B%d->D%d(%d)", lb, first, n);
        }
        printf("\n");
    }
    hipLaunchKernelGGL(k_time, dim3(1), dim3(64), 0, 0, d, 10000, 1.0, 1e-3);
    double c; hipMemcpy(&c, d + 64, 8, hipMemcpyDeviceToHost);
    printf("clocks per v_mfma_f64_4x4x4_4b (one wave, 16 independent accumulators): %.2f\n", c);
    return 0;
}
