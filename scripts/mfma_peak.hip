// fp64 MFMA ceiling microbenchmark: back-to-back v_mfma_f64_16x16x4_f64 on independent accumulators.
// Usage: mfma_peak.bin [waves_per_simd=1] [iters=20000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 1;
    int iters = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    int threads = 256, blocks = cus * wps;   // 4 waves per block = 1 wave per SIMD per block
    double* d; hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 1e-3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 16 * 2048.0;
        printf("wps=%d acc=16 iters=%d: %.3f ms  %.2f TFLOP/s  (%.1f cycles/mfma/SIMD @2.4GHz)\n", wps, iters, ms,
               flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(threads), 0, 0, d, iters * 4, 1.0, 1e-3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 4 * 4 * 2048.0;
        printf("wps=%d acc=4: %.3f ms  %.2f TFLOP/s\n", wps, ms, flop / ms / 1e9);
    }
    return 0;
}
