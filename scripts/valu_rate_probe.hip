// Issue cost of the fp64 VALU instructions the covariance kernels use, in cycles per wave64 instruction on one SIMD:
// 8 independent chains per wave, 4 waves per SIMD (so latency is hidden), s_memtime around 4096 instructions.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void k_probe(double* out, long long* cyc, double seed) {
    double v[8];
    int iv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed + 0.001 * (threadIdx.x + 64 * i); iv[i] = (int)threadIdx.x + i; }
    const double c1 = 1.0000001, c2 = 1e-9;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < 512; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
            if (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i]));
            if (OP == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(iv[i]));
            if (OP == 5) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iv[i]) : "v"(v[i]));
            if (OP == 6) asm volatile("v_sqrt_f64 %0, %0" : "+v"(v[i]));
            if (OP == 7) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i]));
            if (OP == 8) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(iv[i]) : "v"(iv[(i + 1) & 7]));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[i]) : "v"(iv[(i + 1) & 7]));
            if (OP == 10) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
            if (OP == 11) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v[i]), "v"(c1) : "vcc");
            if (OP == 12) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[i]));
            if (OP == 13) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(iv[i]));
            if (OP == 14) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[i]) : "v"(iv[i]));
            if (OP == 15) asm volatile("v_fract_f64 %0, %0" : "+v"(v[i]));
        }
    }
    const long long t1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + iv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 1024 * 8 * 8);
    hipMalloc(&cyc, 8 * 8);
    // one workgroup of 16 waves = 4 waves per SIMD on one CU
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_probe<OP>, dim3(1), dim3(1024), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // clock64 = s_memtime at 100 MHz?  report both raw ticks and per-instruction ticks; the fma row is the unit
    const double per = (double)c / (512.0 * 8.0 * 4.0);      // ticks per wave-instruction per SIMD (4 waves share a SIMD)
    printf("%-16s %10lld ticks   %.4f ticks / wave-instruction\n", name, c, per);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f64");
    run<1>("v_mul_f64");
    run<2>("v_add_f64");
    run<3>("v_rndne_f64");
    run<4>("v_ldexp_f64");
    run<5>("v_cvt_i32_f64");
    run<6>("v_sqrt_f64");
    run<7>("v_rcp_f64");
    run<12>("v_rsq_f64");
    run<8>("v_lshl_add_u32");
    run<9>("v_cndmask_b32");
    run<10>("v_max_f64");
    run<11>("v_cmp_lt_f64");
    run<13>("v_fma_f32");
    run<14>("v_cvt_f64_i32");
    run<15>("v_fract_f64");
    return 0;
}
