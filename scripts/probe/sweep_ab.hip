// sweep_ab.hip -- stand-alone A/B of the library's sweep kernel schedules (launch_sweep_trmm, kernels_sweep.hip is compiled
// INTO this binary: what is timed is the library's code).  Synthetic operands; every variant must give the same checksum.
// Usage: sweep_ab.bin N cols reps tile_order[:super_m] [tile_order[:super_m] ...]        (build: scripts/probe/build.sh, -DGPX_SWEEP_PROBES)
//   tile_order < 32: the library's schedules (include/gpx.h).  Probe-only codes: 19 + 32 c -- c = 1 .. 5 cache policy of the DMA loads
//   (sc0, nt, sc1, sc0 sc1, sc0 nt), 6 every tile upwards, 7 one launch per generation of 512 workgroups; 1000 / 1001 / 1002: ONE
//   workgroup per CU on gemm_tile_128_d / _ld / _w<2> (the factorisation's worker loops); 2000 / 2001: the wave-private loop with /
//   without the diagonal-block skip, every tile upwards.  bash scripts/probe/pmc_fetch.sh adds the L2 -> fabric read bytes.
#include "../../pybo_amd/csrc/kernels_sweep.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>
using namespace gpx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void k_fill(double* p, size_t n, unsigned seed, double scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
        p[i] = scale * ((double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
}
// zero the strictly-upper part of T = U^T per 128-block as the library's inverse leaves it: U[k][m] = 0 for k > m
__global__ void k_tri(double* U, int64_t Np) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(Np * Np); i += (size_t)gridDim.x * blockDim.x) {
        const int64_t k = i / Np, m = i - k * Np;
        if (k > m) U[i] = 0.0;
    }
}

// ONE workgroup per compute unit (two k-step images of LDS): the loops of the task-graph factorisation's workers on the sweep's
// tiles.  WHICH 0: gemm_tile_128_d (registers), 1: gemm_tile_128_ld (DMA).  Selected by tile_order 1000 + WHICH.
template <int WHICH>
__global__ __launch_bounds__(GEMM_THREADS, 1) void k_lone(const double* __restrict__ U, int64_t Np, const double* __restrict__ Ks,
                                                          int NT, const double* __restrict__ avec, double* __restrict__ Qp,
                                                          double* __restrict__ Pp, int64_t ldp, int sm) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    const int nP = (int)(Np / TB);
    int mt, nt, mt2;
    if (!sweep_tile_of<32>(blockIdx.x, 3, sm, NT, nP, mt, nt, mt2)) return;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        if (ph == 1) { if (mt2 < 0) break; mt = mt2; __syncthreads(); }
        const int64_t m0 = (int64_t)mt * TB, n0 = (int64_t)nt * TB;
        d4 acc[4][4];
        acc_zero(acc);
        if (WHICH == 0) gemm_tile_128_d<1>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, dsm);
        else if (WHICH == 1) gemm_tile_128_ld<1>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, dsm);
        else gemm_tile_128_w<2, 1>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, dsm);
        sweep_epilogue<false>(acc, avec, m0, Qp + (int64_t)mt * ldp + n0, Pp + (int64_t)mt * ldp + n0, dsm);
    }
}

// two workgroups per CU on the wave-private loop (one image per wave): tile_order 2000 (PRIO 1) / 2001 (no priority changes)
template <int PRIO>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_sweep_w(const double* __restrict__ U, int64_t Np, const double* __restrict__ Ks,
                                                             int NT, const double* __restrict__ avec, double* __restrict__ Qp,
                                                             double* __restrict__ Pp, int64_t ldp, int sm) {
    __shared__ __attribute__((aligned(16))) double smem[4 * GEMM_W_IMG_F64];
    const int nP = (int)(Np / TB);
    int mt, nt, mt2;
    if (!sweep_tile_of<64>(blockIdx.x, 3, sm, NT, nP, mt, nt, mt2)) return;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        if (ph == 1) { if (mt2 < 0) break; mt = mt2; __syncthreads(); }
        const int64_t m0 = (int64_t)mt * TB, n0 = (int64_t)nt * TB;
        d4 acc[4][4];
        acc_zero(acc);
        gemm_tile_128_w<1, 1, false, true, 0, PRIO != 0>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, smem);      // PRIO reused: 1 = with the diagonal-block skip
        sweep_epilogue<true>(acc, avec, m0, Qp + (int64_t)mt * ldp + n0, Pp + (int64_t)mt * ldp + n0, smem);
    }
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: N cols reps tile_order[:super_m] ...\n"); return 1; }
    const int64_t N = atoll(argv[1]), cols = atoll(argv[2]);
    const int reps = atoi(argv[3]);
    const int64_t Np = (N + 127) / 128 * 128;
    const int nP = (int)(Np / 128);
    double *U, *Ks, *a, *Qp, *Pp;
    unsigned long long* clk;
    CK(hipMalloc(&U, Np * Np * 8));
    CK(hipMalloc(&Ks, cols * Np * 8));
    CK(hipMalloc(&a, Np * 8));
    CK(hipMalloc(&Qp, (size_t)nP * cols * 8));
    CK(hipMalloc(&Pp, (size_t)nP * cols * 8));
    CK(hipMalloc(&clk, 16));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, U, (size_t)(Np * Np), 1u, 1.0);
    hipLaunchKernelGGL(k_tri, dim3(4096), dim3(256), 0, 0, U, Np);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Ks, (size_t)(cols * Np), 2u, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, a, (size_t)Np, 3u, 1.0);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> rq, rp;      // the first variant's results: the reference of the others
    for (int v = 4; v < argc; ++v) {
        int to = atoi(argv[v]), sm = 8;
        if (const char* c = strchr(argv[v], ':')) sm = atoi(c + 1);
        std::vector<float> ms_all;
        CK(hipMemset(Qp, 0xff, (size_t)nP * cols * 8));
        CK(hipMemset(Pp, 0xff, (size_t)nP * cols * 8));
        double mhz = 0;
        for (int rep = 0; rep < reps + 2; ++rep) {
            CK(hipMemsetAsync(clk, 0, 16, 0));
            hipEventRecord(e0);
            if (to >= 2000) {
                const unsigned nblk = sweep_grid<64>(3, sm, (int)(cols / TB), nP);
                if (to == 2000) hipLaunchKernelGGL(k_sweep_w<1>, dim3(nblk), dim3(GEMM_THREADS), 0, 0, U, Np, Ks, (int)(cols / TB), a, Qp, Pp, cols, sm);
                else hipLaunchKernelGGL(k_sweep_w<0>, dim3(nblk), dim3(GEMM_THREADS), 0, 0, U, Np, Ks, (int)(cols / TB), a, Qp, Pp, cols, sm);
            } else if (to >= 1000) {
                const size_t lb = (size_t)2 * GEMM_LDS_F64 * 8;
                const unsigned nblk = sweep_grid<32>(3, sm, (int)(cols / TB), nP);
                if (to == 1000) { CK(hipFuncSetAttribute((const void*)k_lone<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
                    hipLaunchKernelGGL(k_lone<0>, dim3(nblk), dim3(GEMM_THREADS), lb, 0, U, Np, Ks, (int)(cols / TB), a, Qp, Pp, cols, sm); }
                else if (to == 1001) { CK(hipFuncSetAttribute((const void*)k_lone<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
                    hipLaunchKernelGGL(k_lone<1>, dim3(nblk), dim3(GEMM_THREADS), lb, 0, U, Np, Ks, (int)(cols / TB), a, Qp, Pp, cols, sm); }
                else { CK(hipFuncSetAttribute((const void*)k_lone<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
                    hipLaunchKernelGGL(k_lone<2>, dim3(nblk), dim3(GEMM_THREADS), lb, 0, U, Np, Ks, (int)(cols / TB), a, Qp, Pp, cols, sm); }
            } else
            launch_sweep_trmm(0, U, Np, Ks, Np, cols, a, Qp, Pp, cols, to, sm, clk);
            hipEventRecord(e1);
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) ms_all.push_back(ms);
            unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
        }
        std::sort(ms_all.begin(), ms_all.end());
        const double med = ms_all[ms_all.size() / 2];
        std::vector<double> hq((size_t)nP * cols), hp((size_t)nP * cols);
        CK(hipMemcpy(hq.data(), Qp, hq.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hp.data(), Pp, hp.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long x = 0;
        for (size_t i = 0; i < hq.size(); ++i) { unsigned long long b; memcpy(&b, &hq[i], 8); x = x * 1099511628211ull ^ b; memcpy(&b, &hp[i], 8); x = x * 1099511628211ull ^ b; }
        double dq = 0, dp = 0, sq = 0, sp = 0;
        if (rq.empty()) { rq = hq; rp = hp; }
        for (size_t i = 0; i < hq.size(); ++i) {
            dq = std::max(dq, fabs(hq[i] - rq[i])); sq = std::max(sq, fabs(rq[i]));
            dp = std::max(dp, fabs(hp[i] - rp[i])); sp = std::max(sp, fabs(rp[i]));
        }
        printf("RESULT tile_order %d super_m %d N %lld cols %lld: min %.3f median %.3f ms  %.2f TFLOP/s  frac %.4f  sclk %.0f MHz  checksum %016llx  maxdiff/scale q %.2e p %.2e\n",
               to, sm, (long long)N, (long long)cols, ms_all[0], med, (double)N * N * cols / med / 1e9, (double)N * N * cols / med / 1e9 / 78.6, mhz, x, dq / sq, dp / sp);
        fflush(stdout);
    }
    return 0;
}
