// kernels_ens.hip -- (1) the hyper-parameter ENSEMBLE sweep: pybo's default model is reggie.MCMC(gp, n=10)
// (/root/reference/pybo/bayesopt.py:115), whose acquisition is the average over the member GPs; the
// member sweeps stay on the device and only the k winners of the averaged index come back.
// (2) candidate grids generated in HBM: the solver's grid
// (/root/reference/pybo/solvers/lbfgs.py:45 `init_uniform`, pybo/inits/methods.py:24-38,62-77) never crosses
// PCIe and stays resident for the whole BO run.
//
// All of this is HBM-bound elementwise work: one pass, coalesced, grid-stride.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpx_internal.h"

// Results here are compared bit-for-bit with host arithmetic (numpy): no mul+add contraction into FMA
// (the default is -ffp-contract=fast; HIP's __dmul_rn / __dadd_rn are header functions compiled under that
// default and fuse as well, so plain operators under this pragma are used).
#pragma clang fp contract(off)

namespace gpx {

// ---------------------------------------------------------------------------------------------------
// ensemble accumulate / finish.  Order of the additions = member order, then ONE division by n: the same
// association as numpy's mean over axis 0, so the result does not depend on launch geometry.
//   mode 0 (EI / PI / mean):  acc0 += t0
//   mode 1 (UCB, mixture moments):  acc0 += mu_m ;  acc1 += s2_m + mu_m^2
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ens_accum(double* __restrict__ acc0, double* __restrict__ acc1,
                                                   const double* __restrict__ t0,
                                                   const double* __restrict__ t1, int64_t M, int mode,
                                                   int first) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        const double m = t0[i];
        if (mode == 0) {
            acc0[i] = first ? m : acc0[i] + m;
        } else {
            const double m2 = m * m;
            const double q = t1[i] + m2;
            acc0[i] = first ? m : acc0[i] + m;
            acc1[i] = first ? q : acc1[i] + q;
        }
    }
}

__global__ __launch_bounds__(256) void k_ens_finish(const double* __restrict__ acc0,
                                                    const double* __restrict__ acc1, int64_t M, int mode,
                                                    double n, double beta, double* __restrict__ out,
                                                    double* __restrict__ mu_out,
                                                    double* __restrict__ s2_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        const double mu = acc0[i] / n;
        if (mode == 0) {
            out[i] = mu;
        } else {
            const double mu2 = mu * mu;
            double s2 = acc1[i] / n - mu2;
            s2 = fmax(s2, 0.0);
            if (mu_out) mu_out[i] = mu;
            if (s2_out) s2_out[i] = s2;
            if (out) out[i] = mu + sqrt(beta * s2);
        }
    }
}

static inline int ew_blocks(int64_t M) {
    int64_t b = (M + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

void launch_ens_accum(hipStream_t s, double* acc0, double* acc1, const double* t0, const double* t1, int64_t M,
                      int mode, int first) {
    hipLaunchKernelGGL(k_ens_accum, dim3(ew_blocks(M)), dim3(256), 0, s, acc0, acc1, t0, t1, M, mode, first);
}

void launch_ens_finish(hipStream_t s, const double* acc0, const double* acc1, int64_t M, int mode, double n,
                       double beta, double* out, double* mu_out, double* s2_out) {
    hipLaunchKernelGGL(k_ens_finish, dim3(ew_blocks(M)), dim3(256), 0, s, acc0, acc1, M, mode, n, beta, out,
                       mu_out, s2_out);
}

// ---------------------------------------------------------------------------------------------------
// candidate grids
// ---------------------------------------------------------------------------------------------------
// Sobol', unscrambled, in Gray-code order (point i = XOR of the direction numbers v_b over the set bits b of
// i ^ (i >> 1)) -- the order scipy.stats.qmc.Sobol produces; the direction numbers (d, bits) come from the
// caller (the binding reads scipy's Joe-Kuo table; nothing is embedded here).  u = acc * 2^-bits exactly,
// x = lo + u * (hi - lo) with separate multiply and add (no contraction) so the points equal the host's
// `lo + sample * (hi - lo)` bit for bit.
__global__ __launch_bounds__(256) void k_grid_sobol(const uint32_t* __restrict__ sv, int bits, int64_t first,
                                                    int64_t M, int d, const double* __restrict__ bounds,
                                                    double* __restrict__ X) {
    const int64_t total = M * d;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double scale = ldexp(1.0, -bits);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / d;
        const int j = (int)(e - i * d);
        uint64_t g = (uint64_t)(i + first);
        g ^= g >> 1;
        uint32_t acc = 0;
        const uint32_t* v = sv + (int64_t)j * bits;
        for (int b = 0; b < bits && g; ++b, g >>= 1)
            if (g & 1) acc ^= v[b];
        const double lo = bounds[2 * j], hi = bounds[2 * j + 1];
        const double w = hi - lo, u = (double)acc * scale;
        X[e] = lo + u * w;
    }
}

// Uniform grid: counter-based Philox4x32-10 (Salmon et al. 2011), counter = (element pair index, 0, 0, 0),
// key = the 64-bit seed; each 4x32 output gives two 53-bit uniforms in [0, 1): element 2c and 2c+1 of the
// row-major (M, d) array.  Independent of launch geometry; restated on the host in oracle/gp_ref.py.
__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void k_grid_uniform(uint64_t seed, int64_t first_elem, int64_t total, int d,
                                                      const double* __restrict__ bounds,
                                                      double* __restrict__ X) {
    // rows first .. first+M-1 of the stream: GLOBAL element E = first_elem + e lives in output E >> 1, half E & 1,
    // so a shard generated with an offset holds exactly the numbers the whole grid holds at those rows
    const int64_t gp0 = first_elem >> 1;
    const int64_t pairs = ((first_elem + total + 1) >> 1) - gp0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += stride) {
        const uint64_t ctr = (uint64_t)(p + gp0);
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const double u[2] = {
            (double)((((uint64_t)c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0),
            (double)((((uint64_t)c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0)};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t e = 2 * (p + gp0) + t - first_elem;
            if (e >= 0 && e < total) {
                const int j = (int)(e % d);
                const double lo = bounds[2 * j], hi = bounds[2 * j + 1];
                const double w = hi - lo;
                X[e] = lo + u[t] * w;
            }
        }
    }
}

__global__ void k_grid_gather(const double* __restrict__ X, int d, const int64_t* __restrict__ idx, int64_t k,
                              double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < k * d) out[e] = X[idx[e / d] * d + (e % d)];
}

void launch_grid_sobol(hipStream_t s, const uint32_t* sv, int bits, int64_t first, int64_t M, int d,
                       const double* bounds, double* X) {
    hipLaunchKernelGGL(k_grid_sobol, dim3(ew_blocks(M * d)), dim3(256), 0, s, sv, bits, first, M, d, bounds, X);
}

void launch_grid_uniform(hipStream_t s, uint64_t seed, int64_t first, int64_t M, int d, const double* bounds,
                         double* X) {
    hipLaunchKernelGGL(k_grid_uniform, dim3(ew_blocks((M * d + 1) / 2 + 1)), dim3(256), 0, s, seed, first * d,
                       M * d, d, bounds, X);
}

void launch_grid_gather(hipStream_t s, const double* X, int d, const int64_t* idx, int64_t k, double* out) {
    const int64_t n = k * d;
    hipLaunchKernelGGL(k_grid_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, d, idx, k, out);
}

}  // namespace gpx
