// kernels_rff.hip -- Thompson sampling by random Fourier features on gfx950.
//
// Replaces the evaluation of `model.sample_f(n, rng).get(X)` [pybo/policies/simple.py:44-48] over a
// candidate grid.  A draw is f(x) = bias + sum_j theta_j cos(w_j . x + b_j)  (Rahimi & Recht); the
// spectral draws W, b and the weight-posterior noise are made on the host from the caller's seeded
// RandomState so a draw is reproducible; the device does the O(N n^2) feature Gram for the weight
// posterior and the O(M S n d) evaluation sweep.
#include "gpx_internal.h"

namespace gpx {

constexpr int RF_T = 256;   // candidates per block
constexpr int RF_DC = 32;   // coordinates staged per pass

// vals[s][m] = bias + sum_j theta[s][j] * cos(W[s][j][:] . Xc[m][:] + b[s][j])
// grid (ceil(M/256), S).  x tile lives in LDS coordinate-major (conflict-free per-lane reads);
// W/b/theta are wave-uniform -> scalar loads.
__global__ __launch_bounds__(RF_T) void k_rff_eval(const double* __restrict__ W,
                                                   const double* __restrict__ b,
                                                   const double* __restrict__ theta, int n, int d,
                                                   double bias, const double* __restrict__ Xc, int64_t M,
                                                   double* __restrict__ vals) {
    extern __shared__ __attribute__((aligned(16))) double xs[];  // [d][RF_T]
    const int s = blockIdx.y;
    const int t = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * RF_T;
    for (int e = t; e < RF_T * d; e += RF_T) {
        const int row = e / d, k = e - row * d;
        const int64_t gm = m0 + row;
        xs[k * RF_T + row] = (gm < M) ? Xc[gm * d + k] : 0.0;
    }
    __syncthreads();
    const double* Ws = W + (int64_t)s * n * d;
    const double* bs = b + (int64_t)s * n;
    const double* ts = theta + (int64_t)s * n;
    double f = 0.0;
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        double a0 = bs[j], a1 = bs[j + 1], a2 = bs[j + 2], a3 = bs[j + 3];
        for (int k = 0; k < d; ++k) {
            const double xk = xs[k * RF_T + t];
            a0 = fma(Ws[(j + 0) * d + k], xk, a0);
            a1 = fma(Ws[(j + 1) * d + k], xk, a1);
            a2 = fma(Ws[(j + 2) * d + k], xk, a2);
            a3 = fma(Ws[(j + 3) * d + k], xk, a3);
        }
        f = fma(ts[j], cos(a0), f);
        f = fma(ts[j + 1], cos(a1), f);
        f = fma(ts[j + 2], cos(a2), f);
        f = fma(ts[j + 3], cos(a3), f);
    }
    for (; j < n; ++j) {
        double a0 = bs[j];
        for (int k = 0; k < d; ++k) a0 = fma(Ws[j * d + k], xs[k * RF_T + t], a0);
        f = fma(ts[j], cos(a0), f);
    }
    const int64_t gm = m0 + t;
    if (gm < M) vals[(int64_t)s * M + gm] = bias + f;
}

void launch_rff_eval(hipStream_t s, const double* W, const double* b, const double* theta, int S, int n,
                     int d, double bias, const double* Xc, int64_t M, double* vals) {
    dim3 grid((unsigned)((M + RF_T - 1) / RF_T), (unsigned)S);
    const size_t lds = (size_t)d * RF_T * sizeof(double);
    hipLaunchKernelGGL(k_rff_eval, grid, dim3(RF_T), lds, s, W, b, theta, n, d, bias, Xc, M, vals);
}

// Ft[j][i] = cos(w_j . x_i + b_j) for observed points i < N (0 beyond); one thread per (j, i)
__global__ __launch_bounds__(256) void k_rff_features(const double* __restrict__ X, int64_t N, int64_t Np,
                                                      int d, const double* __restrict__ W,
                                                      const double* __restrict__ b, int n,
                                                      double* __restrict__ Ft) {
    const int j = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    double v = 0.0;
    if (i < N) {
        double a = b[j];
        for (int k = 0; k < d; ++k) a = fma(W[j * d + k], X[i * d + k], a);
        v = cos(a);
    }
    Ft[(int64_t)j * Np + i] = v;
}

// A[j1][j2] = sum_i Ft[j1][i] Ft[j2][i];  v[j1] = sum_i Ft[j1][i] (y_i - bias); one wave per output
__global__ __launch_bounds__(256) void k_rff_gram(const double* __restrict__ Ft, int64_t N, int64_t Np,
                                                  int n, const double* __restrict__ y, double bias,
                                                  double* __restrict__ A, double* __restrict__ v) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (int64_t)n * (n + 1)) return;
    const int j1 = (int)(o / (n + 1)), j2 = (int)(o - (int64_t)j1 * (n + 1));
    const double* f1 = Ft + (int64_t)j1 * Np;
    double acc = 0.0;
    if (j2 < n) {
        const double* f2 = Ft + (int64_t)j2 * Np;
        for (int64_t i = lane; i < N; i += 64) acc = fma(f1[i], f2[i], acc);
    } else {
        for (int64_t i = lane; i < N; i += 64) acc = fma(f1[i], y[i] - bias, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        if (j2 < n) A[(int64_t)j1 * n + j2] = acc;
        else v[j1] = acc;
    }
}

// Xraw (N,d) unscaled observed points; Ft scratch (n, Np)
void launch_rff_gram(hipStream_t s, const double* Xraw, const double* Ft_scratch, int64_t N, int d,
                     const double* W, const double* b, int n, const double* y, double bias, double* A,
                     double* v) {
    // Np is recovered by the caller; Ft_scratch is (n, Np) with Np = ceil(N/128)*128
    const int64_t Np = (N + 127) / 128 * 128;
    double* Ft = const_cast<double*>(Ft_scratch);
    dim3 g1((unsigned)((Np + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(k_rff_features, g1, dim3(256), 0, s, Xraw, N, Np, d, W, b, n, Ft);
    const int64_t outs = (int64_t)n * (n + 1);
    hipLaunchKernelGGL(k_rff_gram, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, s, Ft, N, Np, n, y,
                       bias, A, v);
}

}  // namespace gpx
