// kernels_grad.hip -- posterior moments WITH input gradients for small candidate batches, and the
// gradient of an RFF function sample.  These serve the `f(x[None], grad=True)` calls of the L-BFGS
// refinement (/root/reference/pybo/solvers/lbfgs.py:56-58) reaching model.predict(X, True)
// (pybo/policies/simple.py:64-70, pybo/recommenders.py:22) and sample_f(...).get(X, True).
//
//   k* = k(X_obs, x)            g = dk/dr2
//   V  = T k*                   mu = bias + V.a          s2 = rho - V.V
//   w  = U V = K^-1 k*          dmu/dx_j = sum_i dk_i/dx_j alpha_i      ds2/dx_j = -2 sum_i dk_i/dx_j w_i
//   dk_i/dx_j = g_i * 2 (x_j - X_ij) / ell_j^2
//
// Memory-bound by design: one pass over T and one over U per batch of up to GB candidates (matvecs
// with GB right-hand sides, one wave per matrix row, coalesced row reads).
//
// Round 4, the ONE-PASS form for a single point (every call of the reference's single-seed refinement, lbfgs.py:56-58):
//   ds2/dx_j = -2 (T k*).(T dk*/dx_j):  the d derivative vectors travel as extra right-hand sides of the SAME pass over T,
//   and U is not read at all -- 1 + d right-hand sides, one pass (k_tri_matvec_rb), instead of two dependent passes.
#include <string.h>

#include <algorithm>

#include "gpx_internal.h"
#include "gpx_math.h"

namespace gpx {

constexpr int GB = 16;  // candidates per pass over T and U (the 10 lock-step L-BFGS seeds of solve_lbfgs fit in one)

__device__ __forceinline__ void kern_and_grad(int kid, double r2, double rho, double& k, double& g) {
    switch (kid) {
        case GPX_KERN_SE_ARD: {
            k = rho * exp_nonpos(-0.5 * r2);
            g = -0.5 * k;
            break;
        }
        case GPX_KERN_MATERN52: {
            const double s = 2.23606797749978969641 * sqrt_r2(r2);
            const double e = rho * exp_nonpos(-s);
            k = (1.0 + s + (5.0 / 3.0) * r2) * e;
            g = -(5.0 / 6.0) * (1.0 + s) * e;
            break;
        }
        case GPX_KERN_MATERN32: {
            const double s = 1.73205080756887729353 * sqrt_r2(r2);
            const double e = rho * exp_nonpos(-s);
            k = (1.0 + s) * e;
            g = -1.5 * e;
            break;
        }
        default: {
            const double r = sqrt_r2(r2);
            k = rho * exp_nonpos(-r);
            // exp(-r) has a kink at r = 0 (a candidate on top of an observation: the L-BFGS seeds of the
            // recommender ARE observations): dk/dx is +-1 from either side, take the symmetric value 0
            g = (r > 0.0) ? -0.5 * k / r : 0.0;
        }
    }
}

// ks[m][i], g[m][i] for i < Np (0 beyond N); grid (Np/256, mb)
__global__ __launch_bounds__(256) void k_kstar(const double* __restrict__ Xs, int64_t N, int64_t Np, int d,
                                               const double* __restrict__ Xc, const double* __restrict__ invell,
                                               int kid, double rho, double* __restrict__ ks,
                                               double* __restrict__ g) {
    const int m = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    double kv = 0.0, gv = 0.0;
    if (i < N) {
        double r2 = 0.0;
        for (int k = 0; k < d; ++k) {
            const double df = Xs[i * d + k] - Xc[(int64_t)m * d + k] * invell[k];
            r2 = fma(df, df, r2);
        }
        kern_and_grad(kid, r2, rho, kv, gv);
    }
    ks[(int64_t)m * Np + i] = kv;
    g[(int64_t)m * Np + i] = gv;
}

// out[m][row] = sum_j Mx[row][j] * in[m][j], j in [0,row] (mode 0, lower) or [row,N) (mode 1, upper)
// MBT: compile-time bucket (>= mb) so that a single right-hand side (gpx_append, one L-BFGS instance) does not pay
// for the 16-wide register loop of a full batch
template <int MBT>
__global__ __launch_bounds__(256) void k_tri_matvec_multi(const double* __restrict__ Mx, int64_t Np, int64_t N,
                                                          const double* __restrict__ in, int mb, int mode,
                                                          double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Np) return;
    double acc[MBT];
#pragma unroll
    for (int m = 0; m < MBT; ++m) acc[m] = 0.0;
    if (row < N) {
        const int64_t lo = (mode == 0) ? 0 : row;
        const int64_t hi = (mode == 0) ? row + 1 : N;
        const double* mr = Mx + row * Np;
        {
            // The pass is a pure HBM stream of the matrix row: 16-byte loads over the aligned pairs that cover
            // [lo, hi); elements outside [lo, hi) are dropped by select, never multiplied.  With few right-hand sides
            // (one L-BFGS instance, gpx_append) four loads are in flight per lane (2 KB per wave and step).  Every
            // variant adds a lane's pairs in the same order (p0 + lane, + 64, ...; x before y), so a row's result does
            // not depend on the batch it is evaluated in (the lock-step refinement relies on that, tested).
            const int64_t p0 = lo >> 1, p1 = (hi + 1) >> 1;          // pairs [p0, p1)
            const double2* mr2 = reinterpret_cast<const double2*>(mr);
            int64_t p = p0 + lane;
            for (; MBT <= 4 && p + 192 < p1; p += 256) {
                double2 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = mr2[p + 64 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t j = 2 * (p + 64 * u);
                    const double tx = (j >= lo) ? t[u].x : 0.0;       // (j + 1 < hi and j < hi hold inside the main loop
                    const double ty = (j + 1 < hi) ? t[u].y : 0.0;    //  except at the two ragged ends)
#pragma unroll
                    for (int m = 0; m < MBT; ++m)
                        if (m < mb) {
                            const double2 v = *reinterpret_cast<const double2*>(in + (int64_t)m * Np + j);
                            acc[m] = fma(tx, v.x, fma(ty, v.y, acc[m]));
                        }
                }
            }
            for (; p < p1; p += 64) {
                const double2 t = mr2[p];
                const int64_t j = 2 * p;
                const double tx = (j >= lo && j < hi) ? t.x : 0.0;
                const double ty = (j + 1 >= lo && j + 1 < hi) ? t.y : 0.0;
#pragma unroll
                for (int m = 0; m < MBT; ++m)
                    if (m < mb) {
                        const double2 v = *reinterpret_cast<const double2*>(in + (int64_t)m * Np + j);
                        acc[m] = fma(tx, v.x, fma(ty, v.y, acc[m]));
                    }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MBT; ++m) {
        double a = acc[m];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0 && m < mb) out[(int64_t)m * Np + row] = a;
    }
}

static void launch_tri_matvec_multi(hipStream_t s, unsigned rows4, const double* Mx, int64_t Np, int64_t N,
                                    const double* in, int mb, int mode, double* out) {
    if (mb <= 1)
        hipLaunchKernelGGL(k_tri_matvec_multi<1>, dim3(rows4), dim3(256), 0, s, Mx, Np, N, in, mb, mode, out);
    else if (mb <= 4)
        hipLaunchKernelGGL(k_tri_matvec_multi<4>, dim3(rows4), dim3(256), 0, s, Mx, Np, N, in, mb, mode, out);
    else if (mb <= 8)
        hipLaunchKernelGGL(k_tri_matvec_multi<8>, dim3(rows4), dim3(256), 0, s, Mx, Np, N, in, mb, mode, out);
    else
        hipLaunchKernelGGL(k_tri_matvec_multi<GB>, dim3(rows4), dim3(256), 0, s, Mx, Np, N, in, mb, mode, out);
}

// ------------------------------------------------------------------------------------------------
// Register-blocked triangular matvec with several right-hand sides.  k_tri_matvec_multi reads every right-hand side
// once per matrix ROW (one wave per row): with 9 right-hand sides the vector loads are 9x the matrix stream and the
// pass takes 180 us instead of 58 (N = 8192; the L1's 64 B/clk per CU is the limit).  Here a wave owns RB_R = 4
// consecutive rows over one segment of RB_CS columns: a right-hand-side pair is loaded once and used by all four rows
// (vector traffic 1/4), and the pieces (row group x column segment) are all about the same size, so the triangular
// matrix needs no dynamic balancing.  Each piece writes its partial sums part[seg][m][row]; the consumer adds the
// ceil(N / RB_CS) partials of a row in ascending order (empty pieces write zeros), so a row's value depends on neither the
// batch it travels in nor the launch order.  mode 0: lower (columns [0, row]), mode 1: upper (columns [row, N)).
// ------------------------------------------------------------------------------------------------
constexpr int RB_CS_DEFAULT = 2048;

template <int MBT, int R, bool MASK>
__device__ __forceinline__ void rb_piece(const double* __restrict__ Mx, int64_t Np, int64_t N, int64_t r0,
                                         const double* __restrict__ in, int mode, int64_t cb, int64_t ce,
                                         double (&acc)[R][MBT]) {
    const int lane = threadIdx.x & 63;
    const double* mr = Mx + r0 * Np;
#pragma unroll 2
    for (int64_t c = cb + 2 * lane; c < ce; c += 128) {
        double2 t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = *reinterpret_cast<const double2*>(mr + r * Np + c);
        if (MASK) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = r0 + r;
                const bool okx = (mode == 0 ? c <= row : c >= row) && c < N && row < N;
                const bool oky = (mode == 0 ? c + 1 <= row : c + 1 >= row) && c + 1 < N && row < N;
                t[r].x = okx ? t[r].x : 0.0;
                t[r].y = oky ? t[r].y : 0.0;
            }
        }
        // (MBT is the EXACT number of right-hand sides: a run-time "m < mb" test around each vector load made the loads of
        //  a step wait for one another -- nine dependent L2 round trips per step)
        double2 v[MBT];
#pragma unroll
        for (int m = 0; m < MBT; ++m) v[m] = *reinterpret_cast<const double2*>(in + (int64_t)m * Np + c);
#pragma unroll
        for (int m = 0; m < MBT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][m] = fma(t[r].x, v[m].x, fma(t[r].y, v[m].y, acc[r][m]));
    }
}

// Sums over the 64 lanes of NV = 2^K per-lane values at once: stage s exchanges HALF of the values across lane bit 5 - s
// (the lanes with the bit clear keep the even value of a pair, the others the odd one), so the number of live values
// halves per stage -- NV - 1 shuffles instead of 6 NV -- and the last 6 - K stages are plain butterflies on the one value
// left.  Every value is added in the order of the plain xor butterfly (32, 16, .., 1), whatever it is paired with: the
// result does not depend on NV.  Afterwards the lanes whose low 6 - K bits are 0 hold value bitreverse_K(lane >> (6 - K)).
template <int NV>
__device__ __forceinline__ void lanes_sum_many(double (&v)[NV]) {
    const int lane = threadIdx.x & 63;
    int n = NV, off = 32;
#pragma unroll
    for (; n > 1; n >>= 1, off >>= 1) {
        const bool up = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < NV / 2; ++i)
            if (i < n / 2) {
                const double keep = up ? v[2 * i + 1] : v[2 * i];
                const double send = up ? v[2 * i] : v[2 * i + 1];
                v[i] = keep + __shfl_xor(send, off);
            }
    }
#pragma unroll
    for (; off > 0; off >>= 1) v[0] += __shfl_xor(v[0], off);
}
constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
constexpr int pow2_at_least(int v) { return v <= 1 ? 1 : 2 * pow2_at_least((v + 1) / 2); }

// grid (Np / R, nseg), one wave per workgroup; cs = columns per segment (a multiple of 128)
template <int MBT, int R>
__global__ __launch_bounds__(64) void k_tri_matvec_rb(const double* __restrict__ Mx, int64_t Np, int64_t N,
                                                      const double* __restrict__ in, int mode, int cs,
                                                      double* __restrict__ part) {
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int64_t c0 = (int64_t)blockIdx.y * cs;
    double acc[R][MBT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MBT; ++m) acc[r][m] = 0.0;
    // columns any of the R rows needs: lower [0, r0 + R - 1], upper [r0, N)
    const int64_t lo = (mode == 0) ? 0 : (r0 & ~(int64_t)1);
    const int64_t hi = (mode == 0) ? (r0 + R < N ? r0 + R : N) : N;
    const int64_t he = (hi + 1) & ~(int64_t)1;
    const int64_t cb = c0 > lo ? c0 : lo, ce = c0 + cs < he ? c0 + cs : he;
    if (r0 < N && cb < ce) {
        const bool full = (r0 + R <= N) && (mode == 0 ? (ce - 1 <= r0) : (cb >= r0 + R - 1 && ce <= N));
        if (full) rb_piece<MBT, R, false>(Mx, Np, N, r0, in, mode, cb, ce, acc);
        else rb_piece<MBT, R, true>(Mx, Np, N, r0, in, mode, cb, ce, acc);
    }
    double* out = part + (int64_t)blockIdx.y * MBT * Np;
    constexpr int NV = pow2_at_least(R * MBT), K = ilog2c(NV);
    static_assert(NV <= 64 || R * MBT <= 128, "too many values");
    if constexpr (NV <= 64) {
        double v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = (i < R * MBT) ? acc[i % R][i / R] : 0.0;       // value i = (row i % R, vector i / R)
        lanes_sum_many<NV>(v);
        if ((lane & ((64 >> K) - 1)) == 0) {
            const int i = (int)(__brev((unsigned)(lane >> (6 - K))) >> (32 - K));
            if (i < R * MBT) out[(int64_t)(i / R) * Np + r0 + (i % R)] = v[0];
        }
    } else {
        // more than 64 values: two halves of the vectors, 64 values each
        constexpr int MH = MBT / 2;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int m0 = hf * MH, mc = hf == 0 ? MH : MBT - MH;
            double v[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) v[i] = (i < R * mc) ? acc[i % R][m0 + i / R] : 0.0;
            lanes_sum_many<64>(v);
            const int i = (int)(__brev((unsigned)lane) >> 26);
            if (i < R * mc) out[(int64_t)(m0 + i / R) * Np + r0 + (i % R)] = v[0];
        }
    }
}

// out[m][i] = sum over the nseg partials, ascending; grid (Np / 256, mb)
__global__ __launch_bounds__(256) void k_part_sum(const double* __restrict__ part, int64_t Np, int mb, int nseg,
                                                  double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (i >= Np) return;
    double a = 0.0;
    for (int s2 = 0; s2 < nseg; ++s2) a += part[((int64_t)s2 * mb + m) * Np + i];
    out[(int64_t)m * Np + i] = a;
}

static inline int rb_cs(const gpx_handle* h) { return h->grad_rb_cs > 0 ? h->grad_rb_cs : RB_CS_DEFAULT; }
static inline int rb_nseg(const gpx_handle* h, int64_t N) { return (int)((N + rb_cs(h) - 1) / rb_cs(h)); }

template <int R>
static void launch_tri_matvec_rb_r(hipStream_t s, const double* Mx, int64_t Np, int64_t N, const double* in, int mb,
                                   int mode, int cs, double* part) {
    const dim3 g((unsigned)(Np / R), (unsigned)((N + cs - 1) / cs));
#define GPX_RB_CASE(n) case n: hipLaunchKernelGGL((k_tri_matvec_rb<n, R>), g, dim3(64), 0, s, Mx, Np, N, in, mode, cs, part); break;
    switch (mb) {
        GPX_RB_CASE(1) GPX_RB_CASE(2) GPX_RB_CASE(3) GPX_RB_CASE(4) GPX_RB_CASE(5) GPX_RB_CASE(6) GPX_RB_CASE(7) GPX_RB_CASE(8)
        GPX_RB_CASE(9) GPX_RB_CASE(10) GPX_RB_CASE(11) GPX_RB_CASE(12) GPX_RB_CASE(13) GPX_RB_CASE(14) GPX_RB_CASE(15)
        GPX_RB_CASE(16)
    }
#undef GPX_RB_CASE
}
// (8 rows per wave -- half the vector traffic again -- was measured slower: 256 registers, one wave per SIMD, 139-144 us per
//  one-pass call under the profiler against 119-128; segments of 1024-2048 columns tie, 512 and 4096 lose)
static void launch_tri_matvec_rb(const gpx_handle* h, hipStream_t s, const double* Mx, int64_t Np, int64_t N,
                                 const double* in, int mb, int mode, double* part) {
    launch_tri_matvec_rb_r<4>(s, Mx, Np, N, in, mb, mode, rb_cs(h), part);
}

// A single query point travels in the kernel arguments (an H2D copy of 64 bytes costs 4 us of stream time and more of the
// host's); batches read the uploaded array.
struct XArg { double v[GB]; };

// One-pass form, right-hand sides of point m: rhs[(m (1 + d) + 0)][i] = k*_i, rhs[m (1 + d) + 1 + j][i] = dk*_i / dx_j
// (0 beyond N); grid (Np / 256, 1 + d, mb): every vector has its own blocks (the distance is recomputed, the 1 + d
// stores of a point no longer wait for one another in 32 blocks)
__global__ __launch_bounds__(256) void k_kstar_d(const double* __restrict__ Xs, int64_t N, int64_t Np, int d,
                                                 const double* __restrict__ Xc, XArg xa,
                                                 const double* __restrict__ invell,
                                                 int kid, double rho, double* __restrict__ rhs) {
    const int m = blockIdx.z, c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    double kv = 0.0, gv = 0.0, out = 0.0;
    if (i < N) {
        double r2 = 0.0;
        for (int k = 0; k < d; ++k) {
            const double xk = Xc ? Xc[(int64_t)m * d + k] : xa.v[k];
            const double df = Xs[i * d + k] - xk * invell[k];
            r2 = fma(df, df, r2);
        }
        kern_and_grad(kid, r2, rho, kv, gv);
        if (c == 0) {
            out = kv;
        } else {
            const int j = c - 1;
            const double cj = (Xc ? Xc[(int64_t)m * d + j] : xa.v[j]) * invell[j];
            out = gv * (2.0 * invell[j] * (cj - Xs[i * d + j]));
        }
    }
    rhs[((int64_t)m * (1 + d) + c) * Np + i] = out;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// grid (d+1, mb).  blockIdx.x < d: gradient component j; blockIdx.x == d: mu and s2.
// out layout per candidate m: [mu, s2, dmu[0..d), ds2[0..d)]
__global__ __launch_bounds__(256) void k_grad_reduce(const double* __restrict__ Xs, int64_t N, int64_t Np,
                                                     int d, const double* __restrict__ Xc,
                                                     const double* __restrict__ invell,
                                                     const double* __restrict__ g, const double* __restrict__ V,
                                                     const double* __restrict__ w, const double* __restrict__ a,
                                                     const double* __restrict__ alpha, double rho, double bias,
                                                     double* __restrict__ out) {
    __shared__ double sh[4];
    const int j = blockIdx.x, m = blockIdx.y;
    const int64_t base = (int64_t)m * Np;
    double* o = out + (int64_t)m * (2 + 2 * d);
    if (j == d) {
        double p = 0.0, q = 0.0;
        for (int64_t i = threadIdx.x; i < N; i += 256) {
            const double v = V[base + i];
            p = fma(v, a[i], p);
            q = fma(v, v, q);
        }
        p = block_sum(p, sh);
        q = block_sum(q, sh);
        if (threadIdx.x == 0) {
            o[0] = bias + p;
            o[1] = fmax(rho - q, 1e-100);
        }
        return;
    }
    const double cj = Xc[(int64_t)m * d + j] * invell[j];
    const double two_inv = 2.0 * invell[j];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        const double dk = g[base + i] * (two_inv * (cj - Xs[i * d + j]));
        s1 = fma(dk, alpha[i], s1);
        s2 = fma(dk, w[base + i], s2);
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        o[2 + j] = s1;
        o[2 + d + j] = -2.0 * s2;
    }
}

// The mean alone needs neither T nor U: mu = bias + k(x, X).alpha, dmu_j = sum_i alpha_i dk_i/dx_j.  Same grid and
// output layout as k_grad_reduce (the s2 slots are left untouched).
__global__ __launch_bounds__(256) void k_mean_reduce(const double* __restrict__ Xs, int64_t N, int64_t Np, int d,
                                                     const double* __restrict__ Xc,
                                                     const double* __restrict__ invell,
                                                     const double* __restrict__ ks, const double* __restrict__ g,
                                                     const double* __restrict__ alpha, double bias,
                                                     double* __restrict__ out) {
    __shared__ double sh[4];
    const int j = blockIdx.x, m = blockIdx.y;
    const int64_t base = (int64_t)m * Np;
    double* o = out + (int64_t)m * (2 + 2 * d);
    double acc = 0.0;
    if (j == d) {
        for (int64_t i = threadIdx.x; i < N; i += 256) acc = fma(ks[base + i], alpha[i], acc);
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) o[0] = bias + acc;
        return;
    }
    const double cj = Xc[(int64_t)m * d + j] * invell[j];
    const double two_inv = 2.0 * invell[j];
    for (int64_t i = threadIdx.x; i < N; i += 256)
        acc = fma(g[base + i] * (two_inv * (cj - Xs[i * d + j])), alpha[i], acc);
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) o[2 + j] = acc;
}

// One-pass form: grid (d + 1, mb), 1024 threads, same output layout as k_grad_reduce.  part[s][m (1 + d) + c][i]: the partial
// sums of T applied to right-hand side c of point m (k_tri_matvec_rb); V = sum_s part[s][.. + 0], Vd_j = sum_s part[s][.. + 1 + j].
__device__ __forceinline__ double block_sum16(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double a = 0.0;
    for (int i = 0; i < 16; ++i) a += sh[i];
    return a;
}
__global__ __launch_bounds__(1024) void k_grad_reduce_1p(int64_t N, int64_t Np, int d, const double* __restrict__ rhs,
                                                         const double* __restrict__ part, int nseg, int nrhs,
                                                         const double* __restrict__ a,
                                                         const double* __restrict__ alpha, double rho, double bias,
                                                         double* __restrict__ out) {
    __shared__ double sh[16];
    const int j = blockIdx.x, m = blockIdx.y;
    const int c0 = m * (1 + d);
    double* o = out + (int64_t)m * (2 + 2 * d);
    // T applied to right-hand side c, element i: the partial sums of the column segments, ascending
    auto applied = [&](int c, int64_t i) {
        double v = 0.0;
#pragma unroll 4
        for (int sg = 0; sg < nseg; ++sg) v += part[((int64_t)sg * nrhs + c) * Np + i];
        return v;
    };
    if (j == d) {
        double p = 0.0, q = 0.0;
        for (int64_t i = threadIdx.x; i < N; i += 1024) {
            const double v = applied(c0, i);
            p = fma(v, a[i], p);
            q = fma(v, v, q);
        }
        p = block_sum16(p, sh);
        q = block_sum16(q, sh);
        if (threadIdx.x == 0) {
            o[0] = bias + p;
            o[1] = fmax(rho - q, 1e-100);
        }
        return;
    }
    const double* dk = rhs + (int64_t)(c0 + 1 + j) * Np;
    double s1 = 0.0, s2v = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 1024) {
        s1 = fma(dk[i], alpha[i], s1);
        s2v = fma(applied(c0, i), applied(c0 + 1 + j, i), s2v);
    }
    s1 = block_sum16(s1, sh);
    s2v = block_sum16(s2v, sh);
    if (threadIdx.x == 0) {
        o[2 + j] = s1;
        o[2 + d + j] = -2.0 * s2v;
    }
}

// The mean and its gradient of ONE point in one launch (the recommender's refinement, pybo/recommenders.py:22 through
// solve_lbfgs): grid (d + 1), 1024 threads; every block recomputes the N distances (N d flops: nothing) instead of
// reading k* and g back from memory, the point travels in the arguments and the results go straight to the pinned host
// buffer.  Same sums in the same order as k_kstar + k_mean_reduce would give with 1024 threads -- NOT the same bits as the
// batched path (256 threads per block there): a single point is a different summation tree.
__global__ __launch_bounds__(1024) void k_mean_direct(const double* __restrict__ Xs, int64_t N, int d, XArg xa,
                                                      const double* __restrict__ invell, int kid, double rho,
                                                      const double* __restrict__ alpha, double bias,
                                                      double* __restrict__ out) {
    __shared__ double sh[16];
    const int j = blockIdx.x;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 1024) {
        double r2 = 0.0;
        for (int k = 0; k < d; ++k) {
            const double df = Xs[i * d + k] - xa.v[k] * invell[k];
            r2 = fma(df, df, r2);
        }
        double kv, gv;
        kern_and_grad(kid, r2, rho, kv, gv);
        if (j == d) acc = fma(kv, alpha[i], acc);
        else acc = fma(gv * (2.0 * invell[j] * (xa.v[j] * invell[j] - Xs[i * d + j])), alpha[i], acc);
    }
    acc = block_sum16(acc, sh);
    if (threadIdx.x == 0) out[j == d ? 0 : 2 + j] = (j == d) ? bias + acc : acc;
}

// Which form a predict-with-gradients call of M points takes (option grad_form): 0 = auto -- a single point (the
// reference's single-seed refinement) goes the one-pass way when its 1 + d right-hand sides fit one pass; 1 = always two
// passes (batch-independent: the lock-step refinement pins this); 2 = one pass whenever d allows.
static bool grad_one_pass(const gpx_handle* h, int64_t M) {
    if (1 + h->d > GB || h->grad_form == 1) return false;
    return h->grad_form == 2 || M == 1;
}
static int grad_chunk(const gpx_handle* h, bool one_pass) { return one_pass ? GB / (1 + (int)h->d) : GB; }

// rb_call: the two-pass kernel of the whole CALL (decided once from the call's M, not per chunk: a row's value must not
// depend on whether it travels in a chunk of one -- M % 16 == 1 -- or of several)
static int predict_grad_enqueue(gpx_handle* h, const double* Xc, int mb, bool mean_only = false, bool one_pass = false, bool rb_call = false) {
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GPX_EHIP; }
    hipStream_t s = h->stream;
    const int64_t Np = h->Np, N = h->N;
    const int d = (int)h->d;
    const int64_t per = 2 + 2 * d;
    // scratch: [Xc GB*d][ks][g][V][w] (GB*Np each) [out GB*per] [part nseg*GB*Np]
    const int nseg = rb_nseg(h, N);
    // sized for the factor's CAPACITY, not its current size: when an append adds a 128-block (Np grows inside cap_np) the
    // scratch must not be re-allocated -- hipFree waits for the whole device, i.e. for the announced correction pass on the
    // third stream (measured: the recommender's first call after a growth stalled 7.5 ms)
    const int64_t Ncap = std::max<int64_t>(Np, h->cap_np);
    const int64_t need = GB * d + 4 * GB * Ncap + GB * per + (int64_t)rb_nseg(h, Ncap) * GB * Ncap;
    if (need > h->cap_grad) {
        if (h->dgrad) hipFree(h->dgrad);
        h->dgrad = nullptr;
        h->cap_grad = 0;
        if (hipMalloc((void**)&h->dgrad, (size_t)need * 8) != hipSuccess) {
            h->err = "predict: device allocation failed";
            return GPX_EOOM;
        }
        h->cap_grad = need;
    }
    if (GB * per > h->cap_hpin) {     // pinned: a pageable destination would make the download synchronous
        if (h->hpin) hipHostFree(h->hpin);
        h->hpin = nullptr;
        h->cap_hpin = 0;
        if (hipHostMalloc((void**)&h->hpin, (size_t)GB * per * 8, hipHostMallocMapped) != hipSuccess) {
            h->err = "predict: pinned host allocation failed";
            return GPX_EOOM;
        }
        h->cap_hpin = GB * per;
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, h->hpin, 0) != hipSuccess || !dp) {
            h->err = "predict: pinned host buffer is not mapped into the device";
            return GPX_EHIP;
        }
        h->hpin_dev = (double*)dp;
    }
    double* dX = h->dgrad;
    double* dks = dX + GB * d;
    double* dg = dks + GB * Np;
    double* dV = dg + GB * Np;
    double* dw = dV + GB * Np;
    double* dout = dw + GB * Np;
    double* dpart = dout + GB * per;
    const unsigned rows4 = (unsigned)((Np + 3) / 4);
    if (one_pass && !mean_only) {
        // [ks] holds the mb (1 + d) right-hand sides, one pass over T, U untouched.  A single point travels in the kernel
        // arguments and its results are written straight into the pinned host buffer (no copies on the stream).
        const int nrhs = mb * (1 + d);
        const bool direct = (mb == 1);
        XArg xa;
        if (direct) {
            for (int k = 0; k < d; ++k) xa.v[k] = Xc[k];
        } else if (hipMemcpyAsync(dX, Xc, (size_t)mb * d * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
            h->err = "predict: H2D copy failed";
            return GPX_EHIP;
        }
        hipLaunchKernelGGL(k_kstar_d, dim3((unsigned)((Np + 255) / 256), (unsigned)(1 + d), (unsigned)mb), dim3(256), 0, s,
                           h->dXs, N, Np, d, direct ? (const double*)nullptr : dX, xa, h->dinvell, h->kernel_id, h->rho, dks);
        launch_tri_matvec_rb(h, s, h->dT, Np, N, dks, nrhs, 0, dpart);
        hipLaunchKernelGGL(k_grad_reduce_1p, dim3((unsigned)(d + 1), (unsigned)mb), dim3(1024), 0, s, N, Np, d, dks, dpart,
                           nseg, nrhs, h->da, h->dalpha, h->rho, h->bias, direct ? h->hpin_dev : dout);
        if (!direct && hipMemcpyAsync(h->hpin, dout, (size_t)mb * per * 8, hipMemcpyDeviceToHost, s) != hipSuccess) {
            h->err = "predict: D2H copy failed";
            return GPX_EHIP;
        }
        return GPX_OK;
    }
    if (mean_only && one_pass && mb == 1) {      // (decided per CALL by the caller: the last row of a 17-row batch is not 'a single point')
        XArg xa;
        for (int k = 0; k < d; ++k) xa.v[k] = Xc[k];
        hipLaunchKernelGGL(k_mean_direct, dim3((unsigned)(d + 1)), dim3(1024), 0, s, h->dXs, N, d, xa, h->dinvell,
                           h->kernel_id, h->rho, h->dalpha, h->bias, h->hpin_dev);
        return GPX_OK;
    }
    if (hipMemcpyAsync(dX, Xc, (size_t)mb * d * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
        h->err = "predict: H2D copy failed";
        return GPX_EHIP;
    }
    hipLaunchKernelGGL(k_kstar, dim3((unsigned)((Np + 255) / 256), (unsigned)mb), dim3(256), 0, s, h->dXs,
                       N, Np, d, dX, h->dinvell, h->kernel_id, h->rho, dks, dg);
    if (mean_only) {
        hipLaunchKernelGGL(k_mean_reduce, dim3((unsigned)(d + 1), (unsigned)mb), dim3(256), 0, s, h->dXs, N, Np, d,
                           dX, h->dinvell, dks, dg, h->dalpha, h->bias, dout);
    } else {
        if (h->grad_kernel == 1 || (h->grad_kernel < 0 && rb_call)) {
            const dim3 gs((unsigned)((Np + 255) / 256), (unsigned)mb);
            launch_tri_matvec_rb(h, s, h->dT, Np, N, dks, mb, 0, dpart);
            hipLaunchKernelGGL(k_part_sum, gs, dim3(256), 0, s, dpart, Np, mb, nseg, dV);
            launch_tri_matvec_rb(h, s, h->dU, Np, N, dV, mb, 1, dpart);
            hipLaunchKernelGGL(k_part_sum, gs, dim3(256), 0, s, dpart, Np, mb, nseg, dw);
        } else {
            launch_tri_matvec_multi(s, rows4, h->dT, Np, N, dks, mb, 0, dV);
            launch_tri_matvec_multi(s, rows4, h->dU, Np, N, dV, mb, 1, dw);
        }
        hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)(d + 1), (unsigned)mb), dim3(256), 0, s, h->dXs, N, Np,
                           d, dX, h->dinvell, dg, dV, dw, h->da, h->dalpha, h->rho, h->bias, dout);
    }
    if (hipMemcpyAsync(h->hpin, dout, (size_t)mb * per * 8, hipMemcpyDeviceToHost, s) != hipSuccess) {
        h->err = "predict: D2H copy failed";
        return GPX_EHIP;
    }
    return GPX_OK;
}

// wait for the chunk and scatter it into the caller's arrays
static int predict_grad_collect(gpx_handle* h, int mb, double* mu, double* s2, double* dmu, double* ds2) {
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        h->err = "predict: kernel launch or copy failed";
        return GPX_EHIP;
    }
    const int d = (int)h->d;
    const int64_t per = 2 + 2 * d;
    for (int m = 0; m < mb; ++m) {
        const double* o = h->hpin + (size_t)m * per;
        mu[m] = o[0];
        if (s2) s2[m] = o[1];
        for (int j = 0; j < d; ++j) {
            if (dmu) dmu[(int64_t)m * d + j] = o[2 + j];
            if (ds2) ds2[(int64_t)m * d + j] = o[2 + d + j];
        }
    }
    return GPX_OK;
}

int predict_grad_host(gpx_handle* h, const double* Xc, int64_t M, double* mu, double* s2, double* dmu,
                      double* ds2) {
    if (!h->fitted) { h->err = "predict: model is not fitted"; return GPX_ESTATE; }
    if (!Xc || M < 1) { h->err = "predict: need M >= 1 points"; return GPX_EARG; }
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GPX_EHIP; }
    if (int rc0 = ensure_inverse(h)) return rc0;
    const int d = (int)h->d;
    const bool op = grad_one_pass(h, M);
    const int cs = grad_chunk(h, op);
    for (int64_t m0 = 0; m0 < M; m0 += cs) {
        const int mb = (int)std::min<int64_t>(cs, M - m0);
        if (int rc = predict_grad_enqueue(h, Xc + m0 * d, mb, false, op, M > 1)) return rc;
        if (int rc = predict_grad_collect(h, mb, mu + m0, s2 + m0, dmu + m0 * d, ds2 + m0 * d)) return rc;
    }
    return GPX_OK;
}

// Mean (and its gradient) only: k(x, X).alpha -- no pass over T or U per call (alpha itself comes with the lazily built
// inverse, once per fit).
int predict_mean_host(gpx_handle* h, const double* Xc, int64_t M, double* mu, double* dmu) {
    if (!h->fitted) { h->err = "predict_mean: model is not fitted"; return GPX_ESTATE; }
    if (!Xc || !mu || M < 1) { h->err = "predict_mean: need M >= 1 points and a mu output"; return GPX_EARG; }
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GPX_EHIP; }
    if (int rc0 = ensure_inverse(h)) return rc0;
    const int d = (int)h->d;
    for (int64_t m0 = 0; m0 < M; m0 += GB) {
        const int mb = (int)std::min<int64_t>(GB, M - m0);
        if (int rc = predict_grad_enqueue(h, Xc + m0 * d, mb, true, M == 1 && d <= GB && h->grad_form != 1)) return rc;
        if (int rc = predict_grad_collect(h, mb, mu + m0, nullptr, dmu ? dmu + m0 * d : nullptr, nullptr)) return rc;
    }
    return GPX_OK;
}

// The same for every member of a hyper-parameter ensemble: outputs (n, M) / (n, M, d), member-major.  Each chunk is
// enqueued on ALL members' streams before the first wait, so the members' latency-bound kernels overlap.
int ensemble_predict_grad_host(gpx_handle* const* mem, int n, const double* Xc, int64_t M, double* mu, double* s2,
                               double* dmu, double* ds2) {
    gpx_handle* h0 = mem[0];
    if (!Xc || M < 1) { h0->err = "ensemble_predict: need M >= 1 points"; return GPX_EARG; }
    const int d = (int)h0->d;
    for (int i = 0; i < n; ++i) {
        if (!mem[i]->fitted) { h0->err = "ensemble_predict: a member is not fitted"; return GPX_ESTATE; }
        if (int rc0 = ensure_inverse(mem[i])) { if (mem[i] != h0) h0->err = mem[i]->err; return rc0; }
    }
    const bool op = grad_one_pass(h0, M);
    const int cs = grad_chunk(h0, op);
    for (int64_t m0 = 0; m0 < M; m0 += cs) {
        const int mb = (int)std::min<int64_t>(cs, M - m0);
        for (int i = 0; i < n; ++i)
            if (int rc = predict_grad_enqueue(mem[i], Xc + m0 * d, mb, false, op, M > 1)) { if (mem[i] != h0) h0->err = mem[i]->err; return rc; }
        for (int i = 0; i < n; ++i)
            if (int rc = predict_grad_collect(mem[i], mb, mu + i * M + m0, s2 + i * M + m0, dmu + (i * M + m0) * d,
                                              ds2 + (i * M + m0) * d)) { if (mem[i] != h0) h0->err = mem[i]->err; return rc; }
    }
    return GPX_OK;
}

// ------------------------------------------------------------------------------------------------
// Incremental fit: absorb ONE new observation (x, y) into an existing factorisation without the O(N^3)
// refit -- what `model.add_data(x, y)` needs on every iteration of the BO loop (pybo/bayesopt.py:269).
// With L = R^T, T = L^-1, U = T^T and k = k(X, x):
//     r  = T k                          d^2 = rho + sn2 - r.r
//     L' = [L 0; r^T d]                 T'  = [T 0; t^T 1/d],   t = -(U r)/d
//     a' = [a; (y - bias - r.a)/d]      alpha' = [alpha + t a_N; a_N/d]
// i.e. two triangular matvecs (one pass over T, one over U: HBM-read bound, ~0.54 GB at N = 8192) and a
// scatter of one row/column.  The new row lands in the identity padding, so it only applies while
// N < Np; the caller refits when a 128-block boundary is crossed.
// ------------------------------------------------------------------------------------------------
// scal: [0] d  [1] 1/d  [2] a_N  [3] d^2
__global__ __launch_bounds__(256) void k_append_dots(const double* __restrict__ r, const double* __restrict__ a,
                                                     int64_t N, double kss, double resid,
                                                     double* __restrict__ scal, int* __restrict__ flag) {
    __shared__ double sh[4];
    double rr = 0.0, ra = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        const double v = r[i];
        rr = fma(v, v, rr);
        ra = fma(v, a[i], ra);
    }
    rr = block_sum(rr, sh);
    ra = block_sum(ra, sh);
    if (threadIdx.x == 0) {
        const double d2 = kss - rr;
        if (!(d2 > 0.0) || !(d2 < 1.0e300)) {
            *flag = (int)N + 1;
            scal[0] = 1.0; scal[1] = 1.0; scal[2] = 0.0; scal[3] = d2;
        } else {
            const double dd = sqrt(d2);
            scal[0] = dd;
            scal[1] = 1.0 / dd;
            scal[2] = (resid - ra) / dd;
            scal[3] = d2;
        }
    }
}

__global__ __launch_bounds__(256) void k_append_scatter(double* __restrict__ R, double* __restrict__ T,
                                                        double* __restrict__ U, int64_t Np, int64_t N,
                                                        const double* __restrict__ r,
                                                        const double* __restrict__ tu,
                                                        const double* __restrict__ scal, double* __restrict__ a,
                                                        double* __restrict__ alpha, double* __restrict__ y,
                                                        double ynew, double* __restrict__ Xs,
                                                        double* __restrict__ Xraw, int d,
                                                        const double* __restrict__ xnew,
                                                        const double* __restrict__ invell,
                                                        const int* __restrict__ flag) {
    if (*flag != 0) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > N) return;
    const double dd = scal[0], dinv = scal[1], aN = scal[2];
    if (i < N) {
        const double t = -tu[i] * dinv;
        R[i * Np + N] = r[i];
        T[N * Np + i] = t;
        U[i * Np + N] = t;
        alpha[i] = fma(t, aN, alpha[i]);
    } else {
        R[N * Np + N] = dd;
        T[N * Np + N] = dinv;
        U[N * Np + N] = dinv;
        a[N] = aN;
        alpha[N] = aN * dinv;
        y[N] = ynew;
        for (int k = 0; k < d; ++k) {
            Xraw[N * d + k] = xnew[k];
            Xs[N * d + k] = xnew[k] * invell[k];
        }
    }
}

// R, T, U get one more identity-padded 128-block (Np -> Np + 128) when an append finds the current padding
// used up: a re-strided device copy of the three factors (3 x 8 Np^2 bytes, ~1 ms at N = 8192) instead of the
// O(N^3) refit round 1 fell back to at every 128th observation.
__global__ void k_identity_tail(double* __restrict__ R, double* __restrict__ T, double* __restrict__ U, int64_t ld,
                                int64_t lo, int64_t hi) {
    const int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    R[i * ld + i] = 1.0;
    T[i * ld + i] = 1.0;
    U[i * ld + i] = 1.0;
}

static int grow_block(gpx_handle* h) {
    const int64_t Np = h->Np, Nn = Np + NB;
    hipStream_t s = h->stream;
    if (Nn <= h->cap_np) {
        // The buffers were allocated with head-room (alloc_model): only the leading dimension changes.  The three
        // factors are re-strided OUT OF PLACE by rotating through the workspace -- R -> S's buffer, T -> R's old
        // one, U -> T's old one, U's old buffer becomes the workspace -- all in stream order, no allocation and no
        // host synchronisation (a fresh allocation of 4 x 0.55 GB was measured at up to 60 ms on this path).
        double* buf[4] = {h->dS, h->dR, h->dT, h->dU};          // destination i receives source i + 1
        bool ok = true;
        for (int i = 0; i < 3 && ok; ++i) {
            ok = hipMemsetAsync(buf[i], 0, (size_t)Nn * Nn * 8, s) == hipSuccess &&
                 hipMemcpy2DAsync(buf[i], (size_t)Nn * 8, buf[i + 1], (size_t)Np * 8, (size_t)Np * 8, (size_t)Np,
                                  hipMemcpyDeviceToDevice, s) == hipSuccess;
        }
        // padding of the new block: zero rows of the scaled inputs and of y / a / alpha (they sit past the old Np)
        ok = ok && hipMemsetAsync(h->dXs + Np * h->d, 0, (size_t)NB * h->d * 8, s) == hipSuccess &&
             hipMemsetAsync(h->dy + Np, 0, (size_t)NB * 8, s) == hipSuccess &&
             hipMemsetAsync(h->da + Np, 0, (size_t)NB * 8, s) == hipSuccess &&
             hipMemsetAsync(h->dalpha + Np, 0, (size_t)NB * 8, s) == hipSuccess;
        if (!ok) { h->err = "append: growing the factor failed"; return GPX_EHIP; }
        hipLaunchKernelGGL(k_identity_tail, dim3(1), dim3(NB), 0, s, buf[0], buf[1], buf[2], Nn, Np, Nn);
        h->dR = buf[0]; h->dT = buf[1]; h->dU = buf[2]; h->dS = buf[3];
        h->Np = Nn;
        return GPX_OK;
    }
    const int64_t cap = Nn + std::max<int64_t>(NB, Nn / 16 / NB * NB);      // head-room for the next growths
    double* nm[4] = {nullptr, nullptr, nullptr, nullptr};     // S (workspace), R, T, U
    double* nv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // y, a, alpha, Xs, Xraw
    const size_t vbytes[5] = {(size_t)cap * 8, (size_t)cap * 8, (size_t)cap * 8, (size_t)cap * h->cap_d * 8,
                              (size_t)cap * h->cap_d * 8};
    auto bail = [&](const char* msg, int code) {
        for (double* p : nm) if (p) hipFree(p);
        for (double* p : nv) if (p) hipFree(p);
        h->err = msg;
        return code;
    };
    for (int i = 0; i < 4; ++i) {
        if (hipMalloc((void**)&nm[i], (size_t)cap * cap * 8) != hipSuccess) return bail("append: device allocation failed", GPX_EOOM);
        if (hipMemsetAsync(nm[i], 0, (size_t)Nn * Nn * 8, s) != hipSuccess) return bail("append: memset failed", GPX_EHIP);
    }
    for (int i = 0; i < 5; ++i) {
        if (hipMalloc((void**)&nv[i], vbytes[i]) != hipSuccess) return bail("append: device allocation failed", GPX_EOOM);
        if (hipMemsetAsync(nv[i], 0, vbytes[i], s) != hipSuccess) return bail("append: memset failed", GPX_EHIP);
    }
    double* om[4] = {h->dS, h->dR, h->dT, h->dU};
    for (int i = 1; i < 4; ++i)
        if (hipMemcpy2DAsync(nm[i], (size_t)Nn * 8, om[i], (size_t)Np * 8, (size_t)Np * 8, (size_t)Np,
                             hipMemcpyDeviceToDevice, s) != hipSuccess)
            return bail("append: device copy failed", GPX_EHIP);
    hipLaunchKernelGGL(k_identity_tail, dim3(1), dim3(NB), 0, s, nm[1], nm[2], nm[3], Nn, Np, Nn);
    double* ov[5] = {h->dy, h->da, h->dalpha, h->dXs, h->dXraw};
    const size_t cbytes[5] = {(size_t)Np * 8, (size_t)Np * 8, (size_t)Np * 8, (size_t)Np * h->d * 8,
                              (size_t)h->N * h->d * 8};
    for (int i = 0; i < 5; ++i)
        if (hipMemcpyAsync(nv[i], ov[i], cbytes[i], hipMemcpyDeviceToDevice, s) != hipSuccess)
            return bail("append: device copy failed", GPX_EHIP);
    if (hipStreamSynchronize(s) != hipSuccess) return bail("append: growing the factor failed", GPX_EHIP);
    for (double* p : om) hipFree(p);
    for (double* p : ov) hipFree(p);
    h->dS = nm[0]; h->dR = nm[1]; h->dT = nm[2]; h->dU = nm[3];
    h->dy = nv[0]; h->da = nv[1]; h->dalpha = nv[2]; h->dXs = nv[3]; h->dXraw = nv[4];
    h->Np = Nn;
    h->cap_np = cap;
    return GPX_OK;
}

// for gpx_append_begin: an announced point at a block boundary grows the factor first, like the append itself would
int grow_factor_if_full(gpx_handle* h) { return h->N >= h->Np ? grow_block(h) : GPX_OK; }

void launch_append_prepare(gpx_handle* h, hipStream_t s, const double* dx, double* dks, double* dg, double* dr,
                           double* dtu, double resid, double* scal, int* flag) {
    const int64_t Np = h->Np, N = h->N;
    const unsigned rows4 = (unsigned)((Np + 3) / 4);
    hipLaunchKernelGGL(k_kstar, dim3((unsigned)((Np + 255) / 256), 1), dim3(256), 0, s, h->dXs, N, Np, (int)h->d, dx,
                       h->dinvell, h->kernel_id, h->rho, dks, dg);
    launch_tri_matvec_multi(s, rows4, h->dT, Np, N, dks, 1, 0, dr);
    hipLaunchKernelGGL(k_append_dots, dim3(1), dim3(256), 0, s, dr, h->da, N, h->rho + h->sn2, resid, scal, flag);
    launch_tri_matvec_multi(s, rows4, h->dU, Np, N, dr, 1, 1, dtu);
}

SpecBuf spec_layout(const gpx_handle* h) {
    const int64_t xpad = (h->cap_d + 63) / 64 * 64, ld = h->spec_ld;
    SpecBuf b;
    b.x = h->dspec;
    b.xs = b.x + xpad;
    b.ks = b.xs + xpad;
    b.g = b.ks + ld;
    b.r = b.g + ld;
    b.tu = b.r + ld;
    b.row = b.tu + ld;
    b.pscal = b.row + ld;
    b.scal = b.pscal + 2;
    b.v = b.scal + 16 + 46;          // keeps v 512-byte aligned relative to the base (2 + 16 + 46 = 64 doubles)
    return b;
}

int append_host(gpx_handle* h, const double* x, double ynew) {
    if (!h->fitted) { h->err = "append: model is not fitted"; return GPX_ESTATE; }
    if (!x) { h->err = "append: NULL point"; return GPX_EARG; }
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GPX_EHIP; }
    if (int rc0 = ensure_inverse(h)) return rc0;     // the rank-1 extension updates T, U, a, alpha in place
    if (h->N >= h->Np)                               // padding of the last 128-block used up: add a block
        if (int rc0 = grow_block(h)) return rc0;
    hipStream_t s = h->stream;
    const int64_t Np = h->Np, N = h->N;
    const int d = (int)h->d;
    // scratch: [x d (padded to a multiple of 64)][ks Np][g Np][r Np][tu Np]
    const int64_t xpad = (d + 63) / 64 * 64;
    const int64_t need = xpad + 4 * std::max<int64_t>(Np, h->cap_np);      // (capacity: see predict_grad_enqueue)
    if (need > h->cap_grad) {
        if (h->dgrad) hipFree(h->dgrad);
        h->dgrad = nullptr;
        h->cap_grad = 0;
        if (hipMalloc((void**)&h->dgrad, (size_t)need * 8) != hipSuccess) {
            h->err = "append: device allocation failed";
            return GPX_EOOM;
        }
        h->cap_grad = need;
    }
    double* dx = h->dgrad;
    double* dks = dx + xpad;
    double* dg = dks + Np;
    double* dr = dg + Np;
    double* dtu = dr + Np;
    h->app_w = dtu;                                  // w = U r = K^-1 k(X, x): read by the sweep-cache correction
    if (hipMemsetAsync(h->dflag, 0, sizeof(int), s) != hipSuccess ||
        hipMemcpyAsync(dx, x, (size_t)d * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
        h->err = "append: H2D copy failed";
        return GPX_EHIP;
    }
    // an ANNOUNCED point (gpx_append_begin) whose announcement still describes this model: k*, r, tu and the correction
    // pass of the sweep cache were computed ahead; only the value-dependent scalars and the scatter are left
    h->spec_used = h->spec_active && h->spec_gen == h->gen && (int64_t)h->spec_x.size() == d &&
                   memcmp(h->spec_x.data(), x, (size_t)d * 8) == 0;
    h->spec_active = false;
    if (h->spec_used) {
        const SpecBuf sb = spec_layout(h);
        hipLaunchKernelGGL(k_append_dots, dim3(1), dim3(256), 0, s, sb.r, h->da, N, h->rho + h->sn2, ynew - h->bias,
                           h->dscal, h->dflag);
        dr = sb.r;
        dtu = sb.tu;
        dx = sb.x;
        h->app_w = dtu;
    } else {
        launch_append_prepare(h, s, dx, dks, dg, dr, dtu, ynew - h->bias, h->dscal, h->dflag);
    }
    hipLaunchKernelGGL(k_append_scatter, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, s, h->dR, h->dT,
                       h->dU, Np, N, dr, dtu, h->dscal, h->da, h->dalpha, h->dy, ynew, h->dXs, h->dXraw, d, dx,
                       h->dinvell, h->dflag);
    int flag = 0;
    if (hipMemcpyAsync(&flag, h->dflag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
        h->err = "append: kernel or D2H copy failed";
        return GPX_EHIP;
    }
    if (flag != 0) {
        h->fail_pivot = (int64_t)flag - 1;
        h->err = "append: K + sn2*I is not positive definite with the new point";
        return GPX_ENOTPD;
    }
    h->N = N + 1;
    return GPX_OK;
}

// ------------------------------------------------------------------------------------------------
// log marginal likelihood of the current fit:  -1/2 a.a - sum_i log R_ii - N/2 log(2 pi)
// (R&W eq. 2.30 with K + sn2 I = R^T R, a = R^-T (y - bias)).  One workgroup; the quantities exist already.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loglik(const double* __restrict__ R, int64_t Np, int64_t N,
                                                const double* __restrict__ a, double* __restrict__ out) {
    __shared__ double sh[4];
    double q = 0.0, ld = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        const double v = a[i];
        q = fma(v, v, q);
        ld += log(R[i * Np + i]);
    }
    q = block_sum(q, sh);
    ld = block_sum(ld, sh);
    if (threadIdx.x == 0) out[0] = -0.5 * q - ld - 0.5 * (double)N * 1.83787706640934548356;
}

int loglik_host(gpx_handle* h, double* out) {
    if (!h->fitted) { h->err = "loglik: model is not fitted"; return GPX_ESTATE; }
    if (!out) { h->err = "loglik: NULL output"; return GPX_EARG; }
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GPX_EHIP; }
    if (int rc0 = ensure_inverse(h)) return rc0;     // a = T (y - bias)
    hipLaunchKernelGGL(k_loglik, dim3(1), dim3(256), 0, h->stream, h->dR, h->Np, h->N, h->da, h->dscal + 8);
    if (hipMemcpyAsync(out, h->dscal + 8, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        h->err = "loglik: kernel or D2H copy failed";
        return GPX_EHIP;
    }
    return GPX_OK;
}

// ---- RFF sample value + gradient at M points: grid (M), threads over features -------------------
// out per point: [f, df/dx_0 .. df/dx_{d-1}]
__global__ __launch_bounds__(256) void k_rff_grad(const double* __restrict__ W, const double* __restrict__ b,
                                                  const double* __restrict__ theta, int n, int d, double bias,
                                                  const double* __restrict__ Xc, double* __restrict__ out) {
    __shared__ double sh[4];
    const int m = blockIdx.x;
    const double* x = Xc + (int64_t)m * d;
    double f = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        double z = b[j];
        for (int k = 0; k < d; ++k) z = fma(W[j * d + k], x[k], z);
        f = fma(theta[j], cos(z), f);
    }
    f = block_sum(f, sh);
    if (threadIdx.x == 0) out[(int64_t)m * (d + 1)] = bias + f;
    for (int k = 0; k < d; ++k) {
        double gk = 0.0;
        for (int j = threadIdx.x; j < n; j += 256) {
            double z = b[j];
            for (int kk = 0; kk < d; ++kk) z = fma(W[j * d + kk], x[kk], z);
            gk = fma(-theta[j] * sin(z), W[j * d + k], gk);
        }
        gk = block_sum(gk, sh);
        if (threadIdx.x == 0) out[(int64_t)m * (d + 1) + 1 + k] = gk;
    }
}

int rff_grad_host(gpx_handle* h, const double* W, const double* b, const double* theta, int64_t n, int64_t d,
                  double bias, const double* Xc, int64_t M, double* f, double* g) {
    if (!W || !b || !theta || !Xc || !f || !g || n < 1 || d < 1 || d > DMAX || M < 1) {
        h->err = "rff_grad: bad arguments";
        return GPX_EARG;
    }
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GPX_EHIP; }
    hipStream_t s = h->stream;
    const int64_t need = n * d + 2 * n + M * d + M * (d + 1);
    if (need > h->cap_grad) {
        if (h->dgrad) hipFree(h->dgrad);
        h->dgrad = nullptr;
        h->cap_grad = 0;
        if (hipMalloc((void**)&h->dgrad, (size_t)need * 8) != hipSuccess) {
            h->err = "rff_grad: device allocation failed";
            return GPX_EOOM;
        }
        h->cap_grad = need;
    }
    double* dW = h->dgrad;
    double* db = dW + n * d;
    double* dth = db + n;
    double* dX = dth + n;
    double* dout = dX + M * d;
    bool ok = hipMemcpyAsync(dW, W, (size_t)n * d * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(db, b, (size_t)n * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(dth, theta, (size_t)n * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(dX, Xc, (size_t)M * d * 8, hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) { h->err = "rff_grad: H2D copy failed"; return GPX_EHIP; }
    hipLaunchKernelGGL(k_rff_grad, dim3((unsigned)M), dim3(256), 0, s, dW, db, dth, (int)n, (int)d, bias, dX,
                       dout);
    std::vector<double> host((size_t)M * (d + 1));
    if (hipMemcpyAsync(host.data(), dout, host.size() * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
        h->err = "rff_grad: kernel or D2H copy failed";
        return GPX_EHIP;
    }
    for (int64_t m = 0; m < M; ++m) {
        f[m] = host[(size_t)m * (d + 1)];
        for (int64_t k = 0; k < d; ++k) g[m * d + k] = host[(size_t)m * (d + 1) + 1 + k];
    }
    return GPX_OK;
}

}  // namespace gpx
