// kernels_fit.hip -- GP fit on gfx950:  Gram build, blocked Cholesky (K = R^T R, R upper row-major),
// recursive triangular inversion (T = R^-T lower, U = R^-1 upper), a = T (y - bias), alpha = U a.
//
// Replaces the work behind `model.add_data(X, Y)` [pybo/bayesopt.py:114,258,269]; the arithmetic
// itself lives in the un-vendored `reggie` package, so it is restated from Rasmussen & Williams
// Alg. 2.1 (oracle/gp_ref.py is the CPU statement of the same maths).
//
// Storage: everything is (Np, Np) row-major with Np = N rounded up to 128; padding rows/columns carry
// the identity so every tile is full and no kernel has bounds checks.  We factor the UPPER triangle:
// with R row-major, both GEMM operands of the trailing update  S_IJ -= R_pI^T R_pJ  are k-major
// (see gemm_core.h), which is what keeps the loads coalesced without any transpose.
#include <math.h>

#include <vector>
#include <algorithm>

#include <utility>

#include <cstring>

#include "gemm_core.h"
#include "gpx_internal.h"
#include "gpx_math.h"
#include "fit_tiles.h"

namespace gpx {

// (covariance functions: gpx_math.h)
// Xs[i][k] = X[i][k] / ell[k] for i < n, 0 for n <= i < np
// Batched use (gpx_loglik_batch): blockIdx.z = batch element with its own 1/ell (DMAX apart) and its own Xs.
__global__ void k_scale_x(const double* __restrict__ X, int64_t n, int64_t np, int d,
                          const double* __restrict__ invell, double* __restrict__ Xs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= np * d) return;
    invell += (int64_t)blockIdx.z * DMAX;
    Xs += (int64_t)blockIdx.z * np * d;
    const int64_t i = idx / d;
    const int k = (int)(idx - i * d);
    Xs[idx] = (i < n) ? X[idx] * invell[k] : 0.0;
}

void launch_scale_x(hipStream_t s, const double* X, int64_t n, int64_t np, int d, const double* invell,
                    double* Xs) {
    const int64_t tot = np * d;
    hipLaunchKernelGGL(k_scale_x, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, X, n, np, d,
                       invell, Xs);
}

// ------------------------------------------------------------------------------------------------
// R1: symmetric Gram build, upper 128-block triangle, 64x64 tile per workgroup, 4x4 per thread.
// HBM-write bound: 8 B per element, X tiles staged in LDS (coordinate-major so a wave's lanes read
// consecutive addresses).
// ------------------------------------------------------------------------------------------------
constexpr int GT = 64;
constexpr int GDC = 16;  // coordinates staged per pass

// Batched use: blockIdx.z = batch element; hyp (B, 3) = {rho, sn2, bias} per element overrides the scalars.
__global__ __launch_bounds__(256) void k_gram_sym(const double* __restrict__ Xs, int64_t N, int64_t Np,
                                                  int d, int kid, double rho, double sn2,
                                                  double* __restrict__ S, const double* __restrict__ hyp) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if ((bi >> 1) > (bj >> 1)) return;  // below the 128-block diagonal: never read
    if (hyp) {
        rho = hyp[3 * blockIdx.z];
        sn2 = hyp[3 * blockIdx.z + 1];
        Xs += (int64_t)blockIdx.z * Np * d;
        S += (int64_t)blockIdx.z * Np * Np;
    }
    __shared__ double xi[GDC][GT];
    __shared__ double xj[GDC][GT];
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const int64_t i0 = (int64_t)bi * GT, j0 = (int64_t)bj * GT;
    double r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;

    for (int k0 = 0; k0 < d; k0 += GDC) {
        const int kc = min(GDC, d - k0);
        __syncthreads();
        for (int e = t; e < GT * kc; e += 256) {
            const int row = e / kc, k = e - row * kc;
            xi[k][row] = Xs[(i0 + row) * d + k0 + k];
            xj[k][row] = Xs[(j0 + row) * d + k0 + k];
        }
        __syncthreads();
        for (int k = 0; k < kc; ++k) {
            double a4[4], b4[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) a4[a] = xi[k][ty * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) b4[b] = xj[k][tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double df = a4[a] - b4[b];
                    r2[a][b] = fma(df, df, r2[a][b]);
                }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t gi = i0 + ty * 4 + a;
        d4 o;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t gj = j0 + tx * 4 + b;
            double v;
            if (gi < N && gj < N) {
                v = kern_eval(kid, r2[a][b], rho);
                if (gi == gj) v += sn2;
            } else {
                v = (gi == gj) ? 1.0 : 0.0;
            }
            o[b] = v;
        }
        *reinterpret_cast<d4*>(S + gi * Np + j0 + tx * 4) = o;
    }
}

void launch_gram_sym(hipStream_t s, const double* Xs, int64_t N, int64_t Np, int d, int kernel_id,
                     double rho, double sn2, double* S) {
    const unsigned g = (unsigned)(Np / GT);
    hipLaunchKernelGGL(k_gram_sym, dim3(g, g), dim3(256), 0, s, Xs, N, Np, d, kernel_id, rho, sn2, S,
                       (const double*)nullptr);
}

// DBG (scripts/potrf_bench.hip only): wall-clock stamps of the phases go to `dbg` (thread 0)
// Batched use (gpx_loglik_batch): blockIdx.z = batch element, S / R / U `bs` elements apart, one flag each;
// T may be NULL (the batched path needs U_d only and keeps it in the dead diagonal blocks of S).
template <bool DBG>
__global__ __launch_bounds__(256) void k_potrf16(const double* __restrict__ S, double* __restrict__ R,
                                                 double* __restrict__ T, double* __restrict__ U, int64_t Np,
                                                 int p, int* __restrict__ flag, long long* __restrict__ dbg,
                                                 int64_t bs) {
    S += (int64_t)blockIdx.z * bs;
    R += (int64_t)blockIdx.z * bs;
    U += (int64_t)blockIdx.z * bs;
    if (T) T += (int64_t)blockIdx.z * bs;
    flag += blockIdx.z;
    __shared__ double Pn[2 * 16 * PFP];   // the 16-row panels of the current and the previous step
    __shared__ double Ud[256];            // the current 16x16 inverse, transposed (k-major A operand)
    __shared__ int sflag;
    potrf16_body<DBG>(S, R, T, U, Np, p, flag, dbg, Pn, Ud, sflag);
}

// ------------------------------------------------------------------------------------------------
// Triangular inverse of one (or many: blockIdx.x) diagonal 128-block(s) from its factor R_pp and the eight
// 16x16 inverses k_potrf16 left in the diagonal tiles of T / U: recursive doubling over 16-tiles
//     T_21 = - T_22 (L_21 T_11),   L_21 = R_12^T
// on one LDS image that holds R (upper tiles, consumed level by level), T (lower tiles) and U = T^T (upper
// tiles, written over the R tiles a level has consumed); every MFMA operand is a k-major read of that image.
// Off the Cholesky's critical path: all diagonal blocks are inverted by ONE launch after the factorisation.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trtri_diag128(const double* __restrict__ R, double* __restrict__ T,
                                                       double* __restrict__ U, int64_t Np, int pbase,
                                                       const int* __restrict__ flag) {
    __shared__ double Mp[NB * PFP];
    __shared__ double Ui[8 * 256];
    if (*flag != 0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int64_t p0 = (int64_t)(pbase + blockIdx.x) * NB;
    {
        // all loads first (16 + 2 x 32 bytes per thread in flight), then the LDS stores
        d4 v[16], u[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx4 = t + 256 * q;
            const int r = idx4 >> 5, c = (idx4 & 31) * 4;
            const int tr = r >> 4, tc = c >> 4;
            v[q] = (d4){0.0, 0.0, 0.0, 0.0};
            if (tc >= tr)      // R above the diagonal tiles, T_d (row-major, zeros above its diagonal) on them
                v[q] = *reinterpret_cast<const d4*>(((tc > tr) ? R : T) + (p0 + r) * Np + p0 + c);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {       // the diagonal tiles of U: Ui[jb][k][m] = U_d[k][m]
            const int idx4 = t + 256 * q;
            const int jb = idx4 >> 6, k = (idx4 >> 2) & 15, m = (idx4 & 3) * 4;
            u[q] = *reinterpret_cast<const d4*>(U + (p0 + 16 * jb + k) * Np + p0 + 16 * jb + m);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx4 = t + 256 * q;
            *reinterpret_cast<d4*>(Mp + (idx4 >> 5) * PFP + (idx4 & 31) * 4) = v[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<d4*>(Ui + (t + 256 * q) * 4) = u[q];
    }
    __syncthreads();
#pragma unroll 1
    for (int hb = 1; hb < 8; hb *= 2) {
        const int ntile = 4 * hb;                     // (8 / 2hb) groups x hb^2 tiles
        d4 acc[4];
        // phase 1: W = L21 T11
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tt = w + 4 * s;
            acc[s] = (d4){0.0, 0.0, 0.0, 0.0};
            if (tt < ntile) {
                const int grp = tt / (hb * hb), rem = tt - grp * hb * hb;
                const int g0 = grp * 2 * hb, i = g0 + hb + rem / hb, j = g0 + rem % hb;
                for (int k = j; k < g0 + hb; ++k) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double a = Mp[(16 * k + 4 * kk + g) * PFP + 16 * i + n];    // R[k-tile rows][i-tile cols]
                        const double b = Mp[(16 * k + 4 * kk + g) * PFP + 16 * j + n];    // T11 tile (k, j), k >= j
                        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[s], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tt = w + 4 * s;
            if (tt < ntile) {
                const int grp = tt / (hb * hb), rem = tt - grp * hb * hb;
                const int g0 = grp * 2 * hb, i = g0 + hb + rem / hb, j = g0 + rem % hb;
#pragma unroll
                for (int r = 0; r < 4; ++r) Mp[(16 * i + g + 4 * r) * PFP + 16 * j + n] = acc[s][r];   // W in T21's place
            }
        }
        __syncthreads();
        // phase 2: T21 = - T22 W
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tt = w + 4 * s;
            acc[s] = (d4){0.0, 0.0, 0.0, 0.0};
            if (tt < ntile) {
                const int grp = tt / (hb * hb), rem = tt - grp * hb * hb;
                const int g0 = grp * 2 * hb, i = g0 + hb + rem / hb, j = g0 + rem % hb;
                for (int k = g0 + hb; k <= i; ++k) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double a = (k == i) ? Ui[i * 256 + (4 * kk + g) * 16 + n]                 // T_d of tile i
                                                  : Mp[(16 * k + 4 * kk + g) * PFP + 16 * i + n];       // U tile (k, i)
                        const double b = Mp[(16 * k + 4 * kk + g) * PFP + 16 * j + n];                  // W tile (k, j)
                        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[s], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();            // everyone has read W and the R12 tiles of this level
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tt = w + 4 * s;
            if (tt < ntile) {
                const int grp = tt / (hb * hb), rem = tt - grp * hb * hb;
                const int g0 = grp * 2 * hb, i = g0 + hb + rem / hb, j = g0 + rem % hb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = -acc[s][r];
                    Mp[(16 * i + g + 4 * r) * PFP + 16 * j + n] = v;        // T21
                    Mp[(16 * j + n) * PFP + 16 * i + g + 4 * r] = v;        // U12 = T21^T over the consumed R12
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx4 = t + 256 * q;
        const int r = idx4 >> 5, c = (idx4 & 31) * 4;
        const int tr = r >> 4, tc = c >> 4;
        const d4 m = *reinterpret_cast<const d4*>(Mp + r * PFP + c);
        const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
        d4 uv = zero;
        if (tr < tc) uv = m;
        else if (tr == tc) uv = *reinterpret_cast<const d4*>(Ui + tr * 256 + (r & 15) * 16 + (c & 15));
        const int64_t gidx = (p0 + r) * Np + p0 + c;
        *reinterpret_cast<d4*>(T + gidx) = (tr >= tc) ? m : zero;
        *reinterpret_cast<d4*>(U + gidx) = uv;
    }
}

__global__ __launch_bounds__(256) void k_panel_solve16(const double* __restrict__ U, const double* __restrict__ S,
                                                       double* __restrict__ R, int64_t Np, int p,
                                                       const int* __restrict__ flag, int64_t bs) {
    U += (int64_t)blockIdx.z * bs;       // batched use: blockIdx.z = batch element
    S += (int64_t)blockIdx.z * bs;
    R += (int64_t)blockIdx.z * bs;
    flag += blockIdx.z;
    if (*flag != 0) return;
    __builtin_amdgcn_s_setprio(3);                // chain kernel: see k_potrf16
    panel_solve16_body(U, S, R, Np, p, (int)blockIdx.x);
}

// Diagonal block AND panel solve in ONE launch (option chol_fuse): every workgroup of the panel solve factors the
// 128 x 128 diagonal block ITSELF (the same instructions on the same inputs: every copy writes the same bits to R_pp
// and to the 16 x 16 inverses, then reads its own back through L2) and goes on with its 64 columns.  What this buys is
// one launch-to-start delay per 128-block instead of two: next to the trailing updates k_potrf16 runs in 29.6 us once
// it has started but takes 133 us per launch at N = 16384 (profiles/history/r03_chol_parts.txt, section 6), and the panel
// solve 63 instead of 11.5.  The redundant factorisations are ~30 us of ONE workgroup slot each -- next to a chip
// full of 175-us trailing tiles.
__global__ __launch_bounds__(256) void k_potrf_solve16(const double* __restrict__ S, double* __restrict__ R,
                                                       double* __restrict__ T, double* __restrict__ U, int64_t Np,
                                                       int p, int* __restrict__ flag) {
    __shared__ double Pn[2 * 16 * PFP];
    __shared__ double Ud[256];
    __shared__ int sflag;
    potrf16_body<false>(S, R, T, U, Np, p, flag, nullptr, Pn, Ud, sflag);
    __threadfence();                      // this workgroup's copy of R_pp / U_d is in L2 before it is read back
    __syncthreads();
    if (sflag) return;
    panel_solve16_body(U, S, R, Np, p, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// R2c: symmetric update  S_IJ -= sum_{k in [kb0,kb1)} R_kI^T R_kJ  for tiles I = ib0+by <= J = ib0+bx
// (the syrk/gemm on fp64 MFMA).  Used two ways by the two-level blocked factorisation below:
//   * in-panel "row update" (grid (nP-I, 1), ib0 = I): brings block row I up to date with the rows of the
//     current outer panel that are already factored (left-looking inside the panel, K = up to (W-1)*128);
//   * trailing update after an outer panel of W block rows (grid (t, t)): K = W*128 per pass, so every
//     S tile is read-modified-written once per W panels instead of once per panel (the K = 128 version
//     is HBM-bound: 8 flop/B against a machine balance of ~12).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_update(const double* __restrict__ R,
                                                                 double* __restrict__ S, int64_t Np,
                                                                 int kb0, int kb1, int ib0, int jb0, int64_t bs) {
    const int I = ib0 + blockIdx.y, J = jb0 + blockIdx.x;
    if (I > J) return;
    R += (int64_t)blockIdx.z * bs;       // batched use: blockIdx.z = batch element
    S += (int64_t)blockIdx.z * bs;
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    syrk_tile(R, S, Np, kb0, kb1, I, J, smem);
}

// The same over the upper triangle of the t x t block region at (ib0, ib0) as a ONE-dimensional grid of exactly
// t (t + 1) / 2 workgroups (row r holds the t - r tiles J = r .. t-1).  Workgroups go to the XCDs round-robin by
// their linear index; in the square grid above (index = row * t + column, the lower half returning at once) a
// region with t = 0 mod 8 sends whole tile COLUMNS to one XCD, and column J holds J + 1 tiles: at t = 56 the
// busiest XCD carried 224 tiles against 175 on the idlest and set the launch's duration (the per-launch rate of the
// trailing updates swung between 33 and 57 TFLOP/s with t mod 8 -- profiles/history/r02_far_update_launches.txt).
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_update_tri(const double* __restrict__ R,
                                                                     double* __restrict__ S, int64_t Np, int kb0,
                                                                     int kb1, int ib0, int t) {
    const int idx = blockIdx.x;
    int r = (int)((2.0 * t + 1.0 - sqrt((2.0 * t + 1.0) * (2.0 * t + 1.0) - 8.0 * idx)) * 0.5);
    if (r < 0) r = 0;
    while (r > 0 && r * t - r * (r - 1) / 2 > idx) --r;             // (the float estimate is off by at most one)
    while ((r + 1) * t - (r + 1) * r / 2 <= idx) ++r;
    const int c = r + idx - (r * t - r * (r - 1) / 2);
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    syrk_tile(R, S, Np, kb0, kb1, ib0 + r, ib0 + c, smem);
}

// Row update on 64x64 tiles: block rows I .. I+nrows-1 (two 64-row halves each) <- rows kb0..kb1-1 of R.
// grid (2*(nP-I), 2*nrows): blockIdx.y = 2 * (row - I) + row half, blockIdx.x = 64-column tile counted from that
// row's diagonal block.  `prio`: wave priority -- 3 on the chain (see k_potrf16), lower for the launches that run
// beside it on the third stream.
__global__ __launch_bounds__(GEMM64_THREADS) void k_row_update64(const double* __restrict__ R,
                                                               double* __restrict__ S, int64_t Np, int kb0,
                                                               int kb1, int I, int64_t bs, int nP, int prio) {
    const int row = I + (int)(blockIdx.y >> 1), half = (int)(blockIdx.y & 1);
    if ((int)blockIdx.x < half || (int)blockIdx.x >= 2 * (nP - row)) return;   // below the diagonal / past the end
    R += (int64_t)blockIdx.z * bs;         // batched use: blockIdx.z = batch element
    S += (int64_t)blockIdx.z * bs;
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    __shared__ __attribute__((aligned(16))) double smem[GEMM64_LDS_F64];
    const int64_t i0 = (int64_t)row * NB + (int64_t)half * T64;
    const int64_t j0 = (int64_t)row * NB + (int64_t)blockIdx.x * T64;
    d4 acc[2];             // start from the S tile, A enters negated: acc = S - A B (no read-modify-write tail)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = S[(i0 + acc_row64(i, r)) * Np + j0 + acc_col64()];
    gemm_tile_64_g<true>(acc, R + i0, Np, R + j0, Np, kb0 * NB, kb1 * NB, smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(i0 + acc_row64(i, r)) * Np + j0 + acc_col64()] = acc[i][r];
}

constexpr int CHOL_W = 4;   // outer panel width in 128-blocks

// DIAGNOSTIC (option "x_bg", scripts/chol_bg.py): a synthetic background load for the chain kernels -- `G` workgroups
// that do nothing but fp64 MFMAs on registers for `iters` rounds (16 MFMAs = 1024 matrix-pipe cycles per round and
// wave), at the far updates' wave priority, holding LDSKB kilobytes of LDS so that 2 (72) or 1 (100) of them fit on a
// compute unit.  Separates the two ways the trailing updates can slow the chain down: taking its CU SLOTS (not with
// this kernel: G <= the slots it leaves free) and sharing the DOUBLE-PRECISION PIPE of the SIMDs it runs on.
template <int LDSKB>
__global__ __launch_bounds__(256) void k_bg_mfma(int iters, double* __restrict__ sink) {
    __shared__ double pad[LDSKB * 128];
    d4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    const double a = 1e-9 * threadIdx.x, b = 1.0 + 1e-9 * blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += acc[j][0] + acc[j][3];
    if (threadIdx.x == 0) pad[blockIdx.x & 127] = t;
    __syncthreads();
    if (t == 123456.789 && pad[threadIdx.x & 127] == 1.0) sink[0] = t;     // never true: keeps the work alive
}

// Two-level blocked right-looking factorisation with lookahead.  U(Q, P) = update of panel Q's block rows
// with the factored rows of panel P (K = W*128); chain(Q) needs every U(Q, P < Q).
//   chain(P)   rows P0..P1-1 one by one: row update (left-looking inside the panel, 64x64 tiles) -> k_potrf16 ->
//              k_panel_solve16.  Serial and latency-bound (~85-100 us per 128-block), uses a handful of CUs.
//   near(P)    U(P+1, P), row by row on 64x64 tiles: its first block row on the main stream (the next diagonal
//              block needs it), the other rows in one launch on a third stream while the chain works on that row.
//   mid(P)     U(P+2, P) on a fourth (low-priority) stream, after rest(P-1), which wrote the same rows; near(P+1)
//              touches those rows next and waits for it.
//   rest(P)    U(Q >= P+3, P): the bulk, back-to-back on the (low-priority) side stream, concurrently with mid(P)
//              (disjoint block rows); it only has to be finished before mid(P+1) and rest(P+1).
// So the critical path never waits for a whole trailing update, only for the W block rows it is about to use.
static void enqueue_cholesky(gpx_handle* h) {
    const int64_t Np = h->Np;
    const int nP = (int)(Np / NB);
    hipStream_t s = h->stream, s2 = h->stream2, s3 = h->stream3;
    hipMemsetAsync(h->dflag, 0, sizeof(int), s);
    // outer panel width in 128-blocks: wider panels halve the read-modify-write traffic of the big trailing updates
    // (throughput-bound sizes), narrower ones keep the in-panel row updates short (chain-bound sizes); measured at
    // N = 8192 / 16384 (final kernels): W = 3: 6.59 / 30.8 ms, 4: 6.68 / 30.4, 5: 6.74 / 29.7, 6: 7.00 / 30.1
    const int CW = h->chol_w ? h->chol_w : (nP >= 96 ? 5 : CHOL_W);
    // diagnostic (option "x_skip", scripts/chol_parts.py): leave out the far updates (bit 0), the chain kernels
    // (bit 1) or the near updates (bit 2) to time the parts alone -- the results are then NOT a factorisation
    const bool far = !(h->x_skip & 1), chain = !(h->x_skip & 2), near = !(h->x_skip & 4);
    bool mid_pending = false, side_used = false, rest_used = false;
    hipStream_t s4 = h->stream4;
    if (h->x_bg > 0) {            // diagnostic background load on a stream of its own (see k_bg_mfma)
        if (!h->stream_bg) hipStreamCreateWithFlags(&h->stream_bg, hipStreamNonBlocking);
        hipStreamSynchronize(h->stream_bg);                  // (diagnostic: the previous run's load has ended,
        hipStreamSynchronize(s);                             //  both start from an idle device)
        if (h->x_bg_lds >= 100)
            hipLaunchKernelGGL(k_bg_mfma<100>, dim3((unsigned)h->x_bg), dim3(256), 0, h->stream_bg, h->x_bg_iters, h->dscal + 8);
        else
            hipLaunchKernelGGL(k_bg_mfma<72>, dim3((unsigned)h->x_bg), dim3(256), 0, h->stream_bg, h->x_bg_iters, h->dscal + 8);
    }
    int near_rows = 0;            // rows P0+1.. of the CURRENT panel whose near update runs on the third stream
    int pend_k0 = -1;             // first block row of a panel whose rest update was deferred (two-panel accumulation)
    for (int P0 = 0; P0 < nP; P0 += CW) {
        const int P1 = (P0 + CW < nP) ? P0 + CW : nP;
        for (int I = P0; I < P1; ++I) {
            const bool rl = h->chol_rl != 0;     // in-panel updates right-looking (K = 128 per step) instead of left-looking
            if (!rl && I == P0 + 1 && near_rows > 0)      // rows P0+1.. got the previous panel's update on stream 3
                hipStreamWaitEvent(s, h->ev_row[0], 0);
            if (!rl && I > P0 && chain)   // block row I <- contributions of rows P0..I-1 of this panel
                hipLaunchKernelGGL(k_row_update64, dim3((unsigned)(2 * (nP - I)), 2), dim3(GEMM64_THREADS), 0, s,
                                   h->dR, h->dS, Np, P0, I, I, (int64_t)0, nP, 3);
            const int rem = nP - 1 - I;
            if (chain && h->chol_fuse && rem > 0) {
                hipLaunchKernelGGL(k_potrf_solve16, dim3((unsigned)(2 * rem)), dim3(256), 0, s, h->dS, h->dR, h->dT, h->dU,
                                   Np, I, h->dflag);
            } else {
                if (chain)
                    hipLaunchKernelGGL(k_potrf16<false>, dim3(1), dim3(256), 0, s, h->dS, h->dR, h->dT, h->dU, Np, I,
                                       h->dflag, (long long*)nullptr, (int64_t)0);
                if (rem > 0 && chain)
                    hipLaunchKernelGGL(k_panel_solve16, dim3((unsigned)(2 * rem)), dim3(256), 0, s, h->dU, h->dS, h->dR,
                                       Np, I, h->dflag, (int64_t)0);
            }
            if (rl && I + 1 < P1) {
                // RIGHT-LOOKING inside the panel: the rows of this panel that are still to come receive row I's
                // contribution now (K = 128 per launch: the latency of a K = 128 tile, ~12 us, instead of the
                // K = 128 .. 384 of the left-looking row update, 12 .. 24 us, in front of every diagonal block).
                // The same FMAs in the same order per element (accumulators start from S): bit-identical.
                if (I == P0 && near_rows > 0) hipStreamWaitEvent(s, h->ev_row[0], 0);
                if (chain)
                    hipLaunchKernelGGL(k_row_update64, dim3((unsigned)(2 * (nP - I - 1)), (unsigned)(2 * (P1 - I - 1))),
                                       dim3(GEMM64_THREADS), 0, s, h->dR, h->dS, Np, I, I + 1, I + 1, (int64_t)0, nP, 3);
            }
        }
        near_rows = 0;
        if (P1 >= nP) break;
        const int nnear = (P1 + CW < nP) ? CW : nP - P1;
        const int m0 = P1 + nnear;                                    // first block row of mid(P)
        const int nmid = (m0 + CW < nP) ? CW : nP - m0;       // may be 0
        const int r0 = m0 + nmid;                                     // first block row of rest(P)
        const int nrest = nP - r0;
        if (mid_pending) hipStreamWaitEvent(s, h->ev_far, 0);        // mid(P-1) (and rest(P-2)) wrote these rows
        mid_pending = false;
        hipEventRecord(h->ev_chain, s);                               // R rows P0..P1-1 are final, mid(P-1) joined
        {
            // near(P) row by row.  Only block row P1 is needed before the next diagonal block can be factored: it is
            // updated on 64x64 tiles (4x the workgroups, a quarter of the K = 512 tile latency each; the one-launch
            // 128-tile form cost ~115 us of pure critical path per panel = the latency of ONE workgroup per CU at
            // half the matrix pipe -- and so did each row launched separately in that form).  The other rows of the
            // next panel are brought up to date, on 64x64 tiles too, by ONE launch on a third stream (wave priority 2,
            // below the chain's 3) WHILE the chain already works on row P1; the chain waits for its event just
            // before row P1 + 1's in-panel update.
            if (near)
                hipLaunchKernelGGL(k_row_update64, dim3((unsigned)(2 * (nP - P1)), 2), dim3(GEMM64_THREADS), 0, s, h->dR,
                                   h->dS, Np, P0, P1, P1, (int64_t)0, nP, 3);
            if (nnear > 1 && near) {       // rows P1+1 .. in ONE launch at a lower wave priority, one event
                hipStreamWaitEvent(s3, h->ev_chain, 0);
                hipLaunchKernelGGL(k_row_update64, dim3((unsigned)(2 * (nP - P1 - 1)), (unsigned)(2 * (nnear - 1))),
                                   dim3(GEMM64_THREADS), 0, s3, h->dR, h->dS, Np, P0, P1, P1 + 1, (int64_t)0, nP, 2);
                hipEventRecord(h->ev_row[0], s3);
                near_rows = nnear - 1;
            }
        }
        if (nmid > 0) {
            // mid(P) and rest(P) touch disjoint block rows and both follow rest(P-1) (which wrote all of them): they
            // run CONCURRENTLY, mid on a fourth stream -- alone on the side stream its <= 4 x 56 tiles (one workgroup
            // per CU) held the chip for 150-190 us per panel before the big update could start.
            //
            // TWO-PANEL ACCUMULATION (option chol_merge = minimum number of rest block rows, 0 = off): while the far
            // region is large, rest(P) of every other panel is DEFERRED and applied together with the next panel's in one
            // pass with twice the K extent -- half as many read-modify-write passes over the far tiles, and tiles long
            // enough to amortise their S-tile load / store and pipeline fill (K = 512: 50 TFLOP/s, K = 640: 61).  The
            // rows that cannot wait (the next panel's: near, the one after: mid) are updated at once as before; mid of
            // the FLUSHING panel covers both panels, because its rows were in the deferred rest.  Every element still
            // receives every panel's contribution once, k ascending: bit-identical.
            const bool flush = pend_k0 >= 0;
            const int k0 = flush ? pend_k0 : P0;
            const bool defer = !flush && h->chol_merge > 0 && nrest >= h->chol_merge;
            pend_k0 = defer ? P0 : -1;
            hipStreamWaitEvent(s4, h->ev_chain, 0);
            if (rest_used) hipStreamWaitEvent(s4, h->ev_rest, 0);          // rest(P-1) wrote mid(P)'s rows
            if (far)
                hipLaunchKernelGGL(k_syrk_update, dim3((unsigned)(nP - m0), (unsigned)nmid), dim3(GEMM_THREADS), 0, s4,
                               h->dR, h->dS, Np, k0, P1, m0, m0, (int64_t)0);
            hipEventRecord(h->ev_far, s4);
            mid_pending = true;
            side_used = true;
            if (nrest > 0 && !defer) {
                hipStreamWaitEvent(s2, h->ev_chain, 0);
                if (far)
                    hipLaunchKernelGGL(k_syrk_update_tri, dim3((unsigned)(nrest * (nrest + 1) / 2)), dim3(GEMM_THREADS),
                                       0, s2, h->dR, h->dS, Np, k0, P1, r0, nrest);
                hipEventRecord(h->ev_rest, s2);
                rest_used = true;
            }
        }
    }
    if (side_used) {   // join: everything queued on the side streams is done before the caller's stream goes on
        hipEventRecord(h->ev_far, s4);
        hipStreamWaitEvent(s, h->ev_far, 0);
        if (rest_used) hipStreamWaitEvent(s, h->ev_rest, 0);
    }
    // (stream 3 needs no join: every launch on it is followed by an event the main stream has waited on)
    // only the 16x16 inverses are in the diagonal blocks of T / U so far; launch_trtri completes them
    h->diag_inv_pending = true;
}

// The factorisation of one size is the same ~250 launches and ~100 event operations over four streams every time:
// option chol_graph captures them ONCE per (size, schedule options, buffers) into a hipGraph and replays it.  The
// cross-stream waits become edges of the graph (every side stream forks from and joins the handle's stream inside
// enqueue_cholesky), the kernels and their arguments are the captured ones -- the result is bit-identical.
void launch_cholesky(gpx_handle* h) {
    const bool plain = h->x_skip == 0 && h->x_bg <= 0;
    if (!h->chol_graph || !plain) {
        enqueue_cholesky(h);
        return;
    }
    CholGraphKey key;
    std::memset(&key, 0, sizeof key);
    key.Np = h->Np; key.w = h->chol_w; key.rl = h->chol_rl; key.merge = h->chol_merge; key.fuse = h->chol_fuse;
    key.S = h->dS; key.R = h->dR; key.T = h->dT; key.U = h->dU; key.flag = h->dflag;
    key.s2 = h->stream2; key.s3 = h->stream3; key.s4 = h->stream4;
    if (!h->chol_exec || std::memcmp(&key, &h->chol_key, sizeof key) != 0) {
        if (h->chol_exec) { hipGraphExecDestroy(h->chol_exec); h->chol_exec = nullptr; }
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            enqueue_cholesky(h);
            ok = hipStreamEndCapture(h->stream, &g) == hipSuccess && g != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&h->chol_exec, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) hipGraphDestroy(g);
        if (!ok) {                      // capture is an optimisation: without it the launches go out one by one
            (void)hipGetLastError();
            h->chol_exec = nullptr;
            h->chol_graph = 0;
            enqueue_cholesky(h);
            return;
        }
        h->chol_key = key;
    }
    if (hipGraphLaunch(h->chol_exec, h->stream) != hipSuccess) {
        (void)hipGetLastError();
        enqueue_cholesky(h);
        return;
    }
    h->diag_inv_pending = true;
}

// ------------------------------------------------------------------------------------------------
// Triangular inversion by recursive doubling.  With diagonal 128-blocks already inverted, level
// `hb` merges block ranges [r1, r1+hb) and [r1+hb, r1+hb+size2):
//      T_21 = - T_22 (L_21 T_11),   L_21(m,k) = R[r1+k][r2+m]
// as two GEMM launches over all groups.  Every operand is k-major: L_21 via R, T_11 via T (row-major
// lower), T_22 via U = T^T.  GEMM2 stores T_21 and, transposed, U_12, keeping both views current.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_trtri_gemm1(const double* __restrict__ R,
                                                                 const double* __restrict__ T,
                                                                 double* __restrict__ W, int64_t Np,
                                                                 int nP, int hb, int g0) {
    // K-extent of a tile is (hb - bn) blocks: bn is the SLOW grid index so tiles are dispatched heaviest
    // first (LPT) -- with bn fastest, some CU slots drew two K = hb*128 tiles and set the makespan
    const int g = blockIdx.z + g0, bm = blockIdx.x, bn = blockIdx.y;
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * NB, n0 = r1e + (int64_t)bn * NB;
    d4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128_l<32, 1, 2>(acc, R + r1e * Np + m0, Np, T + r1e * Np + n0, Np, bn * NB, hb * NB, smem);      // operands by LDS-DMA (round 6)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) W[(m0 + acc_row(i, r)) * Np + n0 + acc_col(j)] = acc[i][j][r];
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void k_trtri_gemm2(const double* __restrict__ W,
                                                                 double* __restrict__ T,
                                                                 double* __restrict__ U, int64_t Np,
                                                                 int nP, int hb, int g0) {
    // K-extent is (bm + 1) blocks: heaviest (largest bm) first
    const int g = blockIdx.z + g0, bm = hb - 1 - (int)blockIdx.y, bn = blockIdx.x;
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * NB, n0 = r1e + (int64_t)bn * NB;
    d4 acc[4][4];
    acc_zero(acc);
    // A(m,k) = T_22(m,k) = U[r2e+k][m0+m], k <= m  ->  k-blocks [0, bm]
    gemm_tile_128_l<32, 1, 2, true, true>(acc, U + r2e * Np + m0, Np, W + r2e * Np + n0, Np, 0, (bm + 1) * NB, smem);      // the last k-block of T_22 is triangular: its all-zero quarter-rows are skipped (interleaved row blocks: acc_row_ilv)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gm = m0 + acc_row_ilv(i, r), gn = n0 + acc_col(j);
                const double v = -acc[i][j][r];
                T[gm * Np + gn] = v;
                U[gn * Np + gm] = v;
            }
}

// The same two products on 64x64 tiles (8 waves, gemm_tile_64_g) for the levels that do not fill the chip with
// 128x128 tiles: there the makespan was ONE workgroup walking the longest K (hb = 16 at N = 8192: 512 tiles,
// K up to 2048, 480 us for 18 GFLOP = 38 TFLOP/s against 67 at the last level).  Four times the workgroups, a
// quarter of the MFMA work each, and the triangular structure of T11 / T22 is followed at 64-row granularity.
__global__ __launch_bounds__(GEMM64_THREADS) void k_trtri_gemm1_64(const double* __restrict__ R,
                                                                  const double* __restrict__ T,
                                                                  double* __restrict__ W, int64_t Np, int nP,
                                                                  int hb, int g0) {
    const int g = blockIdx.z + g0, bm = blockIdx.x, bn = blockIdx.y;        // 64-row / 64-column tile indices
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= 2 * size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM64_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * T64, n0 = r1e + (int64_t)bn * T64;
    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    gemm_tile_64_g<false>(acc, R + r1e * Np + m0, Np, T + r1e * Np + n0, Np, bn * T64, hb * NB, smem);   // T11[k][n] = 0 for k < n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) W[(m0 + acc_row64(i, r)) * Np + n0 + acc_col64()] = acc[i][r];
}

__global__ __launch_bounds__(GEMM64_THREADS) void k_trtri_gemm2_64(const double* __restrict__ W,
                                                                  double* __restrict__ T,
                                                                  double* __restrict__ U, int64_t Np, int nP,
                                                                  int hb, int g0) {
    const int g = blockIdx.z + g0, bm = 2 * hb - 1 - (int)blockIdx.y, bn = blockIdx.x;   // heaviest (largest bm) first
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= 2 * size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM64_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * T64, n0 = r1e + (int64_t)bn * T64;
    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    // A(m,k) = T_22(m,k) = U[r2e+k][m0+m], zero for k > m  ->  k in [0, (bm + 1) * 64)
    gemm_tile_64_g<false>(acc, U + r2e * Np + m0, Np, W + r2e * Np + n0, Np, 0, (bm + 1) * T64, smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gm = m0 + acc_row64(i, r), gn = n0 + acc_col64();
            const double v = -acc[i][r];
            T[gm * Np + gn] = v;
            U[gn * Np + gm] = v;
        }
}

// ---- the same recursion RE-ASSOCIATED (round 4, option trtri_left):  T_21 = - (T_22 L_21) T_11 -------------------------
// Built on the expectation that this order keeps the LEFT residual (T L - I, the one the sweep's error is proportional to)
// at rounding level: with T_21 = -T_22 (L_21 T_11) the error d of the inner product enters it as T_22 d L_11, which carries
// |T_11| |L_11| ~ the conditioning of the leading block.  MEASURED (profiles/history/r04_illcond_vs_long_double.txt): the left
// residual does not improve (config B, sn2 = 1e-6 rho: 1.9e-11 -> 3.1e-11) -- in this order the OUTER product's rounding
// error, ~eps |W'| |T_11|, is what gets multiplied by L_11, with the same factor.  Only a triangular SOLVE with L_11
// (backward stable: error ~eps |T_21| |L_11|) or the Newton step of refine_inverse removes it.  The mean's error is
// 1.2-1.9x smaller with this order, the variance's is not; it costs one transposition pass (L_21 as a RIGHT factor needs
// L row-major: the strictly upper 128-blocks of R go to the strictly LOWER blocks of the workspace S; W' = T_22 L_21 is
// stored TRANSPOSED into S's upper blocks, where the second product reads it as its k-major left factor).  Off by default.
__global__ __launch_bounds__(1024) void k_transpose_offdiag(const double* __restrict__ R, int64_t Np, double* __restrict__ Lrm) {
    // 32 x 32 tiles of the strictly upper 128-blocks of R -> the mirrored position
    const int by = blockIdx.y, bx = blockIdx.x;
    if ((bx >> 2) <= (by >> 2)) return;
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)by * 32, c0 = (int64_t)bx * 32;
    tile[threadIdx.y][threadIdx.x] = R[(r0 + threadIdx.y) * Np + c0 + threadIdx.x];
    __syncthreads();
    Lrm[(c0 + threadIdx.y) * Np + r0 + threadIdx.x] = tile[threadIdx.x][threadIdx.y];
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void k_trtri_gemm1r(const double* __restrict__ U, double* __restrict__ S,
                                                                  int64_t Np, int nP, int hb) {
    // W'(m, n) = sum_k T_22(m, k) L_21(k, n), k <= m: heaviest (largest bm) first
    const int g = blockIdx.z, bm = hb - 1 - (int)blockIdx.y, bn = blockIdx.x;
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * NB, n0 = r1e + (int64_t)bn * NB;
    d4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128_l<32, 1, 2, true, true>(acc, U + r2e * Np + m0, Np, S + r2e * Np + n0, Np, 0, (bm + 1) * NB, smem);      // the last k-block of T_22 is triangular: its all-zero quarter-rows are skipped (interleaved row blocks: acc_row_ilv)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(n0 + acc_col(j)) * Np + m0 + acc_row_ilv(i, r)] = acc[i][j][r];     // W'^T, upper position
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void k_trtri_gemm2r(const double* __restrict__ S, double* __restrict__ T,
                                                                  double* __restrict__ U, int64_t Np, int nP, int hb) {
    // T_21(m, n) = - sum_k W'(m, k) T_11(k, n), k >= n: bn is the SLOW grid index (heaviest first)
    const int g = blockIdx.z, bm = blockIdx.x, bn = blockIdx.y;
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * NB, n0 = r1e + (int64_t)bn * NB;
    d4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128_l<32, 1, 2>(acc, S + r1e * Np + m0, Np, T + r1e * Np + n0, Np, bn * NB, hb * NB, smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gm = m0 + acc_row(i, r), gn = n0 + acc_col(j);
                const double v = -acc[i][j][r];
                T[gm * Np + gn] = v;
                U[gn * Np + gm] = v;
            }
}

__global__ __launch_bounds__(GEMM64_THREADS) void k_trtri_gemm1r_64(const double* __restrict__ U, double* __restrict__ S,
                                                                   int64_t Np, int nP, int hb) {
    const int g = blockIdx.z, bm = 2 * hb - 1 - (int)blockIdx.y, bn = blockIdx.x;   // heaviest (largest bm) first
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= 2 * size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM64_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * T64, n0 = r1e + (int64_t)bn * T64;
    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    gemm_tile_64_g<false>(acc, U + r2e * Np + m0, Np, S + r2e * Np + n0, Np, 0, (bm + 1) * T64, smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(n0 + acc_col64()) * Np + m0 + acc_row64(i, r)] = acc[i][r];
}

__global__ __launch_bounds__(GEMM64_THREADS) void k_trtri_gemm2r_64(const double* __restrict__ S, double* __restrict__ T,
                                                                   double* __restrict__ U, int64_t Np, int nP, int hb) {
    const int g = blockIdx.z, bm = blockIdx.x, bn = blockIdx.y;
    const int r1 = g * 2 * hb, r2 = r1 + hb;
    if (r2 >= nP) return;
    const int size2 = min(hb, nP - r2);
    if (bm >= 2 * size2) return;
    __shared__ __attribute__((aligned(16))) double smem[GEMM64_LDS_F64];
    const int64_t r1e = (int64_t)r1 * NB, r2e = (int64_t)r2 * NB;
    const int64_t m0 = r2e + (int64_t)bm * T64, n0 = r1e + (int64_t)bn * T64;
    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    gemm_tile_64_g<false>(acc, S + r1e * Np + m0, Np, T + r1e * Np + n0, Np, bn * T64, hb * NB, smem);   // T11[k][n] = 0 for k < n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gm = m0 + acc_row64(i, r), gn = n0 + acc_col64();
            const double v = -acc[i][r];
            T[gm * Np + gn] = v;
            U[gn * Np + gm] = v;
        }
}

// level hb over the groups [g0, g0 + ng): `which` = 1 the first product (W = L_21 T_11), 2 the second (T_21 = -T_22 W), 3 both.
// The tile size follows the WHOLE level (ngroups_all), so that a level computed in two parts runs the same kernels.
static void trtri_level(gpx_handle* h, hipStream_t s, int hb, int g0, int ng, int which) {
    const int64_t Np = h->Np;
    const int nP = (int)(Np / NB);
    if (ng <= 0) return;
    const int ngroups_all = (nP + 2 * hb - 1) / (2 * hb);
    if ((int64_t)hb * hb * ngroups_all >= 1024) {         // enough 128x128 tiles to fill the chip twice over
        dim3 grid((unsigned)hb, (unsigned)hb, (unsigned)ng);
        if (which & 1) hipLaunchKernelGGL(k_trtri_gemm1, grid, dim3(GEMM_THREADS), 0, s, h->dR, h->dT, h->dS, Np, nP, hb, g0);
        if (which & 2) hipLaunchKernelGGL(k_trtri_gemm2, grid, dim3(GEMM_THREADS), 0, s, h->dS, h->dT, h->dU, Np, nP, hb, g0);
    } else {
        dim3 grid((unsigned)(2 * hb), (unsigned)(2 * hb), (unsigned)ng);
        if (which & 1) hipLaunchKernelGGL(k_trtri_gemm1_64, grid, dim3(GEMM64_THREADS), 0, s, h->dR, h->dT, h->dS, Np, nP, hb, g0);
        if (which & 2) hipLaunchKernelGGL(k_trtri_gemm2_64, grid, dim3(GEMM64_THREADS), 0, s, h->dS, h->dT, h->dU, Np, nP, hb, g0);
    }
}

// the top level's split point: the largest power of two below nP (block rows [0, top) | [top, nP))
int trtri_top(int nP) {
    int hb = 1;
    while (2 * hb < nP) hb *= 2;
    return hb;
}

// The part of the inversion that needs only the LEADING `top` block rows of R (and R's block columns right of them, which are
// final as soon as block row top - 1 is solved): the diagonal 128-inverses and every level inside the leading group, then the
// top level's first product W = L_21 T_11 -- five eighths of the inversion's flop when nP is a power of two.  Enqueued on a side
// stream behind the task-graph factorisation's gate (kernels_chol_tg.hip: tg_launch_gate), it runs on the compute units the
// factorisation's chain-bound tail leaves idle.  W lands in the strictly LOWER blocks of S, T / U in their leading blocks:
// nothing the factorisation still reads or writes.  launch_trtri then does the rest (h->ahead_top != 0).
void launch_trtri_ahead(gpx_handle* h, hipStream_t s, int top) {
    const int64_t Np = h->Np;
    hipLaunchKernelGGL(k_trtri_diag128, dim3((unsigned)top), dim3(256), 0, s, h->dR, h->dT, h->dU, Np, 0, h->dflag);
    for (int hb = 1; hb < top; hb *= 2) trtri_level(h, s, hb, 0, top / (2 * hb), 3);
    trtri_level(h, s, top, 0, 1, 1);
}

void launch_trtri(gpx_handle* h) {
    const int64_t Np = h->Np;
    const int nP = (int)(Np / NB);
    hipStream_t s = h->stream;
    if (h->ahead_top > 0) {        // the leading part is on its way (launch_trtri_ahead): wait for it, do the trailing group and the top level's second product
        const int top = h->ahead_top;
        h->ahead_top = 0;
        h->tacc[T_AHEAD] += 1.0;
        (void)hipStreamWaitEvent(s, h->ev_rest, 0);
        hipLaunchKernelGGL(k_trtri_diag128, dim3((unsigned)(nP - top)), dim3(256), 0, s, h->dR, h->dT, h->dU, Np, top, h->dflag);
        h->diag_inv_pending = false;
        for (int hb = 1; hb < top; hb *= 2) {
            const int g0 = top / (2 * hb), ngroups_all = (nP + 2 * hb - 1) / (2 * hb);
            trtri_level(h, s, hb, g0, ngroups_all - g0, 3);
        }
        trtri_level(h, s, top, 0, 1, 2);
        return;
    }
    if (h->diag_inv_pending) {     // all diagonal 128-blocks at once, off the factorisation's critical path
        hipLaunchKernelGGL(k_trtri_diag128, dim3((unsigned)nP), dim3(256), 0, s, h->dR, h->dT, h->dU, Np, 0, h->dflag);
        h->diag_inv_pending = false;
    }
    const bool left = h->trtri_left != 0;          // re-associated recursion (option, off by default)
    if (left && nP > 1)
        hipLaunchKernelGGL(k_transpose_offdiag, dim3((unsigned)(Np / 32), (unsigned)(Np / 32)), dim3(32, 32), 0, s, h->dR, Np, h->dS);
    for (int hb = 1; hb < nP; hb *= 2) {
        const int ngroups = (nP + 2 * hb - 1) / (2 * hb);
        if (!left) { trtri_level(h, s, hb, 0, ngroups, 3); continue; }
        if ((int64_t)hb * hb * ngroups >= 1024) {
            dim3 grid((unsigned)hb, (unsigned)hb, (unsigned)ngroups);
            hipLaunchKernelGGL(k_trtri_gemm1r, grid, dim3(GEMM_THREADS), 0, s, h->dU, h->dS, Np, nP, hb);
            hipLaunchKernelGGL(k_trtri_gemm2r, grid, dim3(GEMM_THREADS), 0, s, h->dS, h->dT, h->dU, Np, nP, hb);
        } else {
            dim3 grid((unsigned)(2 * hb), (unsigned)(2 * hb), (unsigned)ngroups);
            hipLaunchKernelGGL(k_trtri_gemm1r_64, grid, dim3(GEMM64_THREADS), 0, s, h->dU, h->dS, Np, nP, hb);
            hipLaunchKernelGGL(k_trtri_gemm2r_64, grid, dim3(GEMM64_THREADS), 0, s, h->dS, h->dT, h->dU, Np, nP, hb);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// OPTION "refine_inverse": one Newton step on the triangular inverse, in the form that serves the sweep.
//
// The sweep forms V = T k*; its error is (T L - I) V (L = R^T): the LEFT residual of the computed inverse times a
// well-scaled vector.  The recursive doubling above computes T21 = -T22 (L21 T11), which keeps the RIGHT residual
// L T - I small; its rounding error in the left residual carries a factor |T11| |L11| (measured against long double at
// sn2 = 1e-6 rho: the explicit inverse is 10-20x less accurate than substitution).
//     E = I - T L                 (left residual, lower triangular, ~cond * eps)
//     T <- T + E T  = (2I - T L) T
// squares the left residual down to the rounding error of E itself (~eps |T| |L|, what a row-wise substitution leaves).
// Both products are lower x lower triangular GEMMs on the tile engine; L as a k-major B operand needs a transposed
// copy of R (every operand here is k-major: R row-major serves L only as an A operand).  2 N^3 / 3 more flop.
//   k_transpose_full   Lrm = R^T                                  (Np x Np, 32 x 32 tiles through LDS)
//   k_tri_lower_prod<1>  Et(J-cols, I-rows) = (delta - sum_K T(I,K) L(K,J))^T        A = U (= T^T, k-major T), B = Lrm
//   k_tri_lower_prod<2>  T1(I,J) = T(I,J) + sum_K E(I,K) T(K,J), U(J,I) = T1(I,J)^T   A = Et (k-major E), B = T
// K runs over blocks J..I (both factors lower triangular).  T1 goes to the workspace (T is still being read), U is
// rewritten in place (not an input of the second product); the caller swaps the T buffer with the workspace.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_transpose_full(const double* __restrict__ A, int64_t Np, double* __restrict__ At) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    tile[threadIdx.y][threadIdx.x] = A[(r0 + threadIdx.y) * Np + c0 + threadIdx.x];
    __syncthreads();
    At[(c0 + threadIdx.y) * Np + r0 + threadIdx.x] = tile[threadIdx.x][threadIdx.y];
}

template <int MODE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_tri_lower_prod(const double* __restrict__ Ak,
                                                                    const double* __restrict__ Bk,
                                                                    const double* __restrict__ T0,
                                                                    double* __restrict__ out,
                                                                    double* __restrict__ outT, int64_t Np, int nP) {
    // tiles (I, J), I >= J, heaviest (largest I - J) first: linear index over diagonals d = nP-1 .. 0
    int idx = blockIdx.x, dgl = nP - 1;
    while (idx >= nP - dgl) { idx -= nP - dgl; --dgl; }       // diagonal dgl holds nP - dgl tiles
    const int J = idx, I = idx + dgl;
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    const int64_t i0 = (int64_t)I * NB, j0 = (int64_t)J * NB;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = acc_row(i, r), n = acc_col(j);
                acc[i][j][r] = (MODE == 1) ? ((I == J && m == n) ? 1.0 : 0.0) : T0[(i0 + m) * Np + j0 + n];
            }
    if (MODE == 1) gemm_tile_128_g<1, true>(acc, Ak + i0, Np, Bk + j0, Np, J * NB, (I + 1) * NB, smem);
    else gemm_tile_128_g<1, false>(acc, Ak + i0, Np, Bk + j0, Np, J * NB, (I + 1) * NB, smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gm = i0 + acc_row(i, r), gn = j0 + acc_col(j);
                if (MODE == 1) {
                    outT[gn * Np + gm] = acc[i][j][r];                   // E^T: the k-major A operand of the second product
                } else {
                    const double v = (gn <= gm) ? acc[i][j][r] : 0.0;    // exact zeros above the diagonal
                    out[gm * Np + gn] = v;
                    outT[gn * Np + gm] = v;
                }
            }
}

// T, U <- refined (see above).  `tmp` (Np x Np) is the caller's scratch for E^T; S is the workspace (free after launch_trtri).
void launch_refine_inverse(gpx_handle* h, double* tmp) {
    const int64_t Np = h->Np;
    const int nP = (int)(Np / NB);
    hipStream_t s = h->stream;
    const unsigned ntiles = (unsigned)(nP * (nP + 1) / 2);
    hipLaunchKernelGGL(k_transpose_full, dim3((unsigned)(Np / 32), (unsigned)(Np / 32)), dim3(32, 32), 0, s, h->dR, Np, h->dS);
    hipLaunchKernelGGL(k_tri_lower_prod<1>, dim3(ntiles), dim3(GEMM_THREADS), 0, s, h->dU, h->dS, (const double*)nullptr,
                       (double*)nullptr, tmp, Np, nP);
    hipLaunchKernelGGL(k_tri_lower_prod<2>, dim3(ntiles), dim3(GEMM_THREADS), 0, s, tmp, h->dT, h->dT, h->dS, h->dU, Np, nP);
    std::swap(h->dT, h->dS);          // the refined T lives in the old workspace; the old T buffer is the workspace now
}

// ------------------------------------------------------------------------------------------------
// R3: a = T (y - bias) and alpha = U a as row-dot matvecs (one wave per row; HBM-read bound).
// mode 0: out[i] = sum_{j <= i} Mx[i][j] * (v[j] - shift)   (T, lower)
// mode 1: out[i] = sum_{j >= i} Mx[i][j] * v[j]             (U, upper)
// Rows >= N (padding) produce 0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tri_matvec(const double* __restrict__ Mx, int64_t Np, int64_t N,
                                                    const double* __restrict__ v, double shift,
                                                    int mode, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Np) return;
    double acc = 0.0;
    if (row < N) {
        const int64_t lo = (mode == 0) ? 0 : row;
        const int64_t hi = (mode == 0) ? row + 1 : N;
        const double* mr = Mx + row * Np;
        for (int64_t j = lo + lane; j < hi; j += 64) acc = fma(mr[j], v[j] - shift, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[row] = acc;
}

void launch_alpha(gpx_handle* h) {
    const unsigned g = (unsigned)((h->Np + 3) / 4);
    hipLaunchKernelGGL(k_tri_matvec, dim3(g), dim3(256), 0, h->stream, h->dT, h->Np, h->N, h->dy,
                       h->bias, 0, h->da);
    hipLaunchKernelGGL(k_tri_matvec, dim3(g), dim3(256), 0, h->stream, h->dU, h->Np, h->N, h->da, 0.0,
                       1, h->dalpha);
}

// diag(K^-1) = row sums of squares of U = R^-1 (K^-1 = U U^T): the posterior variance AT the observed inputs has the
// closed form  s2_i = sn2 - sn2^2 [K^-1]_ii  (k_i = K e_i - sn2 e_i  =>  k_i^T K^-1 k_i = k_ii - sn2 + sn2^2 [K^-1]_ii),
// one HBM-read pass over U instead of the N x N x N triangular product a sweep over X_obs costs.
__global__ __launch_bounds__(256) void k_row_sumsq_upper(const double* __restrict__ U, int64_t Np, int64_t N,
                                                         double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const double* ur = U + row * Np;
    double acc = 0.0;
    for (int64_t j = row + lane; j < N; j += 64) acc = fma(ur[j], ur[j], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[row] = acc;
}

void launch_kinv_diag(gpx_handle* h, double* out) {
    hipLaunchKernelGGL(k_row_sumsq_upper, dim3((unsigned)((h->N + 3) / 4)), dim3(256), 0, h->stream, h->dU, h->Np, h->N,
                       out);
}

// ------------------------------------------------------------------------------------------------
// Batched log marginal likelihoods: B hyper-parameter vectors on the handle's resident data, ONE launch chain
// (batch element = blockIdx.z of the Gram / Cholesky kernels above) and ONE host synchronisation -- what a
// hyper-parameter sampler asks for, proposal after proposal (reggie.MCMC(gp, n=10, burn=100) refits ~60 + 10
// times per model.add_data, pybo/bayesopt.py:115,269).  Neither the triangular inverse nor the handle's own
// factorisation is touched: a = R^-T (y - bias) comes from a blocked forward substitution with the 16x16
// inverses k_potrf16 leaves behind (kept in the dead diagonal blocks of the batch's Gram buffer).
//     out_b = -1/2 a.a - sum_i log R_ii - N/2 log(2 pi)          (R&W eq. 2.30),   -inf if not positive definite
// One workgroup per batch element walks the 128-blocks in order:
//     r_p   = (y_p - bias) - sum_{k < p0} R[k][p0 + j] a[k]      two threads per column j, coalesced rows of R
//     R_pp^T x = r_p  by substitution over the eight 16-tiles:  x_jb = T_d r_jb ;  r[c] -= R[jb rows][c] . x_jb
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loglik_fwd(const double* __restrict__ R, const double* __restrict__ Ud,
                                                    int64_t bs, int64_t Np, int64_t N, const double* __restrict__ y,
                                                    const double* __restrict__ hyp, const int* __restrict__ flag,
                                                    double* __restrict__ a, double* __restrict__ out) {
    __shared__ double r[NB], x[16], part[2][NB], red[4];
    const int z = blockIdx.x, t = threadIdx.x;
    R += (int64_t)z * bs;
    Ud += (int64_t)z * bs;
    a += (int64_t)z * Np;
    if (flag[z] != 0) {
        if (t == 0) out[z] = -__builtin_huge_val();
        return;
    }
    const double bias = hyp[3 * z + 2];
    const int nP = (int)(Np / NB);
    const int j = t & 127, half = t >> 7;
    double q = 0.0, ld = 0.0;
    for (int p = 0; p < nP; ++p) {
        const int64_t p0 = (int64_t)p * NB;
        double acc = 0.0;
        for (int64_t k = half; k < p0; k += 2) acc = fma(R[k * Np + p0 + j], a[k], acc);
        part[half][j] = acc;
        __syncthreads();
        if (t < NB) {
            const int64_t gi = p0 + t;
            r[t] = ((gi < N) ? y[gi] - bias : 0.0) - (part[0][t] + part[1][t]);
        }
        __syncthreads();
#pragma unroll 1
        for (int jb = 0; jb < 8; ++jb) {
            if (t < 16) {                       // x_jb = T_d r_jb,  T_d[m][k] = Ud[(16jb + k) Np + 16jb + m]
                double v = 0.0;
                for (int k = 0; k <= t; ++k)
                    v = fma(Ud[(p0 + 16 * jb + k) * Np + p0 + 16 * jb + t], r[16 * jb + k], v);
                x[t] = v;
                a[p0 + 16 * jb + t] = v;
                const double dg = R[(p0 + 16 * jb + t) * Np + p0 + 16 * jb + t];
                if (p0 + 16 * jb + t < N) { q = fma(v, v, q); ld += log(dg); }
            }
            __syncthreads();
            if (t < NB && t >= 16 * (jb + 1)) {
                double v = r[t];
                for (int k = 0; k < 16; ++k) v = fma(-R[(p0 + 16 * jb + k) * Np + p0 + t], x[k], v);
                r[t] = v;
            }
            __syncthreads();
        }
        __threadfence_block();                  // a[] of this block is read by every thread in the next one
        __syncthreads();
    }
    // only threads 0..15 carry partial sums (one wave)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        q += __shfl_xor(q, off);
        ld += __shfl_xor(ld, off);
    }
    if (t == 0) { red[0] = q; red[1] = ld; }
    __syncthreads();
    if (t == 0) out[z] = -0.5 * red[0] - red[1] - 0.5 * (double)N * 1.83787706640934548356;
}

// ---- the same forward substitution for LARGE factors (round 5): k_loglik_fwd walks the whole factor with ONE workgroup per
// vector -- 28 ms at N = 8192 (268 MB through one CU), against 5 ms for the factorisation itself.  Right-looking and blocked
// instead: per 128-block p one small launch solves R_pp^T a_p = r_p (the same 16 x 16 substitution) and one launch over all
// column blocks q > p applies  r_q -= R[p, q]^T a_p  at HBM rate.  2 nP launches; values do not depend on the batch.
__global__ void k_fwd_init(const double* __restrict__ y, const double* __restrict__ hyp, int64_t N, int64_t Np,
                           double* __restrict__ r, double* __restrict__ acc) {
    const int z = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Np) r[(int64_t)z * Np + i] = (i < N) ? y[i] - hyp[3 * z + 2] : 0.0;
    if (i < 2) acc[2 * z + i] = 0.0;
}

__global__ __launch_bounds__(128) void k_fwd_diag(const double* __restrict__ R, const double* __restrict__ Ud, int64_t bs,
                                                  int64_t Np, int64_t N, int p, int last, const int* __restrict__ flag,
                                                  double* __restrict__ rbuf, double* __restrict__ a,
                                                  double* __restrict__ acc, double* __restrict__ out) {
    __shared__ double r[NB], x[16];
    const int z = blockIdx.x, t = threadIdx.x;
    if (flag[z] != 0) {
        if (t == 0 && last) out[z] = -__builtin_huge_val();
        return;
    }
    R += (int64_t)z * bs;
    Ud += (int64_t)z * bs;
    const int64_t p0 = (int64_t)p * NB;
    r[t] = rbuf[(int64_t)z * Np + p0 + t];
    double q = 0.0, ld = 0.0;
    // the operands of step jb do not depend on x: they are loaded one step ahead (the first version read them inside the
    // dependent FMA chains: 16 global round trips per step, 30 us per block)
    double ud[2][16], rr[2][16], dgv[2];
    auto fetch = [&](int jb, int slot) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            ud[slot][k] = (t < 16) ? Ud[(p0 + 16 * jb + k) * Np + p0 + 16 * jb + t] : 0.0;
            rr[slot][k] = R[(p0 + 16 * jb + k) * Np + p0 + t];
        }
        dgv[slot] = (t < 16) ? R[(p0 + 16 * jb + t) * Np + p0 + 16 * jb + t] : 1.0;
    };
    fetch(0, 0);
    __syncthreads();
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        const int cur = jb & 1;
        if (jb + 1 < 8) fetch(jb + 1, cur ^ 1);
        if (t < 16) {                       // x_jb = T_d r_jb,  T_d[m][k] = Ud[(16jb + k) Np + 16jb + m]
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k <= t) v = fma(ud[cur][k], r[16 * jb + k], v);
            x[t] = v;
            a[(int64_t)z * Np + p0 + 16 * jb + t] = v;
            if (p0 + 16 * jb + t < N) { q = fma(v, v, q); ld += log(dgv[cur]); }
        }
        __syncthreads();
        if (t >= 16 * (jb + 1)) {
            double v = r[t];
#pragma unroll
            for (int k = 0; k < 16; ++k) v = fma(-rr[cur][k], x[k], v);
            r[t] = v;
        }
        __syncthreads();
    }
    if (t < 64) {                           // threads 0..15 carry the partial sums (one wave)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            q += __shfl_xor(q, off);
            ld += __shfl_xor(ld, off);
        }
        if (t == 0) {
            const double qq = acc[2 * z] + q, ll = acc[2 * z + 1] + ld;
            acc[2 * z] = qq;
            acc[2 * z + 1] = ll;
            if (last) out[z] = -0.5 * qq - ll - 0.5 * (double)N * 1.83787706640934548356;
        }
    }
}

// r[q0 + j] -= sum_{k < 128} R[(p0 + k) Np + q0 + j] a[p0 + k]   for the column blocks q > p (blockIdx.x), vector blockIdx.y
__global__ __launch_bounds__(256) void k_fwd_update(const double* __restrict__ R, int64_t bs, int64_t Np, int p,
                                                    const int* __restrict__ flag, const double* __restrict__ a,
                                                    double* __restrict__ rbuf) {
    __shared__ double av[NB], part[NB];
    const int z = blockIdx.y, t = threadIdx.x;
    if (flag[z] != 0) return;
    R += (int64_t)z * bs;
    const int64_t p0 = (int64_t)p * NB, q0 = (int64_t)(p + 1 + blockIdx.x) * NB;
    if (t < NB) av[t] = a[(int64_t)z * Np + p0 + t];
    __syncthreads();
    const int j = t & 127, half = t >> 7;
    double s = 0.0;
#pragma unroll 8
    for (int k = half; k < NB; k += 2) s = fma(R[(p0 + k) * Np + q0 + j], av[k], s);
    if (half) part[j] = s;
    __syncthreads();
    if (!half) rbuf[(int64_t)z * Np + q0 + j] -= (s + part[j]);
}

static int loglik_batch_chunk(gpx_handle* h, int64_t B, const double* hyp, double* out);

// any number of vectors: 64 (or what 16 GB of batch buffers hold) per launch chain
int loglik_batch_host(gpx_handle* h, int64_t B, const double* hyp, double* out) {
    if (h->stage < 1) { h->err = "loglik_batch: no data on the device (fit first)"; return GPX_ESTATE; }
    if (!hyp || !out || B < 1) { h->err = "loglik_batch: need B >= 1 hyper-parameter vectors"; return GPX_EARG; }
    const double per = (double)h->Np * (double)h->Np * 16.0;
    int64_t step = (int64_t)(16.0e9 / per);
    if (step < 1) { h->err = "loglik_batch: N^2 too large for the batch buffers"; return GPX_EARG; }
    if (step > 64) step = 64;
    for (int64_t b0 = 0; b0 < B; b0 += step) {
        const int rc = loglik_batch_chunk(h, std::min(step, B - b0), hyp + b0 * (h->d + 3), out + b0);
        if (rc != GPX_OK) return rc;
    }
    return GPX_OK;
}

static int loglik_batch_chunk(gpx_handle* h, int64_t B, const double* hyp, double* out) {
    if (B < 1 || B > 64) { h->err = "loglik_batch: internal chunk size"; return GPX_EARG; }
    const int64_t N = h->N, Np = h->Np, d = h->d, bs = Np * Np;
    const int nP = (int)(Np / NB);
    if ((double)B * (double)bs * 16.0 > 16.0e9) { h->err = "loglik_batch: B * N^2 too large for the batch buffers"; return GPX_EARG; }
    std::vector<double> stage((size_t)B * (3 + DMAX), 0.0);     // [hyp B x 3 = rho, sn2, bias][invell B x DMAX]
    for (int64_t b = 0; b < B; ++b) {
        const double* v = hyp + b * (d + 3);                    // sn2, rho, ell[d], bias
        const double sn2 = v[0], rho = v[1], bias = v[2 + d];
        bool ok = (rho > 0) && (sn2 >= 0) && std::isfinite(rho) && std::isfinite(sn2) && std::isfinite(bias);
        for (int64_t k = 0; k < d; ++k) ok = ok && (v[2 + k] > 0) && std::isfinite(v[2 + k]);
        if (!ok) { h->err = "loglik_batch: need finite rho > 0, sn2 >= 0, ell > 0, bias"; return GPX_EARG; }
        stage[(size_t)b * 3] = rho;
        stage[(size_t)b * 3 + 1] = sn2;
        stage[(size_t)b * 3 + 2] = bias;
        for (int64_t k = 0; k < d; ++k) stage[(size_t)B * 3 + b * DMAX + k] = 1.0 / v[2 + k];
    }
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GPX_EHIP; }
    hipStream_t s = h->stream;
    // one allocation: [S B bs][R B bs][Xs B Np d][a B Np][hyp 3B][invell B DMAX][out B][flag B ints]
    const int64_t need = 2 * B * bs + B * Np * d + 2 * B * Np + B * (3 + DMAX) + 3 * B + (B + 1) / 2 + 8;
    if (need > h->cap_batch) {
        if (h->dbatch) hipFree(h->dbatch);
        h->dbatch = nullptr;
        h->cap_batch = 0;
        if (hipMalloc((void**)&h->dbatch, (size_t)need * 8) != hipSuccess) { h->err = "loglik_batch: device allocation failed"; return GPX_EOOM; }
        h->cap_batch = need;
    }
    double* bS = h->dbatch;
    double* bR = bS + B * bs;
    double* bXs = bR + B * bs;
    double* ba = bXs + B * Np * d;
    double* bhyp = ba + B * Np;
    double* binv = bhyp + 3 * B;
    double* bout = binv + B * DMAX;
    double* br = bout + B;                     // (large factors) the running right-hand sides, B x Np
    double* bacc = br + B * Np;                // ... and {a.a, sum log R_ii} per vector
    int* bflag = reinterpret_cast<int*>(bacc + 2 * B);
    if (hipMemcpyAsync(bhyp, stage.data(), stage.size() * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemsetAsync(bflag, 0, (size_t)B * sizeof(int), s) != hipSuccess) {
        h->err = "loglik_batch: H2D copy failed";
        return GPX_EHIP;
    }
    const unsigned Bz = (unsigned)B;
    {
        const int64_t tot = Np * d;
        hipLaunchKernelGGL(k_scale_x, dim3((unsigned)((tot + 255) / 256), 1, Bz), dim3(256), 0, s, h->dXraw, N, Np,
                           (int)d, binv, bXs);
        const unsigned g = (unsigned)(Np / GT);
        hipLaunchKernelGGL(k_gram_sym, dim3(g, g, Bz), dim3(256), 0, s, bXs, N, Np, (int)d, h->kernel_id, 1.0, 0.0, bS,
                           (const double*)bhyp);
    }
    // LARGE factors (round 5; the reference's default model at BASELINE scale, pybo/bayesopt.py:115): the persistent task-graph
    // kernel, vector after vector, on the batch's buffers (the 16 x 16 inverses land in the dead diagonal tiles of the Gram
    // buffer, exactly where k_potrf16 puts them below).  The handle's own factor is not touched: its buffer pointers are lent
    // to the launcher and restored.  A launch that cannot be prepared, or gives up, sends the WHOLE chunk down the batched
    // kernels below (which rebuild nothing: they start from the Gram matrices, so those are formed again first).
    bool big = h->chol_tg && nP >= 16 && nP >= h->tg_min && nP <= h->tg_max;
    if (big) {
        double *sS = h->dS, *sR = h->dR, *sT = h->dT, *sU = h->dU;
        // (the handle's own model may have the leading part of its inversion still riding on the side stream: that state is the
        //  model's, not these factorisations' -- launch_cholesky_tg clears it)
        const bool s_pending = h->diag_inv_pending, s_launched = h->tg_launched, s_want = h->want_ahead;
        const int s_ahead = h->ahead_top;
        // ... but it polls the handle's control block and reads the handle's pivot flag, both of which the factorisations below
        // reuse: let it finish first (its results stay the model's)
        if (s_ahead != 0 && h->stream2) (void)hipStreamSynchronize(h->stream2);
        h->want_ahead = false;
        for (int64_t b = 0; b < B && big; ++b) {
            h->dS = bS + b * bs; h->dR = bR + b * bs; h->dT = nullptr; h->dU = bS + b * bs;
            const bool ok = launch_cholesky_tg(h);
            if (ok) (void)hipMemcpyAsync(bflag + b, h->dflag, sizeof(int), hipMemcpyDeviceToDevice, s);
            if (!ok || hipStreamSynchronize(s) != hipSuccess || tg_abort_code(h) == 2) big = false;
        }
        h->dS = sS; h->dR = sR; h->dT = sT; h->dU = sU;
        h->diag_inv_pending = s_pending; h->tg_launched = s_launched; h->want_ahead = s_want; h->ahead_top = s_ahead;
        // the pivot flag is the model's again: a fitted model's flag is clear, whatever the last vector of this batch left in it
        // (the model's lazy inverse starts with `if (*flag != 0) return`)
        (void)hipMemsetAsync(h->dflag, 0, 2 * sizeof(int), s);
        if (!big) {
            (void)hipGetLastError();
            (void)hipMemsetAsync(bflag, 0, (size_t)B * sizeof(int), s);
            const unsigned g = (unsigned)(Np / GT);
            hipLaunchKernelGGL(k_gram_sym, dim3(g, g, Bz), dim3(256), 0, s, bXs, N, Np, (int)d, h->kernel_id, 1.0, 0.0, bS,
                               (const double*)bhyp);
        }
    }
    if (big) {
        hipLaunchKernelGGL(k_fwd_init, dim3((unsigned)((Np + 255) / 256), Bz), dim3(256), 0, s, h->dy, bhyp, N, Np, br, bacc);
        for (int p = 0; p < nP; ++p) {
            hipLaunchKernelGGL(k_fwd_diag, dim3(Bz), dim3(128), 0, s, bR, bS, bs, Np, N, p, p == nP - 1 ? 1 : 0, bflag, br, ba,
                               bacc, bout);
            if (p + 1 < nP)
                hipLaunchKernelGGL(k_fwd_update, dim3((unsigned)(nP - 1 - p), Bz), dim3(256), 0, s, bR, bs, Np, p, bflag, ba, br);
        }
        if (hipMemcpyAsync(out, bout, (size_t)B * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
            h->err = "loglik_batch: kernel or D2H copy failed";
            return GPX_EHIP;
        }
        return GPX_OK;
    }
    // right-looking blocked factorisation, every launch over the whole batch; single stream, no lookahead (the
    // sizes a sampler works at are a few blocks)
    for (int P0 = 0; P0 < nP; P0 += CHOL_W) {
        const int P1 = (P0 + CHOL_W < nP) ? P0 + CHOL_W : nP;
        for (int I = P0; I < P1; ++I) {
            if (I > P0)
                hipLaunchKernelGGL(k_row_update64, dim3((unsigned)(2 * (nP - I)), 2, Bz), dim3(GEMM64_THREADS), 0, s, bR, bS,
                                   Np, P0, I, I, bs, nP, 0);
            hipLaunchKernelGGL(k_potrf16<false>, dim3(1, 1, Bz), dim3(256), 0, s, bS, bR, (double*)nullptr, bS, Np, I,
                               bflag, (long long*)nullptr, bs);
            const int rem = nP - 1 - I;
            if (rem > 0)
                hipLaunchKernelGGL(k_panel_solve16, dim3((unsigned)(2 * rem), 1, Bz), dim3(256), 0, s, bS, bS, bR, Np, I,
                                   bflag, bs);
        }
        if (P1 < nP)
            hipLaunchKernelGGL(k_syrk_update, dim3((unsigned)(nP - P1), (unsigned)(nP - P1), Bz), dim3(GEMM_THREADS), 0, s,
                               bR, bS, Np, P0, P1, P1, P1, bs);
    }
    hipLaunchKernelGGL(k_loglik_fwd, dim3(Bz), dim3(256), 0, s, bR, bS, bs, Np, N, h->dy, bhyp, bflag, ba, bout);
    if (hipMemcpyAsync(out, bout, (size_t)B * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
        h->err = "loglik_batch: kernel or D2H copy failed";
        return GPX_EHIP;
    }
    return GPX_OK;
}


// ------------------------------------------------------------------------------------------------
// A small SPD system on the device (the n x n weight posterior of a Thompson draw with n >= 128 random features,
// pybo/policies/simple.py:44 -- `n` is a free keyword there; the LDS-resident k_rff_posterior stops at 127):
//   k_form_posterior_matrix   B = sc^2 A + sn2 I into the upper 128-block triangle of an (np, np) buffer, identity padding
//   launch_cholesky_small     B = R^T R with the blocked kernels above on ONE stream (no lookahead: a few blocks)
//   k_posterior_solve         theta = sc R^-1 ( R^-T (sc v) + sqrt(sn2) z ): one forward and one backward substitution
//                             on vectors, one workgroup (column sweeps over R in global memory; O(n^2))
// ------------------------------------------------------------------------------------------------
__global__ void k_form_posterior_matrix(const double* __restrict__ A, int n, int64_t np, double sc2, double sn2,
                                        double* __restrict__ B) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= np * np) return;
    const int64_t i = idx / np, j = idx - i * np;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < n && j < n) v = sc2 * A[i * n + j] + ((i == j) ? sn2 : 0.0);
    B[idx] = v;
}

void launch_cholesky_small(hipStream_t s, double* S, double* R, int64_t np, int* flag) {
    const int nP = (int)(np / NB);
    hipMemsetAsync(flag, 0, sizeof(int), s);
    for (int P0 = 0; P0 < nP; P0 += CHOL_W) {
        const int P1 = (P0 + CHOL_W < nP) ? P0 + CHOL_W : nP;
        for (int I = P0; I < P1; ++I) {
            if (I > P0)
                hipLaunchKernelGGL(k_row_update64, dim3((unsigned)(2 * (nP - I)), 2, 1), dim3(GEMM64_THREADS), 0, s, R, S, np, P0,
                                   I, I, (int64_t)0, nP, 0);
            // (the 16 x 16 inverses go to the dead diagonal blocks of S, as in the batched likelihood path)
            hipLaunchKernelGGL(k_potrf16<false>, dim3(1, 1, 1), dim3(256), 0, s, S, R, (double*)nullptr, S, np, I, flag,
                               (long long*)nullptr, (int64_t)0);
            const int rem = nP - 1 - I;
            if (rem > 0)
                hipLaunchKernelGGL(k_panel_solve16, dim3((unsigned)(2 * rem), 1, 1), dim3(256), 0, s, S, S, R, np, I, flag,
                                   (int64_t)0);
        }
        if (P1 < nP)
            hipLaunchKernelGGL(k_syrk_update, dim3((unsigned)(nP - P1), (unsigned)(nP - P1), 1), dim3(GEMM_THREADS), 0, s, R, S,
                               np, P0, P1, P1, P1, (int64_t)0);
    }
}

__global__ __launch_bounds__(256) void k_posterior_solve(const double* __restrict__ R, int64_t np, int n,
                                                         const double* __restrict__ v, const double* __restrict__ z,
                                                         double sc, double sn2, const int* __restrict__ flag,
                                                         double* __restrict__ work, double* __restrict__ theta) {
    // work: n doubles of scratch (the running right-hand side)
    if (*flag != 0) return;
    const int t = threadIdx.x;
    __shared__ double piv;
    for (int i = t; i < n; i += 256) work[i] = sc * v[i];
    __syncthreads();
    // forward: R^T u = sc v  (R^T lower: column i of R^T = row i of R, contiguous)
    for (int i = 0; i < n; ++i) {
        if (t == 0) { piv = work[i] / R[(int64_t)i * np + i]; work[i] = piv; }
        __syncthreads();
        const double u = piv;
        const double* row = R + (int64_t)i * np;
        for (int j = i + 1 + t; j < n; j += 256) work[j] = fma(-row[j], u, work[j]);
        __syncthreads();
    }
    const double sq = sqrt(sn2);
    for (int i = t; i < n; i += 256) work[i] += sq * z[i];
    __syncthreads();
    // backward: R x = u + sqrt(sn2) z  (row i of R against the solved tail: a dot product per step)
    __shared__ double part[256];
    for (int i = n - 1; i >= 0; --i) {
        const double* row = R + (int64_t)i * np;
        double acc = 0.0;
        for (int j = i + 1 + t; j < n; j += 256) acc = fma(row[j], work[j], acc);
        part[t] = acc;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (t < off) part[t] += part[t + off];
            __syncthreads();
        }
        if (t == 0) work[i] = (work[i] - part[0]) / row[i];
        __syncthreads();
    }
    for (int i = t; i < n; i += 256) theta[i] = sc * work[i];
}

void launch_posterior_wide(hipStream_t s, const double* A, const double* v, const double* z, int n, int64_t np, double sc,
                           double sn2, double* B, double* R, double* work, int* flag, double* theta) {
    const int64_t tot = np * np;
    hipLaunchKernelGGL(k_form_posterior_matrix, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, n, np, sc * sc, sn2, B);
    launch_cholesky_small(s, B, R, np, flag);
    hipLaunchKernelGGL(k_posterior_solve, dim3(1), dim3(256), 0, s, R, np, n, v, z, sc, sn2, flag, work, theta);
}

// out (N,N) row-major = transpose of the leading N x N part of src (Np,Np) keeping only the part
// that is lower-triangular in `out` (used to hand L = R^T to the parity tests)
__global__ void k_transpose_lower(const double* __restrict__ src, int64_t Np, double* __restrict__ out,
                                  int64_t N) {
    __shared__ double tile[32][33];
    const int64_t bx = (int64_t)blockIdx.x * 32, by = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int64_t sr = by + r, sc = bx + tx;
        tile[r][tx] = (sr < N && sc < N) ? src[sr * Np + sc] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t orow = bx + r, ocol = by + tx;  // out[orow][ocol] = src[ocol][orow]
        if (orow < N && ocol < N) out[orow * N + ocol] = (ocol <= orow) ? tile[tx][r] : 0.0;
    }
}

void launch_transpose_lower(hipStream_t s, const double* src, int64_t Np, double* out, int64_t N) {
    const unsigned g = (unsigned)((N + 31) / 32);
    hipLaunchKernelGGL(k_transpose_lower, dim3(g, g), dim3(256), 0, s, src, Np, out, N);
}

}  // namespace gpx
