// kernels_rff.hip -- Thompson sampling by random Fourier features on gfx950.
//
// Replaces the evaluation of `model.sample_f(n, rng).get(X)` [pybo/policies/simple.py:44-48] over a
// candidate grid.  A draw is f(x) = bias + sum_j theta_j cos(w_j . x + b_j)  (Rahimi & Recht); the
// spectral draws W, b and the weight-posterior noise are made on the host from the caller's seeded
// RandomState so a draw is reproducible; the device does the O(N n^2) feature Gram for the weight
// posterior and the O(M S n d) evaluation sweep.
#include "gemm_core.h"
#include "gpx_internal.h"

namespace gpx {

// ------------------------------------------------------------------------------------------------
// MFMA form of the Thompson sweep.  The projection Z = Xc W^T is a GEMM (M x d) x (d x S*n): on the VALU it
// is operand-delivery bound (one LDS/scalar read per FMA); the 16x16x4 outer-product structure of
// v_mfma_f64 needs 0.03 operand reads per FMA instead.  (fp64 MFMA and fp64 VALU share the DP lanes on
// gfx950, so this buys operand bandwidth, not extra flop/s -- see DESIGN.md.)
//   grid (ceil(M/128), S); workgroup = 4 waves (2x2), tile = 128 candidates x 128 features of ONE draw
//   (features zero-padded to a multiple of 128 with theta = 0), K = d padded to a multiple of 4.
//   The candidate tile stays in LDS (k-major) for the workgroup's life; feature tiles stream through.
//   Epilogue per accumulator element: theta_f * cos(z + b_f), summed over the tile's 128 columns into a
//   per-row register, reduced across lanes / the two column-waves once at the end.
// cos: 3-term Cody-Waite reduction by pi + one even polynomial on [-pi/2, pi/2] (see cos_cw).  |z| = |w.x + b| stays
// far below the 2^20 pi validity range of the reduction.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double cos_cw(double z) {
    // Round 3: reduction by pi (not pi/2) and ONE even polynomial on [-pi/2, pi/2]: cos z = (-1)^n cos r, r = z - n pi.
    // The quadrant form evaluated BOTH fdlibm kernels (sine and cosine on [-pi/4, pi/4]) and selected: 37 VALU
    // instructions per evaluation, and k_rff_mfma is bound by VALU + MFMA issue on the shared double-precision pipe
    // (profiles/history/r03_pmc_rff_mfma.txt: 43 % + 37 % of the cycles).
    // Round 4 (26 -> 20 instructions, 22 -> 18 of them on the double-precision pipe):
    //   * n = round(z / pi) by the magic-number addition t = z / pi + 1.5 * 2^52 (one FMA), n = t - 1.5 * 2^52: the integer
    //     sits in t's low mantissa bits, so the parity of n is bit 0 of t's low word -- no v_rndne, no v_cvt_i32_f64;
    //   * the sign (-1)^n is that bit shifted into the sign position and XORed into the result's high word (two 32-bit
    //     operations, off the double-precision pipe) instead of a compare and two selects;
    //   * two-term Cody-Waite: n pi_hi is exact (pi_hi has 33 significant bits, |n| < 2^20), the dropped third term is
    //     n * 4e-21 -- below 1e-19 for any argument a feature can see;
    //   * the Taylor series through r^20 / 20! (the next term is 1.8e-17 at |r| = pi/2).
    // Absolute error <= 3e-16 (Horner over terms that sum to cosh(pi/2) = 2.5 in magnitude), as before; the features enter
    // sums of ~100 terms of size theta ~ 0.1, compared with the oracle at 1e-9 relative.
    const double t = fma(z, 3.18309886183790671538e-01, 6755399441055744.0);
    const double n = t - 6755399441055744.0;
    double r = fma(-n, 3.14159265346825122833e+00, z);
    r = fma(-n, 1.21542010126079319532e-10, r);
    const double r2 = r * r;
    double p = 4.11031762331216485548e-19;                  //  1/20!
    p = fma(r2, p, -1.56192069685862264622e-16);            // -1/18!
    p = fma(r2, p, 4.77947733238738529744e-14);             //  1/16!
    p = fma(r2, p, -1.14707455977297247139e-11);            // -1/14!
    p = fma(r2, p, 2.08767569878680989792e-09);             //  1/12!
    p = fma(r2, p, -2.75573192239858906526e-07);            // -1/10!
    p = fma(r2, p, 2.48015873015873015873e-05);             //  1/8!
    p = fma(r2, p, -1.38888888888888888889e-03);            // -1/6!
    p = fma(r2, p, 4.16666666666666666667e-02);             //  1/4!
    p = fma(r2, p, -0.5);
    const double cs = fma(r2, p, 1.0);
    return __hiloint2double(__double2hiint(cs) ^ (__double2loint(t) << 31), __double2loint(cs));
}

// Wt:  [S][nfb][dp][128]   feature tiles, k-major (zero padded);  bt, tt: [S][nfb][128]
// grid (ceil(M/128)): one workgroup per 128-candidate tile, looping over all S draws.
// Wave layout 4 x 1: wave w owns candidate rows 32w .. 32w+31 and ALL 128 feature columns of the tile (2 x 8
// accumulators), so that the 16-column groups beyond the draw's n features are skipped by every wave alike --
// n = 100 (the Thompson default) uses 7 of 8 groups: 12.5 % fewer MFMAs and cosines than the padded tile; with the
// 2 x 2 layout of the GEMM engine only the waves of the right half could have skipped, and the workgroup would
// have waited for the others.  A row's sum over the features then lives in ONE wave (16-lane reduction, no LDS).
// PF (dp <= 32, candidate tile resident): the NEXT feature tile's 32 KB are loaded into registers (8 x 16 B per thread)
// before the matrix phase and written to LDS between the matrix phase and the cosine epilogue -- the global-load latency
// of a tile no longer sits between two barriers (counters: 31 % of the wave time was spent parked there).
template <bool PF>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_rff_mfma(const double* __restrict__ Wt,
                                                              const double* __restrict__ bt,
                                                              const double* __restrict__ tt, int S, int nfb, int n,
                                                              int d, int dp, int dk, double bias,
                                                              const double* __restrict__ Xc, int64_t M,
                                                              double* __restrict__ vals) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // At[dk][LDT] | Bt[dk][LDT]
    // (single-buffered feature tile: 2*dp*144*8 B = 74 KB at d = 32, so TWO workgroups share a CU and hide
    //  each other's tile loads and cosine chains; a double-buffered tile would leave one workgroup per CU)
    // dk = rows of the k-range held in LDS at a time.  dk == dp (d <= 64): the candidate tile is transposed into LDS
    // once and stays for the workgroup's life, only feature tiles stream through.  dk < dp (d > 64, round 3: the
    // Thompson entry points used to stop at d = 64): the projection walks the coordinates dk at a time, candidate and
    // feature chunks both streaming -- the accumulators carry over, the k order is the same ascending one.
    double* At = lds;
    double* Bt = lds + dk * LDT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * TB;
    const bool resident = (dk == dp);
    auto load_a = [&](int k0, int kc) {   // candidate chunk, transposed into k-major (lane <-> candidate: conflict-free LDS stores)
        for (int e = t; e < TB * kc; e += GEMM_THREADS) {
            const int k = e >> 7, m = e & 127;
            const int64_t gm = m0 + m;
            At[k * LDT + m] = (k0 + k < d && gm < M) ? Xc[gm * d + k0 + k] : 0.0;
        }
    };
    if (resident) load_a(0, dp);
    const int fr = lane & 15, fk = lane >> 4;
    const int ntile = S * nfb;
    auto load_b = [&](int tile, int k0, int kc) {   // 16-byte pieces, coalesced rows of 1 KiB
        const double* Wtile = Wt + ((int64_t)tile * dp + k0) * TB;
        double* dst = Bt;
        for (int e = t; e < (TB / 2) * kc; e += GEMM_THREADS) {
            const int k = e >> 6, c = (e & 63) * 2;
            *reinterpret_cast<d2*>(dst + k * LDT + c) = *reinterpret_cast<const d2*>(Wtile + (int64_t)k * TB + c);
        }
    };
    double rowsum[2][4];
    if (PF) {
        load_b(0, 0, dp);
        __syncthreads();
    }
    for (int tile = 0; tile < ntile; ++tile) {
        const int fb = tile % nfb, s = tile / nfb;
        const int nv = min(TB, n - fb * TB);          // features of this tile that exist
        const int jt = (nv + 15) >> 4;                // ... in 16-column groups (uniform)
        if (fb == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) rowsum[i][r] = 0.0;
        }
        d4 acc[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
        const double* as = At + w * 32 + fr;
        const double* bs = Bt + fr;
        d2 pre[8];
        if (PF) {
            // Bt already holds this tile (loaded before the loop / written below from `pre`); fetch the next one
            if (tile + 1 < ntile) {
                const double* Wnext = Wt + (int64_t)(tile + 1) * dp * TB;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = t + u * GEMM_THREADS;
                    if (e < (TB / 2) * dp) pre[u] = *reinterpret_cast<const d2*>(Wnext + (int64_t)(e >> 6) * TB + (e & 63) * 2);
                }
            }
            for (int kk = 0; kk < dp / 4; ++kk) {
                const int kr = kk * 4 + fk;
                const double a0 = as[kr * LDT], a1 = as[kr * LDT + 16];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < jt) {
                        const double b = bs[kr * LDT + j * 16];
                        acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][j], 0, 0, 0);
                    }
                }
            }
            __syncthreads();   // everyone has read this feature tile
            if (tile + 1 < ntile) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = t + u * GEMM_THREADS;
                    if (e < (TB / 2) * dp) *reinterpret_cast<d2*>(Bt + (e >> 6) * LDT + (e & 63) * 2) = pre[u];
                }
            }
        } else
        for (int k0 = 0; k0 < dp; k0 += dk) {
            const int kc = min(dk, dp - k0);
            if (!resident) load_a(k0, kc);
            load_b(tile, k0, kc);
            __syncthreads();   // this k-range of the feature tile (and of the candidate tile) is complete
            for (int kk = 0; kk < kc / 4; ++kk) {
                const int kr = kk * 4 + fk;
                const double a0 = as[kr * LDT], a1 = as[kr * LDT + 16];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < jt) {
                        const double b = bs[kr * LDT + j * 16];
                        acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][j], 0, 0, 0);
                    }
                }
            }
            if (k0 + dk < dp) __syncthreads();   // everyone has read this k-range: the next one may overwrite it
        }
        const double* bb = bt + (int64_t)tile * TB + fr;
        const double* th = tt + (int64_t)tile * TB + fr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < jt) {
                const double bj = bb[j * 16], tj = th[j * 16];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rowsum[i][r] = fma(tj, cos_cw(acc[i][j][r] + bj), rowsum[i][r]);
            }
        }
        if (fb == nfb - 1) {
            // draw s complete: reduce over the 16 lanes sharing a row; the row lives in this wave alone
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = rowsum[i][r];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 8);
                    const int64_t gm = m0 + w * 32 + i * 16 + fk + 4 * r;
                    if (fr == 0 && gm < M) vals[(int64_t)s * M + gm] = bias + v;
                }
        }
        __syncthreads();   // everyone is done with this feature tile: it may be overwritten
    }
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    union { double d; int i[2]; } u, o;
    u.d = v;
    o.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xF, 0xF, true);
    o.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xF, 0xF, true);
    return o.d;
}

// DB (dp <= 16: the LDS holds it beside a second workgroup): TWO images of the feature tile -- the next one is written while
// this one is still being read, and the only barrier of a tile stands at its end.
template <int JF, int NSUB, bool DB>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_rff_mfma5(const double* __restrict__ Wt, const double* __restrict__ bt,
                                                               const double* __restrict__ tt, int S, int d, int dp,
                                                               double bias, const double* __restrict__ Xc, int64_t M,
                                                               double* __restrict__ vals, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // At[dp][LDT] | Bt[1 or 2][dp + 2][LDT] (rows dp, dp + 1: b, theta)
    // the shader clock this launch sustains (it is power-bound: ~2.1 GHz against the 2.4 GHz of the peak): workgroup lifetimes
    // in s_memtime and in 100 MHz ticks, as in k_sweep_trmm
    const unsigned long long clk_c0 = __builtin_readcyclecounter();
    const unsigned long long clk_r0 = wall_clock64();
    double* At = lds;
    double* Bt0 = lds + dp * LDT;
    const int bstride = (dp + 2) * LDT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * TB;
    const int fr = lane & 15, fk = lane >> 4;
    const int nkk = dp >> 2;
    // the thread's 16-byte pieces of a feature tile: rows (t >> 6) + 4 u, columns 2 (t & 63); of the b / theta rows: thread t < 128
    const int vo = (int)((((t >> 6) * TB) + (t & 63) * 2) * 8);
    const int vo2 = (int)(((t & 63) * 2) * 8);
    d2 pre[8], pre2;
    auto fetch = [&](int tile) {
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Wt + (int64_t)tile * dp * TB), 0, -1, 0x00020000);
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(((t < 64) ? bt : tt) + (int64_t)tile * TB), 0, -1, 0x00020000);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (4 * u < dp) pre[u] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rw, vo, u * 4 * TB * 8, 0));
        if (t < 128) pre2 = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rb, vo2, 0, 0));
    };
    auto stash = [&](double* Bt) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (4 * u < dp) *reinterpret_cast<d2*>(Bt + ((t >> 6) + 4 * u) * LDT + (t & 63) * 2) = pre[u];
        if (t < 128) *reinterpret_cast<d2*>(Bt + (dp + (t >> 6)) * LDT + (t & 63) * 2) = pre2;
    };
    fetch(0);                                             // (in flight while the candidate tile is staged)
    for (int e = t; e < TB * dp; e += GEMM_THREADS) {     // candidate tile, transposed into k-major, for the workgroup's life
        const int k = e >> 7, m = e & 127;
        const int64_t gm = m0 + m;
        At[k * LDT + m] = (k < d && gm < M) ? Xc[gm * d + k] : 0.0;
    }
    stash(Bt0);
    __syncthreads();
    const double* as = At + w * 32 + fr;
    double rmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rmask[r] = (((lane >> 2) & 3) == r) ? 1.0 : 0.0;
    for (int tile = 0; tile < S; ++tile) {
        const double* Bt = Bt0 + ((DB && (tile & 1)) ? bstride : 0);
        const double* bs = Bt + fr;
        const double* bq = Bt + 16 * JF + (lane & 3);          // the remainder columns of this lane
        d4 acc[2][JF > 0 ? JF : 1];
        double accr[2][NSUB > 0 ? NSUB : 1];
        // the draw's phases and weights: rows dp, dp + 1 of the tile's image.  The accumulators START from the phase of their
        // column (z = b + w.x accumulates in the matrix instruction: 48 double-precision additions per tile less)
        double bj[JF > 0 ? JF : 1], tj[JF > 0 ? JF : 1], bjr[NSUB > 0 ? NSUB : 1], tjr[NSUB > 0 ? NSUB : 1];
#pragma unroll
        for (int j = 0; j < JF; ++j) {
            bj[j] = bs[dp * LDT + j * 16];
            tj[j] = bs[(dp + 1) * LDT + j * 16];
        }
#pragma unroll
        for (int q = 0; q < NSUB; ++q) {
            bjr[q] = bq[dp * LDT + 4 * q];
            tjr[q] = bq[(dp + 1) * LDT + 4 * q];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < JF; ++j) acc[i][j] = (d4){bj[j], bj[j], bj[j], bj[j]};
#pragma unroll
            for (int q = 0; q < NSUB; ++q) accr[i][q] = bjr[q];
        }
        const bool more = tile + 1 < S;
        if (more) fetch(tile + 1);
        // matrix phase: fragments of group kk + 1 requested before the MFMAs of group kk
        double a[2][2], b[2][JF > 0 ? JF : 1], br[2][NSUB > 0 ? NSUB : 1];
        auto frag = [&](int kk, int slot) {
            const int kr = kk * 4 + fk;
            a[slot][0] = as[kr * LDT];
            a[slot][1] = as[kr * LDT + 16];
#pragma unroll
            for (int j = 0; j < JF; ++j) b[slot][j] = bs[kr * LDT + j * 16];
#pragma unroll
            for (int q = 0; q < NSUB; ++q) br[slot][q] = bq[kr * LDT + 4 * q];
        };
        auto mfmas = [&](int slot) {
#pragma unroll
            for (int j = 0; j < JF; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[slot][0], b[slot][j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[slot][1], b[slot][j], acc[1][j], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NSUB; ++q) {
                accr[0][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[slot][0], br[slot][q], accr[0][q], 0, 0, 0);
                accr[1][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[slot][1], br[slot][q], accr[1][q], 0, 0, 0);
            }
        };
        frag(0, 0);
#pragma unroll 1
        for (int kk = 0; kk < nkk; kk += 2) {       // two groups per trip: the fragment slots are compile-time indices
            const bool two = kk + 1 < nkk;
            if (two) frag(kk + 1, 1);
            mfmas(0);
            if (two) {
                if (kk + 2 < nkk) frag(kk + 2, 0);
                mfmas(1);
            }
        }
        if (!DB) __syncthreads();   // everyone has read this feature tile
        if (more) stash(Bt0 + ((DB && !(tile & 1)) ? bstride : 0));
        double rowsum[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) rowsum[i][r] = 0.0;
#pragma unroll
        for (int j = 0; j < JF; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) rowsum[i][r] = fma(tj[j], cos_cw(acc[i][j][r]), rowsum[i][r]);
#pragma unroll
        for (int q = 0; q < NSUB; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double v = tjr[q] * cos_cw(accr[i][q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) rowsum[i][r] = fma(rmask[r], v, rowsum[i][r]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // the 16 lanes of a DPP row share the candidate row: quad butterflies, then the mirrors (no LDS round trips)
                double v = rowsum[i][r];
                v += dpp_f64<0xB1>(v);       // quad_perm [1, 0, 3, 2]
                v += dpp_f64<0x4E>(v);       // quad_perm [2, 3, 0, 1]
                v += dpp_f64<0x141>(v);      // row_half_mirror: quad 0 <-> 1, 2 <-> 3
                v += dpp_f64<0x140>(v);      // row_mirror: half 0 <-> 1
                const int64_t gm = m0 + w * 32 + i * 16 + fk + 4 * r;
                if (fr == 0 && gm < M) vals[(int64_t)tile * M + gm] = bias + v;
            }
        __syncthreads();   // the next tile's image is complete
    }
    if (clk && t == 0) {
        atomicAdd(clk, (unsigned long long)__builtin_readcyclecounter() - clk_c0);
        atomicAdd(clk + 1, (unsigned long long)wall_clock64() - clk_r0);
    }
}

// device staging layout for the MFMA path: [Wt S*nfb*dp*128][bt S*nfb*128][tt S*nfb*128]
// rows of the k-range the RFF kernels keep in LDS: the whole (padded) input dimension up to 64 coordinates (147 KB),
// 32 at a time beyond (74 KB: two workgroups per CU)
int g_rff_variant = 0;      // diagnostic (option "x_rff"): 0 = by size (round 5's kernel where it applies), 1 = round 3's single-buffer kernel everywhere (witness)
static int rff_k_chunk(int dp) { return dp <= DMAX_RFF_RESIDENT ? dp : 32; }

void launch_rff_mfma(hipStream_t s, const double* Wt, const double* bt, const double* tt, int S, int nfb, int n, int d,
                     int dp, double bias, const double* Xc, int64_t M, double* vals, unsigned long long* clk) {
    dim3 grid((unsigned)((M + TB - 1) / TB));
    const int dk = rff_k_chunk(dp);
    const size_t ldsb = (size_t)(2 * dk * LDT) * sizeof(double);
    // round 5: one feature tile per draw (n <= 128) and dp <= 32 -- the reference's defaults
    if (nfb == 1 && dp <= 32 && dk == dp && g_rff_variant == 0) {
        const int rem = n & 15;
        const int jf = (rem == 0 || rem > 8) ? (n + 15) / 16 : n / 16, nsub = (rem == 0 || rem > 8) ? 0 : (rem + 3) / 4;
        const bool db = dp <= 16;
        const size_t l5 = (size_t)((dp + (db ? 2 : 1) * (dp + 2)) * LDT) * sizeof(double);
#define GPX_RFF5(JF, NS)                                                                                                   \
        case (JF) * 4 + (NS):                                                                                              \
            if (db) {                                                                                                      \
                hipLaunchKernelGGL((k_rff_mfma5<JF, NS, true>), grid, dim3(GEMM_THREADS), l5, s, Wt, bt, tt, S, d, dp, bias, Xc, M, vals, clk); \
            } else {                                                                                                       \
                if (l5 > 64 * 1024) hipFuncSetAttribute((const void*)k_rff_mfma5<JF, NS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l5); \
                hipLaunchKernelGGL((k_rff_mfma5<JF, NS, false>), grid, dim3(GEMM_THREADS), l5, s, Wt, bt, tt, S, d, dp, bias, Xc, M, vals, clk); \
            }                                                                                                              \
            return;
        switch (jf * 4 + nsub) {
            GPX_RFF5(0, 1) GPX_RFF5(0, 2)
            GPX_RFF5(1, 0) GPX_RFF5(1, 1) GPX_RFF5(1, 2) GPX_RFF5(2, 0) GPX_RFF5(2, 1) GPX_RFF5(2, 2)
            GPX_RFF5(3, 0) GPX_RFF5(3, 1) GPX_RFF5(3, 2) GPX_RFF5(4, 0) GPX_RFF5(4, 1) GPX_RFF5(4, 2)
            GPX_RFF5(5, 0) GPX_RFF5(5, 1) GPX_RFF5(5, 2) GPX_RFF5(6, 0) GPX_RFF5(6, 1) GPX_RFF5(6, 2)
            GPX_RFF5(7, 0) GPX_RFF5(7, 1) GPX_RFF5(7, 2) GPX_RFF5(8, 0)
            default: break;
        }
#undef GPX_RFF5
    }
    // several feature tiles per draw (n > 128), or the witness (x_rff = 1): round 3's kernel
    if (dk == dp && dp <= 32) {
        hipLaunchKernelGGL(k_rff_mfma<true>, grid, dim3(GEMM_THREADS), ldsb, s, Wt, bt, tt, S, nfb, n, d, dp, dk, bias, Xc,
                           M, vals);
        return;
    }
    if (ldsb > 64 * 1024)
        hipFuncSetAttribute((const void*)k_rff_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipLaunchKernelGGL(k_rff_mfma<false>, grid, dim3(GEMM_THREADS), ldsb, s, Wt, bt, tt, S, nfb, n, d, dp, dk, bias, Xc, M,
                       vals);
}

// ------------------------------------------------------------------------------------------------
// Weight-posterior statistics for S draws at once:  A_s = C_s^T C_s (n x n),  v_s = C_s^T (y - bias),
// C_s = cos(X_obs W_s^T + b_s).  Three launches, all on the MFMA tile engine:
//   k_rff_phi     Phi[s][i][f] = cos(w_sf . x_i + b_sf) for f < n;  Phi[s][i][n] = y_i - bias (the residual
//                 rides along as one extra "feature" in the zero padding);  0 beyond; rows i >= N are 0.
//   k_rff_gram_sk split-K Gram of the (Np x 128) slab of each draw: partial[s][sp] = Phi_s[rows sp]^T Phi_s[..]
//                 (both operands are the same k-major matrix: A(m,k) = B(k,m) = Phi_s[k][m])
//   k_rff_gram_rd fixed-order reduction of the partials (deterministic) -> A (S,n,n), v (S,n)
// Requires n < 128 (the Thompson default is 100); larger n takes the per-draw path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_rff_phi(const double* __restrict__ Wt,
                                                             const double* __restrict__ bt, int S, int n, int d,
                                                             int dp, int dk, const double* __restrict__ X, int64_t N,
                                                             int64_t Np, const double* __restrict__ y,
                                                             double bias, double* __restrict__ Phi) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // At[dk][LDT] | Bt[dk][LDT]  (dk: see k_rff_mfma)
    double* At = lds;
    double* Bt = lds + dk * LDT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int64_t m0 = (int64_t)blockIdx.x * TB;
    const bool resident = (dk == dp);
    auto load_a = [&](int k0, int kc) {
        for (int e = t; e < TB * kc; e += GEMM_THREADS) {
            const int k = e >> 7, m = e & 127;
            const int64_t gm = m0 + m;
            At[k * LDT + m] = (k0 + k < d && gm < N) ? X[gm * d + k0 + k] : 0.0;
        }
    };
    if (resident) load_a(0, dp);
    const int fr = lane & 15, fk = lane >> 4;
    // blockIdx.y strides over the draws: with Np / 128 workgroups alone (128 at N = 16384, 64 at 8192) half the chip or more sat
    // idle while each workgroup walked all S draws (1.0 ms for 1 GB of Phi at config D); a draw's tile does not depend on the others
    for (int s = blockIdx.y; s < S; s += gridDim.y) {
        d4 acc[4][4];
        acc_zero(acc);
        const double* as = At + wm * 64 + fr;
        const double* bs = Bt + wn * 64 + fr;
        for (int k0 = 0; k0 < dp; k0 += dk) {
            const int kc = min(dk, dp - k0);
            if (!resident) load_a(k0, kc);
            const double* Wtile = Wt + ((int64_t)s * dp + k0) * TB;
            for (int e = t; e < (TB / 2) * kc; e += GEMM_THREADS) {
                const int k = e >> 6, c = (e & 63) * 2;
                *reinterpret_cast<d2*>(Bt + k * LDT + c) = *reinterpret_cast<const d2*>(Wtile + (int64_t)k * TB + c);
            }
            __syncthreads();
            for (int kk = 0; kk < kc / 4; ++kk) {
                const int kr = kk * 4 + fk;
                double a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = as[kr * LDT + i * 16];
                    b[i] = bs[kr * LDT + i * 16];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (k0 + dk < dp) __syncthreads();
        }
        double* out = Phi + (int64_t)s * Np * TB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = wn * 64 + j * 16 + fr;
            const double bj = bt[(int64_t)s * TB + f];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = m0 + wm * 64 + i * 16 + fk + 4 * r;
                    double v = 0.0;
                    if (gi < N) v = (f < n) ? cos_cw(acc[i][j][r] + bj) : ((f == n) ? y[gi] - bias : 0.0);
                    out[gi * TB + f] = v;
                }
        }
        __syncthreads();
    }
}

constexpr int RFF_SPLIT = 16;

// grid (RFF_SPLIT, S): partial Gram over rows [sp*rows, (sp+1)*rows) of draw s
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_rff_gram_sk(const double* __restrict__ Phi, int64_t Np,
                                                                 double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    const int sp = blockIdx.x, s = blockIdx.y;
    const int64_t rows = (Np / TB + RFF_SPLIT - 1) / RFF_SPLIT * TB;   // multiple of 128 (hence of BK)
    const int64_t k_lo = (int64_t)sp * rows;
    const int64_t k_hi = (k_lo + rows < Np) ? k_lo + rows : Np;
    const double* P = Phi + (int64_t)s * Np * TB;
    d4 acc[4][4];
    acc_zero(acc);
    if (k_lo < k_hi) gemm_tile_128_b<false>(acc, P, TB, P, TB, (int)k_lo, (int)k_hi, smem);
    double* out = part + ((int64_t)s * RFF_SPLIT + sp) * TB * TB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[acc_row(i, r) * TB + acc_col(j)] = acc[i][j][r];
}

// one thread per (s, j1, j2 <= n): A[s][j1][j2] for j2 < n, v[s][j1] for j2 == n
__global__ __launch_bounds__(256) void k_rff_gram_rd(const double* __restrict__ part, int S, int n,
                                                     double* __restrict__ A, double* __restrict__ v) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)n * (n + 1);
    if (idx >= per * S) return;
    const int s = (int)(idx / per);
    const int64_t r = idx - (int64_t)s * per;
    const int j1 = (int)(r / (n + 1)), j2 = (int)(r - (int64_t)j1 * (n + 1));
    double acc = 0.0;
    for (int sp = 0; sp < RFF_SPLIT; ++sp)
        acc += part[(((int64_t)s * RFF_SPLIT + sp) * TB + j1) * TB + j2];
    if (j2 < n) A[((int64_t)s * n + j1) * n + j2] = acc;
    else v[(int64_t)s * n + j1] = acc;
}

// scratch: Phi (S*Np*128) | part (S*RFF_SPLIT*128*128)
// ------------------------------------------------------------------------------------------------
// Weight posterior of the random-feature model, one workgroup per draw, everything in LDS:
//     B = sc^2 A + sn2 I = L L^T,     theta = sc L^-T ( L^-1 (sc v) + sqrt(sn2) z )
// (= sc (B^-1 sc v + sqrt(sn2) L^-T z): posterior mean + a N(0, sn2 B^-1) draw -- the n x n solve inside
// model.sample_f(n, rng), pybo/policies/simple.py:48).  Right-looking Cholesky on the AUGMENTED lower triangle
// [B; (sc v)^T]: the extra row leaves the loop as L^-1 (sc v), so only the back substitution follows.
// n <= 127 (the batched feature path), row pitch odd => conflict-free column walks.  2n + n barriers.
// ------------------------------------------------------------------------------------------------
constexpr int RFP_NMAX = 127;
__global__ __launch_bounds__(256) void k_rff_posterior(const double* __restrict__ A, const double* __restrict__ v,
                                                       const double* __restrict__ z, int n, double sc, double sn2,
                                                       double* __restrict__ theta, int* __restrict__ flag) {
    __shared__ double Bm[(RFP_NMAX + 1) * (RFP_NMAX + 2)];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int ld = (n + 1) | 1;
    const int64_t q = blockIdx.x;
    A += q * n * n;
    v += q * n;
    z += q * n;
    theta += q * n;
    for (int e = t; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        if (j <= i) Bm[i * ld + j] = sc * sc * A[e] + ((i == j) ? sn2 : 0.0);
    }
    for (int j = t; j < n; j += 256) Bm[n * ld + j] = sc * v[j];
    bool failed = false;
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const double piv = Bm[j * ld + j];
        if (!(piv > 0.0) || !(piv < 1.0e300)) {          // uniform: every thread reads the same value ...
            if (t == 0) atomicCAS(flag, 0, j + 1);
            failed = true;                               // ... and leaves on its own copy of the verdict
            break;
        }
        const double rinv = 1.0 / sqrt(piv);
        for (int i = j + 1 + t; i <= n; i += 256) Bm[i * ld + j] *= rinv;
        __syncthreads();
        if (t == 0) Bm[j * ld + j] = piv * rinv;
        for (int i = j + 1 + ty; i <= n; i += 16) {
            const double li = Bm[i * ld + j];
            const int khi = (i < n) ? i : n - 1;         // the augmented row has no diagonal entry
            for (int k = j + 1 + tx; k <= khi; k += 16) Bm[i * ld + k] = fma(-li, Bm[k * ld + j], Bm[i * ld + k]);
        }
        __syncthreads();
    }
    if (failed) return;
    // w = L^-1 (sc v) + sqrt(sn2) z, in place in the augmented row; then theta = sc L^-T w
    const double sn = sqrt(sn2);
    for (int j = t; j < n; j += 256) Bm[n * ld + j] = fma(sn, z[j], Bm[n * ld + j]);
    for (int j = n - 1; j >= 0; --j) {
        __syncthreads();
        const double tj = Bm[n * ld + j] / Bm[j * ld + j];
        for (int k = t; k < j; k += 256) Bm[n * ld + k] = fma(-Bm[j * ld + k], tj, Bm[n * ld + k]);
        if (t == 0) theta[j] = sc * tj;
    }
}

void launch_rff_posterior(hipStream_t s, const double* A, const double* v, const double* z, int S, int n, double sc,
                          double sn2, double* theta, int* flag) {
    hipLaunchKernelGGL(k_rff_posterior, dim3((unsigned)S), dim3(256), 0, s, A, v, z, n, sc, sn2, theta, flag);
}

int64_t rff_gram_batch_scratch(int64_t S, int64_t Np) { return S * Np * TB + S * RFF_SPLIT * TB * TB; }

void launch_rff_gram_batch(hipStream_t s, const double* Xraw, int64_t N, int64_t Np, int d, int dp,
                           const double* Wt, const double* bt, int S, int n, const double* y, double bias,
                           double* scratch, double* A, double* v) {
    double* Phi = scratch;
    double* part = scratch + (int64_t)S * Np * TB;
    const int dk = rff_k_chunk(dp);
    const size_t ldsb = (size_t)(2 * dk * LDT) * sizeof(double);
    if (ldsb > 64 * 1024)
        hipFuncSetAttribute((const void*)k_rff_phi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const unsigned ny = (unsigned)std::max<int64_t>(1, std::min<int64_t>(S, 2048 / std::max<int64_t>(1, Np / TB)));
    hipLaunchKernelGGL(k_rff_phi, dim3((unsigned)(Np / TB), ny), dim3(GEMM_THREADS), ldsb, s, Wt, bt, S, n, d, dp, dk,
                       Xraw, N, Np, y, bias, Phi);
    hipLaunchKernelGGL(k_rff_gram_sk, dim3(RFF_SPLIT, (unsigned)S), dim3(GEMM_THREADS), 0, s, Phi, Np, part);
    const int64_t outs = (int64_t)S * n * (n + 1);
    hipLaunchKernelGGL(k_rff_gram_rd, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, s, part, S, n, A, v);
}

// ---- per-draw path (any n): Ft[j][i] = cos(w_j . x_i + b_j) for observed points i < N (0 beyond)
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
