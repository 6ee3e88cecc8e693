// kernels_sweep.hip -- the acquisition sweep on gfx950.
//
// Replaces, for M candidates at once, what pybo reaches through `finit = f(xgrid, grad=False)`
// [pybo/solvers/lbfgs.py:50] -> index(X) [pybo/policies/simple.py:23-25,37-39,62-73] ->
// model.predict / get_improvement / get_tail (reggie, un-vendored), followed by
// `np.argsort(finit)[::-1]` [pybo/solvers/lbfgs.py:51] of which only the first nbest are used.
//
// Per chunk of candidate columns:
//   1. k_cross_gram   Ks[nt][k][c] = k(x_k, cand_{128 nt + c})    (HBM-write bound; tile-blocked so that every
//                     128-candidate column panel is one contiguous Np x 128 block)
//   2. k_sweep_trmm*  V = T Ks on fp64 MFMA, tile (mt, nt); V never leaves registers: the epilogue
//                     reduces colsum(V^2) and V^T a per 128-row block into Qp/Pp[mt][n]   (default: k_sweep_trmm_l -- operands by
//                     LDS-DMA, the zero rows of T's diagonal block skipped; the other schedules are bit-identical witnesses)
//   3. k_acq          q = sum_mt Qp, p = sum_mt Pp (fixed order -> deterministic),
//                     mu = bias + p, s2 = max(rho - q, 1e-100), acquisition value
// then one block-local + one merge top-k pass over all M values.
#include "gemm_core.h"
#include "gpx_internal.h"
#include "gpx_math.h"

namespace gpx {

// ------------------------------------------------------------------------------------------------
// cross-Gram: tile 64 observed rows x 128 candidate columns, 8x4 outputs per thread
// ------------------------------------------------------------------------------------------------
constexpr int XK = 64, XN = 128, XDC = 16;

// Staging of a (rows x kc) block of row-major coordinates into LDS as [k][row] WITHOUT index divisions: a thread owns
// one row and every STRIDE-th coordinate of it (counters: the e / kc, e % kc form cost ~13 of 50-58 VALU instructions
// per covariance evaluation at d = 8 -- these kernels are VALU-issue-bound).
// Squared scaled distances of an 8 x 4 block per thread, one dimension at a time; the first dimension starts the sums
// (df * df == fma(df, df, +0): same bits as a zero-initialised accumulator).
template <bool FIRST>
__device__ __forceinline__ void dist_step(const double (*xo)[XK], const double (*xc)[XN], int k, int ty, int tx,
                                          double (&r2)[8][4]) {
    double a8[8], b4[4];
#pragma unroll
    for (int a = 0; a < 8; ++a) a8[a] = xo[k][ty * 8 + a];
#pragma unroll
    for (int b = 0; b < 4; ++b) b4[b] = xc[k][tx * 4 + b];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double df = a8[a] - b4[b];
            r2[a][b] = FIRST ? df * df : fma(df, df, r2[a][b]);
        }
}

// One workgroup: 128 candidates x xrt consecutive 64-row tiles (the candidates' coordinates are staged once when
// d <= XDC).  grid (ceil(Np / 64 / xrt), cols / 128): consecutive workgroups write consecutive runs of Ks.
constexpr int XRT = 8;

template <int KID>
__global__ __launch_bounds__(256, 3) void k_cross_gram(const double* __restrict__ Xs, int64_t N, int64_t Np, int d,
                                                       const double* __restrict__ Xc, int64_t m0, int64_t M,
                                                       const double* __restrict__ invell, double rho,
                                                       double* __restrict__ Ks, int64_t ldk, int xrt) {
    __shared__ double xo[XDC][XK];
    __shared__ double xc[XDC][XN];
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int crow = t & (XN - 1), ckq = t >> 7;      // staging of candidates: row, first coordinate (stride 2)
    const int orow = t & (XK - 1), okq = t >> 6;      // staging of observed rows: row, first coordinate (stride 4)
    const int64_t n0 = (int64_t)blockIdx.y * XN;      // chunk-local candidate origin
    const int64_t gmc = m0 + n0 + crow;
    const bool onepass = (d <= XDC);
    if (onepass)
        for (int k = ckq; k < d; k += 2) xc[k][crow] = (gmc < M) ? Xc[gmc * d + k] * invell[k] : 0.0;
    for (int rt = 0; rt < xrt; ++rt) {
        const int64_t k0 = ((int64_t)blockIdx.x * xrt + rt) * XK;   // observed row origin
        if (k0 >= Np) break;
        double r2[8][4];
        auto stage = [&](int c0, int kc) {
            __syncthreads();
            for (int k = okq; k < kc; k += 4) xo[k][orow] = Xs[(k0 + orow) * d + c0 + k];
            if (!onepass)
                for (int k = ckq; k < kc; k += 2)
                    xc[k][crow] = (gmc < M) ? Xc[gmc * d + c0 + k] * invell[c0 + k] : 0.0;
            __syncthreads();
        };
        {
            const int kc = min(XDC, d);
            stage(0, kc);
            dist_step<true>(xo, xc, 0, ty, tx, r2);
#pragma unroll 1
            for (int k = 1; k < kc; ++k) dist_step<false>(xo, xc, k, ty, tx, r2);
        }
        for (int c0 = XDC; c0 < d; c0 += XDC) {
            const int kc = min(XDC, d - c0);
            stage(c0, kc);
#pragma unroll 1
            for (int k = 0; k < kc; ++k) dist_step<false>(xo, xc, k, ty, tx, r2);
        }
        // a tile without padding rows / columns (every tile of the north-star launch) needs no selects: two of the 58 VALU
        // instructions per entry of a kernel that is bound by their issue (workgroup-uniform branch)
        const bool full = (k0 + XK <= N) && (m0 + n0 + XN <= M);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int64_t gk = k0 + ty * 8 + a;
            d4 o;
            if (full) {
#pragma unroll
                for (int b = 0; b < 4; ++b) o[b] = kern_eval(KID, r2[a][b], rho);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int64_t gm = m0 + n0 + tx * 4 + b;
                    o[b] = (gk < N && gm < M) ? kern_eval(KID, r2[a][b], rho) : 0.0;
                }
            }
            // tile-blocked layout [nt][k][128]: the 64 x 128 outputs of a row tile are ONE contiguous 64 KB run
            *reinterpret_cast<d4*>(Ks + ((int64_t)blockIdx.y * ldk + gk) * XN + tx * 4) = o;
        }
    }
}

void launch_cross_gram(hipStream_t s, const double* Xs, int64_t Np, int64_t N, int d, const double* Xc,
                       int64_t m0, int64_t M, int64_t cols, const double* invell, int kernel_id,
                       double rho, double* Ks, int64_t ldk) {
    const int64_t tiles = Np / XK;
    // row tiles per workgroup: the SE kernel at small d is bound by its 8 B/evaluation of stores (one tile per
    // workgroup interleaves them best: 3.5 vs 4.3 ms per 2^31 evaluations), the others by their VALU work (the
    // candidates staged once per 8 tiles: Matern-5/2 4.35 vs 4.85 ms); non-temporal stores change nothing
    const int XRTa = (kernel_id == GPX_KERN_SE_ARD && d <= XDC) ? 1 : XRT;       // (re-measured in round 6: SE 14.3 / 15.5 / 17.5 ms per 2^20 candidates with 1 / 2 / 4 tiles)
    const int XRTv = XRTa;
    dim3 grid((unsigned)((tiles + XRTa - 1) / XRTa), (unsigned)(cols / XN));   // x: row groups of one candidate tile = one contiguous run of Ks
#define GPX_CG(KID) \
    hipLaunchKernelGGL(k_cross_gram<KID>, grid, dim3(256), 0, s, Xs, N, Np, d, Xc, m0, M, invell, rho, Ks, ldk, XRTv)
    switch (kernel_id) {
        case GPX_KERN_SE_ARD: GPX_CG(GPX_KERN_SE_ARD); break;
        case GPX_KERN_MATERN52: GPX_CG(GPX_KERN_MATERN52); break;
        case GPX_KERN_MATERN32: GPX_CG(GPX_KERN_MATERN32); break;
        default: GPX_CG(GPX_KERN_MATERN12); break;
    }
#undef GPX_CG
}

// ------------------------------------------------------------------------------------------------
// THE dominant kernel: V(m,n) = sum_{k <= m} T(m,k) Ks(k,n),  T(m,k) = U[k][m]  (both k-major).
// Tile (mt, nt): 128 observed rows x 128 candidates, K-extent (mt+1)*128 (T is lower triangular),
// N^2 * M flop in total.  Heavy tiles (large mt) are dispatched first.
// ------------------------------------------------------------------------------------------------
// blockIdx -> tile(s) of the launch.  Returns false when the block has nothing to do; mt2 >= 0: the workgroup also
// computes tile (mt2, nt) afterwards.  RES = workgroups resident per XCD (32 CUs x workgroups per CU): the size of a super-tile.
template <int RES>
__device__ __forceinline__ bool sweep_tile_of(int b, int order, int sm, int NT, int nP, int& mt, int& nt, int& mt2) {
    mt2 = -1;
    if (order == 1) {
        // XCD-aware: block b runs on XCD b%8 (observed, speed only).  Give each XCD its own
        // contiguous slice of candidate tiles so the tiles resident on one XCD walk the
        // SAME mt (shared T rows in that XCD's L2) over neighbouring nt.
        const int x = b & 7, q = b >> 3;          // q-th block of XCD x
        const int per = (NT + 7) / 8;             // candidate tiles per XCD
        const int lm = q / per, ln = q - lm * per;
        mt = nP - 1 - lm;
        nt = x * per + ln;
        return !(nt >= NT || mt < 0);
    }
    if (order == 2 || order == 3) {
        // XCD-aware 2-D super-tiles: the RES workgroups resident on one XCD form an sm (mt) x SN (nt) patch, so every
        // T row-panel and every Ks column-panel fetched into that XCD's L2 is used by several tiles.
        // order 3: PAIRED tiles on the super-tile map: the workgroup computes (nP-1-i, nt) and then (i, nt), so every
        // workgroup of the launch does the same (nP+1)*128 of K.  Equal durations keep the workgroups
        // of a super-tile in step for the whole launch: tiles that share a Ks column panel (same nt,
        // different mt) walk k together instead of drifting apart by their K-extent difference.
        const int x = b & 7, q = b >> 3;
        const int SN = RES / sm;                  // super-tile = sm (mt) x SN (nt) = RES workgroups
        const int per = (NT + 7) / 8;             // candidate tiles per XCD (contiguous slice)
        const int hper = (per + SN - 1) / SN;     // n-groups per XCD
        const int s = q / RES, r = q - s * RES;
        const int G = s / hper, H = s - G * hper;
        const int i = G * sm + r / SN;
        const int ln = H * SN + (r - (r / SN) * SN);
        nt = x * per + ln;
        mt = nP - 1 - i;
        if (order == 2) return !(ln >= per || nt >= NT || mt < 0);
        if (ln >= per || nt >= NT || i > mt) return false;
        if (i < mt) mt2 = i;
        return true;
    }
    mt = nP - 1 - b / NT;
    nt = b - (b / NT) * NT;
    return true;
}

template <int RES>
static unsigned sweep_grid(int order, int super_m, int NT, int nP) {
    const int per = (NT + 7) / 8;
    if (order == 1) return (unsigned)(8 * per * nP);
    if (order == 2 || order == 3) {
        const int SN = RES / super_m;
        const int hper = (per + SN - 1) / SN;
        const int rows = (order == 3) ? (nP + 1) / 2 : nP;
        const int gm = (rows + super_m - 1) / super_m;
        return (unsigned)(8 * RES * hper * gm);
    }
    return (unsigned)(NT * nP);
}

// The summation order of a tile along k (the SAME in every schedule and every tile map: results stay bit-identical): tiles of the
// lower half, 2 mt < nP - 1 -- in the paired map exactly the SECOND tile of every pair -- take their 32-row k-steps downwards when
// the factor has at least 32 block rows.  In the paired map every workgroup of a super-tile then reads the same rows of its Ks
// panel at the same time in BOTH phases (row t in the first, row (nP + 1) 128 - t in the second, whatever its pair index), which
// lets an XCD's L2 serve the panel once: L2 -> fabric reads 101 -> 88 GB per launch at N = 8192 and, the kernel being power-bound,
// 0.8 % more clock on the boxes that sustain 2300 MHz (nothing on those at 2385) -- profiles/r06_sweep_power_probes.txt.  Below 32
// block rows the second tiles are short and it costs 0.3 %.
__device__ __forceinline__ bool sweep_tile_rev(int mt, int nP) { return nP >= 32 && 2 * mt < nP - 1; }

// epilogue of a tile: column sums of V^2 and V*a over its 128 rows -> Qp / Pp[mt][n0 ..] (red: 512 doubles of LDS, free)
template <bool ILV = false>
__device__ __forceinline__ void sweep_epilogue(const d4 (&acc)[4][4], const double* __restrict__ avec, int64_t m0,
                                               double* __restrict__ Qrow, double* __restrict__ Prow, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    double qs[4] = {0.0, 0.0, 0.0, 0.0}, ps[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = avec[m0 + (ILV ? (2 * i + wm) * 16 : wm * 64 + i * 16) + (lane >> 4) + 4 * r];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = acc[i][j][r];
                qs[j] = fma(v, v, qs[j]);
                ps[j] = fma(v, av[r], ps[j]);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double q = qs[j], p = ps[j];
        q += __shfl_xor(q, 16);
        p += __shfl_xor(p, 16);
        q += __shfl_xor(q, 32);
        p += __shfl_xor(p, 32);
        qs[j] = q;
        ps[j] = p;
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wn * 64 + j * 16 + lane;
            red[(wm * TB + c) * 2 + 0] = qs[j];
            red[(wm * TB + c) * 2 + 1] = ps[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < TB) {
        const int c = threadIdx.x;
        Qrow[c] = red[c * 2] + red[(TB + c) * 2];
        Prow[c] = red[c * 2 + 1] + red[(TB + c) * 2 + 1];
    }
}

template <int VAR>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_sweep_trmm(const double* __restrict__ U, int64_t Np,
                                                                const double* __restrict__ Ks,
                                                                int64_t ldk, int NT,
                                                                const double* __restrict__ avec,
                                                                double* __restrict__ Qp,
                                                                double* __restrict__ Pp, int64_t ldp,
                                                                int order, int sm, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    // the sustained shader clock of this launch: every workgroup adds its lifetime in s_memtime ticks (shader clocks) and in
    // s_memrealtime ticks (100 MHz) to two counters -- the bench line's roofline.frac_at_measured_clock (two atomics per
    // workgroup of ~1.8 ms)
    const unsigned long long clk_c0 = __builtin_readcyclecounter();
    const unsigned long long clk_r0 = wall_clock64();
    const int nP = (int)(Np / TB);
    int mt, nt, mt2;
    if (!sweep_tile_of<64>(blockIdx.x, order, sm, NT, nP, mt, nt, mt2)) return;
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {
    if (ph == 1) {
        if (mt2 < 0) break;
        mt = mt2;
        __syncthreads();               // the epilogue's LDS reads are done before the next tile stages
    }
    const int64_t m0 = (int64_t)mt * TB, n0 = (int64_t)nt * TB;
    d4 acc[4][4];
    acc_zero(acc);
    // (interleaved row blocks in every schedule: the column sums then add a tile's rows in the same order everywhere)
    const double* At = U + m0;
    const double* Bt = Ks + (int64_t)nt * Np * TB;
    const int kend = (mt + 1) * TB;
    if (!sweep_tile_rev(mt, nP)) {
        if (VAR == 2) gemm_tile_128_b<true, true>(acc, At, Np, Bt, TB, 0, kend, smem);
        else if (VAR == 5) gemm_tile_128_g<1, false, true>(acc, At, Np, Bt, TB, 0, kend, smem);
        else gemm_tile_128_s<1, false, true>(acc, At, Np, Bt, TB, 0, kend, smem);
    } else {
        if (VAR == 2) gemm_rev32(0, kend, [&](int k0, int k1) { gemm_tile_128_b<true, true>(acc, At, Np, Bt, TB, k0, k1, smem); });
        else if (VAR == 5) gemm_tile_128_g<1, false, true, true>(acc, At, Np, Bt, TB, 0, kend, smem);
        else gemm_tile_128_s<1, false, true, true>(acc, At, Np, Bt, TB, 0, kend, smem);
    }
    // the k-loop ended on a barrier, LDS is free
    sweep_epilogue<true>(acc, avec, m0, Qp + (int64_t)mt * ldp + n0, Pp + (int64_t)mt * ldp + n0, smem);
  }
    if (clk && threadIdx.x == 0) {
        atomicAdd(clk, (unsigned long long)__builtin_readcyclecounter() - clk_c0);
        atomicAdd(clk + 1, (unsigned long long)wall_clock64() - clk_r0);
    }
}

// The same tiles through the register-free k-loop (gemm_tile_128_l): WGS workgroups per compute unit.
// TRI: the all-zero quarter-rows of T's diagonal block are skipped (1.5 of 4 k-steps of every tile).
// DOWN = false: every tile upwards (probe builds only: the A/B of the order rule -- NOT bit-identical with the shipped schedules)
template <int BKL, int WGS, int PRIO, int NSET, bool TRI, int AUX = 0, bool DOWN = true>
__global__ __launch_bounds__(GEMM_THREADS, WGS) void k_sweep_trmm_l(const double* __restrict__ U, int64_t Np,
                                                                    const double* __restrict__ Ks, int64_t ldk, int NT,
                                                                    const double* __restrict__ avec,
                                                                    double* __restrict__ Qp, double* __restrict__ Pp,
                                                                    int64_t ldp, int order, int sm,
                                                                    unsigned long long* clk, int b0) {
    __shared__ __attribute__((aligned(16))) double smem[gemm_l_lds_f64<BKL>()];
    const unsigned long long clk_c0 = __builtin_readcyclecounter();
    const unsigned long long clk_r0 = wall_clock64();
    const int nP = (int)(Np / TB);
    int mt, nt, mt2;
    if (!sweep_tile_of<32 * WGS>((int)blockIdx.x + b0, order, sm, NT, nP, mt, nt, mt2)) return;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        if (ph == 1) {
            if (mt2 < 0) break;
            mt = mt2;
            __syncthreads();               // the epilogue's LDS reads are done before the next tile's rows land
        }
        const int64_t m0 = (int64_t)mt * TB, n0 = (int64_t)nt * TB;
        d4 acc[4][4];
        acc_zero(acc);
        const double* At = U + m0;
        const double* Bt = Ks + (int64_t)nt * Np * TB;
        const int kend = (mt + 1) * TB;
        const bool rev = DOWN && sweep_tile_rev(mt, nP);
        if (!rev) gemm_tile_128_l<BKL, PRIO, NSET, true, TRI, false, AUX>(acc, At, Np, Bt, TB, 0, kend, smem);
        else if constexpr (BKL == 32) gemm_tile_128_l<32, PRIO, NSET, true, TRI, false, AUX, true>(acc, At, Np, Bt, TB, 0, kend, smem);
        else gemm_rev32(0, kend, [&](int k0, int k1) { gemm_tile_128_l<BKL, PRIO, NSET, true, false, false, AUX>(acc, At, Np, Bt, TB, k0, k1, smem); });
        sweep_epilogue<true>(acc, avec, m0, Qp + (int64_t)mt * ldp + n0, Pp + (int64_t)mt * ldp + n0, smem);
    }
    if (clk && threadIdx.x == 0) {
        atomicAdd(clk, (unsigned long long)__builtin_readcyclecounter() - clk_c0);
        atomicAdd(clk + 1, (unsigned long long)wall_clock64() - clk_r0);
    }
}

// The same tiles on the barrier-free loop (gemm_tile_128_w: every wave fetches its own operand halves, one 16-row image per
// wave, two workgroups per CU), with the diagonal block's zero rows skipped.  Twice the L2 -> LDS bytes of k_sweep_trmm_l.
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_sweep_trmm_w(const double* __restrict__ U, int64_t Np,
                                                                  const double* __restrict__ Ks, int64_t ldk, int NT,
                                                                  const double* __restrict__ avec, double* __restrict__ Qp,
                                                                  double* __restrict__ Pp, int64_t ldp, int order, int sm,
                                                                  unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) double smem[4 * GEMM_W_IMG_F64];
    const unsigned long long clk_c0 = __builtin_readcyclecounter();
    const unsigned long long clk_r0 = wall_clock64();
    const int nP = (int)(Np / TB);
    int mt, nt, mt2;
    if (!sweep_tile_of<64>(blockIdx.x, order, sm, NT, nP, mt, nt, mt2)) return;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        if (ph == 1) {
            if (mt2 < 0) break;
            mt = mt2;
            __syncthreads();
        }
        const int64_t m0 = (int64_t)mt * TB, n0 = (int64_t)nt * TB;
        d4 acc[4][4];
        acc_zero(acc);
        const double* At = U + m0;
        const double* Bt = Ks + (int64_t)nt * Np * TB;
        const int kend = (mt + 1) * TB;
        if (!sweep_tile_rev(mt, nP)) gemm_tile_128_w<1, 1, false, true, 0, true>(acc, At, Np, Bt, TB, 0, kend, smem);
        else gemm_rev32(0, kend, [&](int k0, int k1) { gemm_tile_128_w<1, 1, false, true, 0, false>(acc, At, Np, Bt, TB, k0, k1, smem); });
        sweep_epilogue<true>(acc, avec, m0, Qp + (int64_t)mt * ldp + n0, Pp + (int64_t)mt * ldp + n0, smem);
    }
    if (clk && threadIdx.x == 0) {
        atomicAdd(clk, (unsigned long long)__builtin_readcyclecounter() - clk_c0);
        atomicAdd(clk + 1, (unsigned long long)wall_clock64() - clk_r0);
    }
}

void launch_sweep_trmm(hipStream_t s, const double* U, int64_t Np, const double* Ks, int64_t ldk,
                       int64_t cols, const double* a, double* Qp, double* Pp, int64_t ldp,
                       int tile_order, int super_m, unsigned long long* clk) {
    const int NT = (int)(cols / TB);
    const int nP = (int)(Np / TB);
    // bits 0-1: tile map, bits 2-4: k-loop (4 = default: operands by LDS-DMA, k-step 32, two workgroups per CU, the zero rows of
    // T's diagonal block skipped; 3: the same without the skip; 1: the barrier-free wave-private loop; 7: k-step 16, three workgroups per CU; 6, 5, 2: the
    // register-staged schedules of rounds 5, 2, 1 -- all kept as independently scheduled witnesses of the bit-identity test)
    const int order = tile_order & 3, var = (tile_order >> 2) & 7;
#define GPX_SW(K) hipLaunchKernelGGL(K, dim3(nblk), dim3(GEMM_THREADS), 0, s, U, Np, Ks, ldk, NT, a, Qp, Pp, ldp, order, super_m, clk)
#define GPX_SWL(K) hipLaunchKernelGGL(K, dim3(nblk), dim3(GEMM_THREADS), 0, s, U, Np, Ks, ldk, NT, a, Qp, Pp, ldp, order, super_m, clk, 0)
    if (var == 7) {                // three workgroups per CU, k-step 16
        const unsigned nblk = sweep_grid<96>(order, super_m, NT, nP);
        GPX_SWL((k_sweep_trmm_l<16, 3, 1, 2, false>));
    } else if (var == 1) {               // barrier-free: every wave keeps its own operands
        const unsigned nblk = sweep_grid<64>(order, super_m, NT, nP);
        GPX_SW(k_sweep_trmm_w);
    } else if (var == 4 || var == 3) {   // two workgroups per CU, k-step 32; 4: with the diagonal block's zero rows skipped
        const unsigned nblk = sweep_grid<64>(order, super_m, NT, nP);
#ifdef GPX_SWEEP_PROBES          // scripts/probe/sweep_ab.hip only: variants that are NOT schedules of the library (profiles/r06_sweep_power_probes.txt)
        const int aux = tile_order >> 5;         // (probe builds only: cache policy of the operand loads; gpx_set_option admits 0)
        if (var == 4 && aux == 1) GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, true, 1>));
        else if (var == 4 && aux == 2) GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, true, 2>));
        else if (var == 4 && aux == 3) GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, true, 16>));
        else if (var == 4 && aux == 4) GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, true, 17>));
        else if (var == 4 && aux == 5) GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, true, 3>));
        else if (var == 4 && aux == 6) GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, true, 0, false>));      // every tile upwards (round 6's first form)
        else if (var == 4 && aux == 7) {        // probe: one launch per generation of 512 workgroups (every generation starts aligned)
            for (unsigned g0 = 0; g0 < nblk; g0 += 512)
                hipLaunchKernelGGL((k_sweep_trmm_l<32, 2, 1, 2, true>), dim3(nblk - g0 < 512 ? nblk - g0 : 512), dim3(GEMM_THREADS), 0, s, U, Np, Ks, ldk, NT, a,
                                   Qp, Pp, ldp, order, super_m, clk, (int)g0);
        }
        else
#endif
        if (var == 4) GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, true>));
        else GPX_SWL((k_sweep_trmm_l<32, 2, 1, 2, false>));
    } else {
        const unsigned nblk = sweep_grid<64>(order, super_m, NT, nP);
        if (var == 2) GPX_SW(k_sweep_trmm<2>);
        else if (var == 6) GPX_SW(k_sweep_trmm<6>);
        else GPX_SW(k_sweep_trmm<5>);
    }
#undef GPX_SW
#undef GPX_SWL
}

// ------------------------------------------------------------------------------------------------
// acquisition values from the reduced partials (one thread per candidate; coalesced over n)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }
__device__ __forceinline__ double norm_pdf(double z) {
    return 0.39894228040143267794 * exp(-0.5 * z * z);
}

// nrb > 0: Qp/Pp are per-row-block partials (nrb, ldp) of this chunk, reduced here in block order.
// nrb = 0: Qp/Pp are the already reduced per-candidate sums of the whole grid (the sweep cache), m0 = 0.
// qsum/psum (optional): the reduced sums are stored per candidate -- the state gpx_append's rank-1 correction
// keeps current (warm BO step).
__global__ __launch_bounds__(256) void k_acq(const double* __restrict__ Qp, const double* __restrict__ Pp,
                                             int64_t ldp, int nrb, int64_t m0, int64_t cols_valid,
                                             double rho, double bias, int acq_id, double p0,
                                             double* __restrict__ acq_out, double* __restrict__ mu_out,
                                             double* __restrict__ s2_out, double* __restrict__ qsum,
                                             double* __restrict__ psum) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= cols_valid) return;
    double q = 0.0, p = 0.0;
    if (nrb == 0) {
        q = Qp[n];
        p = Pp[n];
    }
    for (int rb = 0; rb < nrb; ++rb) {
        q += Qp[(int64_t)rb * ldp + n];
        p += Pp[(int64_t)rb * ldp + n];
    }
    if (qsum) {
        qsum[m0 + n] = q;
        psum[m0 + n] = p;
    }
    const double mu = bias + p;
    const double s2 = fmax(rho - q, 1e-100);
    double val;
    switch (acq_id) {
        case GPX_ACQ_EI: {
            const double s = sqrt(s2);
            const double dlt = mu - p0;
            const double z = dlt / s;
            val = dlt * norm_cdf(z) + s * norm_pdf(z);
            break;
        }
        case GPX_ACQ_PI: {
            const double z = (mu - p0) / sqrt(s2);
            val = norm_cdf(z);
            break;
        }
        case GPX_ACQ_UCB:
            val = mu + sqrt(p0 * s2);
            break;
        default:
            val = mu;
    }
    acq_out[m0 + n] = val;
    if (mu_out) mu_out[m0 + n] = mu;
    if (s2_out) s2_out[m0 + n] = s2;
}

void launch_acq(hipStream_t s, const double* Qp, const double* Pp, int64_t ldp, int nrb, int64_t m0,
                int64_t cols_valid, double rho, double bias, int acq_id, double p0, double* acq_out,
                double* mu_out, double* s2_out, double* qsum, double* psum) {
    const unsigned g = (unsigned)((cols_valid + 255) / 256);
    hipLaunchKernelGGL(k_acq, dim3(g), dim3(256), 0, s, Qp, Pp, ldp, nrb, m0, cols_valid, rho, bias,
                       acq_id, p0, acq_out, mu_out, s2_out, qsum, psum);
}

// ------------------------------------------------------------------------------------------------
// Warm BO step: correction of the cached per-candidate sums after q appended observations (q <= Q).
// With w_j = K_j^-1 k(X_j, x_j) (K_j, X_j: the model just before point j was appended), d_j the posterior std of
// observation j (incl. noise) and a_j the new entry of a:
//     v_jn = ( k(x_j, z_n) - sum_{i < N_j} w_ji k(x_i, z_n) ) / d_j        (the new row j of V = T K*)
//     q_n += sum_j v_jn^2          p_n += sum_j v_jn a_j
// i.e. ONE pass of N*M covariance evaluations with q fused row-dots instead of the N^2 M triangular product --
// what `model.add_data(x, y)` + the next `index(xgrid)` cost in the reference is a full refit and a full solve
// (pybo/bayesopt.py:269, pybo/solvers/lbfgs.py:50).  The q points of one `add_data(X, Y)` call share the pass:
// the covariance evaluations dominate (46 fp64 instructions each against one FMA per extra point).
// Wq (q, ldw): row j = [w_j (N_j entries), -1 at position N_j (the point's own row of Xs), zeros], so that
// v_jn = - (Wq_j . k(X_all, z_n)) / d_j with one dot over all Ntot = N_0 + q rows.  pscal (q, 2) = {1/d_j, a_j}.
// One workgroup owns 128 candidates and walks all observed rows in tiles of 64 (fixed order: results do not
// depend on the launch geometry); thread (ty, tx) accumulates rows ty*8..+7 of each tile for candidates
// tx*4..+3, the 8 row groups are combined through LDS at the end.
// ------------------------------------------------------------------------------------------------
template <int Q, int KID>
__global__ __launch_bounds__(256, (Q == 1) ? 3 : 2) void k_sweep_rankq(const double* __restrict__ Xs, int64_t Ntot, int d,
                                                     const double* __restrict__ Wq, int64_t ldw, int q,
                                                     const double* __restrict__ pscal,
                                                     const double* __restrict__ Z, int64_t M,
                                                     const double* __restrict__ invell, double rho,
                                                     double* __restrict__ qsum, double* __restrict__ psum,
                                                     const double* __restrict__ xlast, double* __restrict__ vout) {
    // xlast (optional): the scaled coordinates of row Ntot - 1 when that row is NOT in Xs yet -- an ANNOUNCED
    // observation (gpx_append_begin): its location is known, its value is not, the factor is untouched.
    // vout (optional, q == 1): write v_n instead of updating the sums; gpx_append applies q += v^2, p += v a once the
    // value has arrived (k_cache_apply: the same two FMAs, the same bits).
    __shared__ double xo[XDC][XK];
    __shared__ double xc[XDC][XN];
    __shared__ double wv[Q][XK];
    __shared__ double red[8][XN];
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int64_t n0 = (int64_t)blockIdx.x * XN;
    double acc[Q][4];
#pragma unroll
    for (int j = 0; j < Q; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[j][b] = 0.0;
    const bool onepass = (d <= XDC);          // candidates' coordinates stay in LDS across row tiles
    const int crow = t & (XN - 1), ckq = t >> 7;      // staging of candidates: row, first coordinate (stride 2)
    const int orow = t & (XK - 1), okq = t >> 6;      // staging of observed rows: row, first coordinate (stride 4)
    const int64_t gmc = n0 + crow;
    if (onepass)
        for (int k = ckq; k < d; k += 2) xc[k][crow] = (gmc < M) ? Z[gmc * d + k] * invell[k] : 0.0;
    for (int64_t k0 = 0; k0 < Ntot; k0 += XK) {
        double r2[8][4];
        auto stage = [&](int c0, int kc) {
            __syncthreads();
            const int64_t gr = k0 + orow;
            const double* src = (xlast && gr == Ntot - 1) ? xlast + c0 : Xs + gr * d + c0;
            for (int k = okq; k < kc; k += 4) xo[k][orow] = (gr < Ntot) ? src[k] : 0.0;
            if (c0 == 0) {
                for (int e = t; e < Q * XK; e += 256) {
                    const int j = e / XK, row = e - j * XK;      // XK is a compile-time power of two
                    wv[j][row] = (j < q && k0 + row < Ntot) ? Wq[(int64_t)j * ldw + k0 + row] : 0.0;
                }
            }
            if (!onepass)
                for (int k = ckq; k < kc; k += 2)
                    xc[k][crow] = (gmc < M) ? Z[gmc * d + c0 + k] * invell[c0 + k] : 0.0;
            __syncthreads();
        };
        {
            const int kc = min(XDC, d);
            stage(0, kc);
            dist_step<true>(xo, xc, 0, ty, tx, r2);
#pragma unroll 1
            for (int k = 1; k < kc; ++k) dist_step<false>(xo, xc, k, ty, tx, r2);
        }
        for (int c0 = XDC; c0 < d; c0 += XDC) {
            const int kc = min(XDC, d - c0);
            stage(c0, kc);
#pragma unroll 1
            for (int k = 0; k < kc; ++k) dist_step<false>(xo, xc, k, ty, tx, r2);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            double kv[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) kv[b] = kern_eval(KID, r2[a][b], rho);
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const double wa = wv[j][ty * 8 + a];      // 0 beyond a point's own row: contributes nothing
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[j][b] = fma(wa, kv[b], acc[j][b]);
            }
        }
    }
    double dq = 0.0, dp = 0.0;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; ++b) red[ty][tx * 4 + b] = acc[j][b];
        __syncthreads();
        if (t < XN && j < q) {
            double dot = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) dot += red[g][t];
            const double v = -dot * pscal[2 * j];
            if (vout) {
                if (n0 + t < M) vout[n0 + t] = v;
            } else {
                dq = fma(v, v, dq);
                dp = fma(v, pscal[2 * j + 1], dp);
            }
        }
    }
    if (t < XN && !vout) {
        const int64_t gm = n0 + t;
        if (gm < M) {
            qsum[gm] += dq;
            psum[gm] += dp;
        }
    }
}

// the value of an announced observation has arrived: q_n += v_n^2, p_n += v_n a  (a = scal[2], device-resident)
__global__ void k_cache_apply(const double* __restrict__ v, const double* __restrict__ scal, int64_t M,
                              double* __restrict__ qsum, double* __restrict__ psum) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double vi = v[i];
    qsum[i] += fma(vi, vi, 0.0);
    psum[i] += fma(vi, scal[2], 0.0);
}

void launch_cache_apply(hipStream_t s, const double* v, const double* scal, int64_t M, double* qsum, double* psum) {
    hipLaunchKernelGGL(k_cache_apply, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, v, scal, M, qsum, psum);
}

// x scaled by 1/ell (the row an announced observation will occupy in Xs)
__global__ void k_scale_point(const double* __restrict__ x, const double* __restrict__ invell, int d,
                              double* __restrict__ xs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < d) xs[k] = x[k] * invell[k];
}

void launch_scale_point(hipStream_t s, const double* x, const double* invell, int d, double* xs) {
    hipLaunchKernelGGL(k_scale_point, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, s, x, invell, d, xs);
}

template <int Q>
static void launch_rankq_kid(hipStream_t s, dim3 grid, const double* Xs, int64_t Ntot, int d, const double* Wq,
                             int64_t ldw, int q, const double* pscal, const double* Z, int64_t M, const double* invell,
                             int kernel_id, double rho, double* qsum, double* psum, const double* xlast, double* vout) {
#define GPX_RANKQ(KID)                                                                                              \
    hipLaunchKernelGGL((k_sweep_rankq<Q, KID>), grid, dim3(256), 0, s, Xs, Ntot, d, Wq, ldw, q, pscal, Z, M, invell, \
                       rho, qsum, psum, xlast, vout)
    switch (kernel_id) {
        case GPX_KERN_SE_ARD: GPX_RANKQ(GPX_KERN_SE_ARD); break;
        case GPX_KERN_MATERN52: GPX_RANKQ(GPX_KERN_MATERN52); break;
        case GPX_KERN_MATERN32: GPX_RANKQ(GPX_KERN_MATERN32); break;
        default: GPX_RANKQ(GPX_KERN_MATERN12); break;
    }
#undef GPX_RANKQ
}

// the correction pass of ONE announced observation: v_n for every cached candidate -> vout (sums untouched)
void launch_sweep_rank1_v(hipStream_t s, const double* Xs, int64_t Ntot, int d, const double* Wq, int64_t ldw,
                          const double* pscal, const double* Z, int64_t M, const double* invell, int kernel_id,
                          double rho, const double* xlast, double* vout) {
    const dim3 grid((unsigned)((M + XN - 1) / XN));
    launch_rankq_kid<1>(s, grid, Xs, Ntot, d, Wq, ldw, 1, pscal, Z, M, invell, kernel_id, rho, nullptr, nullptr, xlast,
                        vout);
}

// row j of the pending-correction table: [w (Nj entries), -1, zeros up to ldw]; pscal_j = {1/d, a_new}
__global__ void k_pend_store(const double* __restrict__ w, int64_t Nj, int64_t ldw, const double* __restrict__ scal,
                             double* __restrict__ row, double* __restrict__ pscal_j) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ldw) row[i] = (i < Nj) ? w[i] : ((i == Nj) ? -1.0 : 0.0);
    if (i == 0) {
        pscal_j[0] = scal[1];
        pscal_j[1] = scal[2];
    }
}

void launch_pend_store(hipStream_t s, const double* w, int64_t Nj, int64_t ldw, const double* scal, double* row,
                       double* pscal_j) {
    hipLaunchKernelGGL(k_pend_store, dim3((unsigned)((ldw + 255) / 256)), dim3(256), 0, s, w, Nj, ldw, scal, row,
                       pscal_j);
}

void launch_sweep_rankq(hipStream_t s, const double* Xs, int64_t Ntot, int d, const double* Wq, int64_t ldw, int q,
                        const double* pscal, const double* Z, int64_t M, const double* invell, int kernel_id,
                        double rho, double* qsum, double* psum) {
    const dim3 grid((unsigned)((M + XN - 1) / XN));
    if (q == 1)
        launch_rankq_kid<1>(s, grid, Xs, Ntot, d, Wq, ldw, q, pscal, Z, M, invell, kernel_id, rho, qsum, psum, nullptr,
                            nullptr);
    else if (q <= 4)
        launch_rankq_kid<4>(s, grid, Xs, Ntot, d, Wq, ldw, q, pscal, Z, M, invell, kernel_id, rho, qsum, psum, nullptr,
                            nullptr);
    else
        launch_rankq_kid<8>(s, grid, Xs, Ntot, d, Wq, ldw, q, pscal, Z, M, invell, kernel_id, rho, qsum, psum, nullptr,
                            nullptr);
}

// ------------------------------------------------------------------------------------------------
// top-k: value descending, ties -> lower index; NaN ranks below everything.
// ------------------------------------------------------------------------------------------------
constexpr int TK_PER_THREAD = 16;
constexpr int TK_PER_BLOCK = 256 * TK_PER_THREAD;
#define GPX_NEG_INF (-__builtin_huge_val())
#define GPX_IDX_NONE ((int64_t)0x7fffffffffffffffLL)

__device__ __forceinline__ bool better(double av, int64_t ai, double bv, int64_t bi) {
    return (av > bv) || (av == bv && ai < bi);
}

// block-wide argmax of (v, i); result broadcast to all threads
__device__ __forceinline__ void block_argmax(double& v, int64_t& i, double* sv, int64_t* si) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int64_t oi = __shfl_xor((long long)i, off);
        if (better(ov, oi, v, i)) { v = ov; i = oi; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { sv[w] = v; si[w] = i; }
    __syncthreads();
    v = sv[0]; i = si[0];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww)
        if (better(sv[ww], si[ww], v, i)) { v = sv[ww]; i = si[ww]; }
}

// cutv / cuti (optional, device): only candidates that rank strictly AFTER the pair (*cutv, *cuti) take part -- the
// k-th best of an earlier pass: requests beyond TOPK_PASS entries are served TOPK_PASS at a time (launch_topk).
__global__ __launch_bounds__(256) void k_topk_block(const double* __restrict__ vals, int64_t M, int k,
                                                    double* __restrict__ blkv,
                                                    int64_t* __restrict__ blki, const double* __restrict__ cutv,
                                                    const int64_t* __restrict__ cuti) {
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    // blockIdx.y: one of several value rows ranked by the same launch (the draws of a Thompson sweep): row q's values start
    // at vals + q M, its block lists at blkv / blki + q gridDim.x k
    vals += (int64_t)blockIdx.y * M;
    blkv += (int64_t)blockIdx.y * gridDim.x * k;
    blki += (int64_t)blockIdx.y * gridDim.x * k;
    const int64_t base = (int64_t)blockIdx.x * TK_PER_BLOCK;
    const bool cut = cutv != nullptr;
    const double cv = cut ? *cutv : 0.0;
    const int64_t ci = cut ? *cuti : 0;
    double v[TK_PER_THREAD];
#pragma unroll
    for (int e = 0; e < TK_PER_THREAD; ++e) {
        const int64_t idx = base + e * 256 + threadIdx.x;
        double x = (idx < M) ? vals[idx] : GPX_NEG_INF;
        if (x != x) x = GPX_NEG_INF;
        v[e] = x;
    }
    unsigned used = 0;  // bit e set: element e already emitted (or ranked at / before the cut)
    if (cut) {
#pragma unroll
        for (int e = 0; e < TK_PER_THREAD; ++e) {
            const int64_t idx = base + e * 256 + threadIdx.x;
            // (ci < 0: the earlier pass already ran out of candidates -- its last entry is the -1 marker: nothing is left)
            if (ci < 0 || !better(cv, ci, v[e], idx)) used |= 1u << e;
        }
    }
    for (int it = 0; it < k; ++it) {
        double bv = GPX_NEG_INF;
        int64_t bi = GPX_IDX_NONE;
#pragma unroll
        for (int e = 0; e < TK_PER_THREAD; ++e) {
            const int64_t idx = base + e * 256 + threadIdx.x;
            if (!((used >> e) & 1u) && idx < M && better(v[e], idx, bv, bi)) { bv = v[e]; bi = idx; }
        }
        block_argmax(bv, bi, sv, si);
        if (bi != GPX_IDX_NONE) {
            const int64_t off = bi - base;
            if ((int)(off & 255) == (int)threadIdx.x) used |= 1u << (unsigned)(off >> 8);
        }
        if (threadIdx.x == 0) {
            blkv[(int64_t)blockIdx.x * k + it] = bv;
            blki[(int64_t)blockIdx.x * k + it] = bi;
        }
    }
}

__global__ __launch_bounds__(256) void k_topk_merge(double* __restrict__ blkv, int64_t* __restrict__ blki,
                                                    int64_t n, int k, double* __restrict__ topv,
                                                    int64_t* __restrict__ topi) {
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    blkv += (int64_t)blockIdx.x * n;          // blockIdx.x: the value row (see k_topk_block)
    blki += (int64_t)blockIdx.x * n;
    topv += (int64_t)blockIdx.x * k;
    topi += (int64_t)blockIdx.x * k;
    for (int it = 0; it < k; ++it) {
        double bv = GPX_NEG_INF;
        int64_t bi = GPX_IDX_NONE;
        int64_t bpos = -1;
        for (int64_t e = threadIdx.x; e < n; e += 256) {
            const int64_t idx = blki[e];
            if (idx != GPX_IDX_NONE && better(blkv[e], idx, bv, bi)) { bv = blkv[e]; bi = idx; bpos = e; }
        }
        const int64_t mine = bi;
        block_argmax(bv, bi, sv, si);
        if (bi != GPX_IDX_NONE && mine == bi && bpos >= 0) blki[bpos] = GPX_IDX_NONE;  // consume
        if (threadIdx.x == 0) { topv[it] = bv; topi[it] = (bi == GPX_IDX_NONE) ? -1 : bi; }
        __syncthreads();
    }
}

void launch_topk_merge(hipStream_t s, double* vals, int64_t* idx, int64_t n, int k, double* topv, int64_t* topi) {
    hipLaunchKernelGGL(k_topk_merge, dim3(1), dim3(256), 0, s, vals, idx, n, k, topv, topi);
}

int64_t topk_blocks(int64_t M) { return (M + TK_PER_BLOCK - 1) / TK_PER_BLOCK; }

// The k <= TOPK_PASS best of EACH of S value rows (row q at vals + q M) in two launches instead of 2 S (a Thompson sweep of
// 64 draws ranked its rows one after the other: 128 launches of 5 us).  blkv / blki: S * nblk * k entries; topv / topi: (S, k).
void launch_topk_rows(hipStream_t s, const double* vals, int64_t M, int64_t S, int k, double* blkv, int64_t* blki,
                      int64_t nblk, double* topv, int64_t* topi) {
    hipLaunchKernelGGL(k_topk_block, dim3((unsigned)nblk, (unsigned)S), dim3(256), 0, s, vals, M, k, blkv, blki,
                       (const double*)nullptr, (const int64_t*)nullptr);
    hipLaunchKernelGGL(k_topk_merge, dim3((unsigned)S), dim3(256), 0, s, blkv, blki, nblk * k, k, topv, topi);
}

// k <= TOPK_MAX entries, TOPK_PASS per pass: pass p ranks only what comes strictly after the last entry of pass p-1
// (value descending, index ascending: a total order, so the passes concatenate to exactly the k best).  blkv / blki need
// nblk * min(k, TOPK_PASS) entries; topv / topi k.  The reference's ranking is a full argsort (pybo/solvers/lbfgs.py:51).
void launch_topk(hipStream_t s, const double* vals, int64_t M, int k, double* blkv, int64_t* blki,
                 int64_t nblk, double* topv, int64_t* topi) {
    for (int done = 0; done < k; done += TOPK_PASS) {
        const int kk = (k - done < TOPK_PASS) ? k - done : TOPK_PASS;
        const double* cv = done ? topv + done - 1 : nullptr;
        const int64_t* ci = done ? topi + done - 1 : nullptr;
        hipLaunchKernelGGL(k_topk_block, dim3((unsigned)nblk), dim3(256), 0, s, vals, M, kk, blkv, blki, cv, ci);
        hipLaunchKernelGGL(k_topk_merge, dim3(1), dim3(256), 0, s, blkv, blki, nblk * kk, kk, topv + done, topi + done);
    }
}

}  // namespace gpx
