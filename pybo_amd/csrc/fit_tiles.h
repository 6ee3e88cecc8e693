// fit_tiles.h -- the tile bodies of the blocked Cholesky (shared by the stream-scheduled kernels in kernels_fit.hip
// and the persistent task-graph kernel in kernels_chol_tg.hip).
#pragma once
#include "gemm_core.h"
#include "gpx_internal.h"

namespace gpx {

// Memory policy of the tile bodies.  AG = false: the stream-scheduled kernels -- every hand-off is a kernel boundary,
// plain loads and stores.  AG = true: the persistent task-graph kernel -- tiles travel between workgroups of ONE launch,
// whose L1s are never refreshed by other CUs' stores and whose XCD L2s are write-back: every word another workgroup
// reads is stored write-through at agent scope (global_store ... sc1) and loaded past the L1 (global_load ... sc1); the
// publishing wave drains its stores (s_waitcnt vmcnt(0)) before ONE lane raises the flag
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").
template <bool AG>
__device__ __forceinline__ double ldg(const double* p) {
    if constexpr (AG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool AG>
__device__ __forceinline__ void stg(double* p, double v) {
    if constexpr (AG) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---- a 128 x 128 accumulator tile (gemm_core.h layout: acc[i][j][r] = row (w>>1)*64 + 16 i + (lane>>4) + 4 r, column
// (w&1)*64 + 16 j + (lane&15)) to and from global memory.  Plain policy: one 8-byte access per element.  Agent policy:
// 8-byte write-through stores are one 8-byte fabric write per LANE (a 128 KB tile = 16 384 partial-line writes: the first
// version of the task-graph kernel spent its time there); neighbouring lanes therefore swap one register of every pair
// (DPP quad_perm [1,0,3,2]) so that even lanes hold columns (n, n+1) of row g + 4 r0 and odd lanes columns (n-1, n) of
// row g + 4 r1: 16-byte sc1 accesses, eight lanes = one full 128-byte line per row.
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double dpp_swap1(double v) {        // the value of lane ^ 1
    union { double d; int i[2]; } u, o;
    u.d = v;
    o.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0xB1, 0xF, 0xF, true);
    o.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0xB1, 0xF, 0xF, true);
    return o.d;
}

// base = the tile's origin (wave-uniform), ld = row pitch in elements (ld * 8 * 128 < 2^31)
template <bool AG>
__device__ __forceinline__ void tile128_load(d4 (&acc)[4][4], const double* base, int64_t ld) {
    if constexpr (!AG) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = base[(int64_t)acc_row(i, r) * ld + acc_col(j)];
    } else {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, n = lane & 15;
        const bool odd = n & 1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, (int)(128 * ld * 8), 0x00020000);
        u4v raw[4][4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = (w >> 1) * 64 + 16 * i + g + 4 * (2 * h + (odd ? 1 : 0));
                    const int col = (w & 1) * 64 + 16 * j + (n & ~1);
                    raw[i][j][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((row * ld + col) * 8), 0, 16);
                }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    union { u4v v; double d[2]; } u;
                    u.v = raw[i][j][h];
                    // even lane: d[0] is its own register 2h, d[1] belongs to the odd neighbour's register 2h;
                    // odd lane:  d[1] is its own register 2h + 1, d[0] the even neighbour's register 2h + 1
                    const double got = dpp_swap1(odd ? u.d[0] : u.d[1]);
                    acc[i][j][2 * h] = odd ? got : u.d[0];
                    acc[i][j][2 * h + 1] = odd ? u.d[1] : got;
                }
    }
}

template <bool AG>
__device__ __forceinline__ void tile128_store(const d4 (&acc)[4][4], double* base, int64_t ld) {
    if constexpr (!AG) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) base[(int64_t)acc_row(i, r) * ld + acc_col(j)] = acc[i][j][r];
    } else {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, n = lane & 15;
        const bool odd = n & 1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(128 * ld * 8), 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double got = dpp_swap1(odd ? acc[i][j][2 * h] : acc[i][j][2 * h + 1]);
                    union { u4v v; double d[2]; } u;
                    u.d[0] = odd ? got : acc[i][j][2 * h];
                    u.d[1] = odd ? acc[i][j][2 * h + 1] : got;
                    const int row = (w >> 1) * 64 + 16 * i + g + 4 * (2 * h + (odd ? 1 : 0));
                    const int col = (w & 1) * 64 + 16 * j + (n & ~1);
                    __builtin_amdgcn_raw_buffer_store_b128(u.v, rs, (int)((row * ld + col) * 8), 0, 16);
                }
    }
}

// 1/sqrt(x) to fp64 round-off: hardware v_rsq_f64 seed (measured: 5.2e-8 relative) + Newton steps (4.1e-15 after
// one, 2.5e-16 after two -- scripts/potrf_bench.hip).  No denormal / scale handling needed: pivots of a PD matrix
// with unit-scale entries; a non-positive or NaN pivot is caught by the uniform pivot check before it is used.
#ifndef GPX_PF_NR
#define GPX_PF_NR 2        // Newton steps after v_rsq_f64 inside factor16 (scripts/potrf_bench.hip builds 1 and 2)
#endif
__device__ __forceinline__ double rsqrt_pf(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
#pragma unroll
    for (int i = 0; i < GPX_PF_NR; ++i) y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}

// ------------------------------------------------------------------------------------------------
// R2a' (round 2): diagonal-block factorisation, MFMA-blocked 16 wide.  One workgroup = 4 waves.
//
// The 128x128 block is 8x8 tiles of 16x16, each living in ONE wave's MFMA accumulators (f64 16x16x4 result
// layout: lane (g = lane>>4, n = lane&15), register r -> element [g + 4r][n]).  That layout IS the B-operand
// layout of the same instruction (step kk takes rows 4kk + g), and read as an A operand it is the transpose
// -- so tiles feed the next MFMA straight from registers.  Wave w owns tile COLUMNS w and 7-w (9 upper tiles).
// Step jb = 0..7 (right-looking; the order in time is 2, 3 + 1 of the NEXT tile -- see the lookahead note at
// pf16_factor / pf16_panel / pf16_trail):
//   1. the owner of tile (jb,jb) factors the augmented [D | I] IN THE WAVE, in place in its accumulators
//      (factor16 below: 4x4 pivot blocks, scalar 4x4 Cholesky, MFMA rank-4 updates) -- no LDS, no barrier
//      inside the 16 pivots.  Out: R_d (upper) to global memory, T_d = R_d^-T (as its transpose, k-major) into
//      a small LDS array.
//   2. barrier; every wave: R[jb,c] = T_d * S[jb,c] for its columns c > jb (4 MFMAs, B straight from the
//      accumulators), written into the LDS row panel and to global memory.
//   3. barrier; trailing update of its tiles (r,c), jb < r <= c: acc -= R[jb,r]^T R[jb,c], both operands k-major
//      reads of the panel (4 MFMAs per tile).
// 128 pivots cost 8 x ~1 us of in-wave chain instead of 64 barrier-separated pivot pairs on 256 threads that
// each carried 64 matrix elements and ~450 issue slots per pair (round 1: 75 us, instruction-issue-bound).
// The inverse of the whole block is NOT formed here: only the eight 16x16 inverses leave this kernel (into the
// diagonal tiles of T and U); k_trtri_diag128 below completes T_pp / U_pp.
// ------------------------------------------------------------------------------------------------
constexpr int PFP = 136;                  // row pitch (f64) of LDS images of 128-wide rows

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

// In-wave factorisation of one 16x16 tile held in the accumulator layout (D[r] = row g + 4r, column n), together
// with the same row operations on the identity (I -> T_d = R_d^-T): four steps over 4x4 pivot blocks.
//   * the 10 entries of the symmetric 4x4 pivot block are v_readlane'd into uniform values and every lane
//     factors it redundantly in scalars (4 dependent rsqrt chains: this is what is left of the serial chain);
//   * ALL cross-lane work is matrix instructions: pivot rows  [Rp | Tp] = T44 [D | I](rows of the block)  is one
//     16x16x4 MFMA each (A = T44 padded to 16 rows, B = the accumulator register that holds those rows), and the
//     rank-4 update of the remaining rows  [D | I] -= Rp^T [Rp | Tp]  is one MFMA each with A = the pivot rows as
//     they sit in the lanes (lane (g, n) holds Rp[g][n] = A[m = n][k = g]) and B = the same registers.
// 128 pivots then cost 32 x (20 readlanes + a 4x4 scalar Cholesky + 4 MFMAs) instead of 128 x (2 + 2 (15 - j))
// readlanes with a dependent FMA behind each (measured: 3.2 us per 16 pivots for that version).
// Returns -1, or the local index of the first non-positive pivot.
__device__ __forceinline__ int factor16(d4& D, d4& I, int g, int n) {
    int bad = -1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double ds = D[s];
        const double a00 = readlane_f64(ds, 4 * s), a01 = readlane_f64(ds, 4 * s + 1);
        const double a02 = readlane_f64(ds, 4 * s + 2), a03 = readlane_f64(ds, 4 * s + 3);
        const double a11 = readlane_f64(ds, 16 + 4 * s + 1), a12 = readlane_f64(ds, 16 + 4 * s + 2);
        const double a13 = readlane_f64(ds, 16 + 4 * s + 3);
        const double a22 = readlane_f64(ds, 32 + 4 * s + 2), a23 = readlane_f64(ds, 32 + 4 * s + 3);
        const double a33 = readlane_f64(ds, 48 + 4 * s + 3);
        // 4x4 Cholesky A44 = R44^T R44 (upper R44), i_k = 1 / R44[k][k]
        const double i0 = rsqrt_pf(a00);
        const double r01 = a01 * i0, r02 = a02 * i0, r03 = a03 * i0;
        const double p1 = fma(-r01, r01, a11);
        const double i1 = rsqrt_pf(p1);
        const double r12 = fma(-r01, r02, a12) * i1, r13 = fma(-r01, r03, a13) * i1;
        const double p2 = fma(-r12, r12, fma(-r02, r02, a22));
        const double i2 = rsqrt_pf(p2);
        const double r23 = fma(-r12, r13, fma(-r02, r03, a23)) * i2;
        const double p3 = fma(-r23, r23, fma(-r13, r13, fma(-r03, r03, a33)));
        const double i3 = rsqrt_pf(p3);
        // T44 = R44^-T (lower): forward substitution on the columns of the identity
        const double t10 = -(r01 * i0) * i1;
        const double t20 = -fma(r12, t10, r02 * i0) * i2, t21 = -(r12 * i1) * i2;
        const double t30 = -fma(r23, t20, fma(r13, t10, r03 * i0)) * i3;
        const double t31 = -fma(r23, t21, r13 * i1) * i3, t32 = -(r23 * i2) * i3;
        // A operand of the pivot-row products: A[m][k] = T44[m][k] for m < 4 (lane: m = n, k = g), else 0
        // (flat selects: written as an if / else-if chain over the lane's column the compiler emitted ten exec-mask branches here,
        //  in the middle of the block's longest dependent chain)
        const bool g0 = g == 0, g1 = g == 1, g2 = g == 2;
        const double q0 = g0 ? i0 : 0.0;
        const double q1 = g0 ? t10 : (g1 ? i1 : 0.0);
        const double q2 = g0 ? t20 : (g1 ? t21 : (g2 ? i2 : 0.0));
        const double q3 = g0 ? t30 : (g1 ? t31 : (g2 ? t32 : i3));
        const double q01 = (n == 0) ? q0 : q1, q23 = (n == 2) ? q2 : q3;
        const double ta = (n < 2) ? q01 : ((n < 4) ? q23 : 0.0);
        const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
        const d4 pr = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, ds, zero, 0, 0, 0);
        const d4 pt = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, I[s], zero, 0, 0, 0);
        // rows 0..3 of the products sit in register 0: lane (g, n) now holds row 4s + g of [R | T]
        const double rrow = (n >= 4 * s + g) ? pr[0] : 0.0;       // exact zeros left of the diagonal
        const double trow = pt[0];
        D[s] = rrow;
        I[s] = trow;
        if (s < 3) {
            const double aop = (n >= 4 * s + 4) ? -rrow : 0.0;    // rows m = n below the pivot block only
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, rrow, D, 0, 0, 0);
            I = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, trow, I, 0, 0, 0);
        }
        // the pivot checks AFTER the block's arithmetic is on its way: the pivots are wave-uniform, so each check is a scalar
        // compare-and-branch that waits for its pivot -- between the four dependent rsqrt chains they cost the chain their latency
        // (a bad pivot's garbage travels on harmlessly: the block is abandoned through `bad`)
        const double pv[4] = {a00, p1, p2, p3};
#pragma unroll
        for (int k = 0; k < 4; ++k) bad = (bad < 0 && (!(pv[k] > 0.0) || !(pv[k] < 1.0e300))) ? 4 * s + k : bad;
    }
    return bad;
}

// ---- the steps of k_potrf16, with LOOKAHEAD over the diagonal tiles --------------------------------------------
// Pn[2]: the 16-row panels of the current and the previous step ([16][PFP] each); Ud: the current T_d^T [16][16];
// Rg / Tg / Ug: the block in global memory (results leave from registers as they are produced).
//   factor   the owner of tile (jb, jb) factors it in the wave (factor16) and publishes T_d^T;
//   panel    every wave forms R[jb, c] = T_d S[jb, c] for its columns c > jb (LDS panel jb & 1 + global);
//   trail    trailing update with panel jb -- except that the wave that owns tile (jb+1, jb+1) brings only THAT
//            tile up to date, factors it at once (next to the other waves' trailing updates instead of after them:
//            the in-wave chain of factor16 is 2.6 of a step's 4.4 us and three waves idled through it), fixes its
//            other tile of row jb+1 (the next panel needs it) and DEFERS the rest of its trailing tiles: it applies
//            panel jb to them one step later, together with panel jb+1 (updates commute; the previous panel stays
//            in the other LDS buffer for exactly that long).
template <bool AG>
__device__ __forceinline__ bool pf16_factor(d4& D, int jb, int g, int n, double* __restrict__ Ud, volatile int* sflag,
                                            int64_t p0, int* flag, double* __restrict__ Rg, double* __restrict__ Tg,
                                            double* __restrict__ Ug, int64_t Np, int lane) {
    d4 I;
#pragma unroll
    for (int r = 0; r < 4; ++r) I[r] = (g + 4 * r == n) ? 1.0 : 0.0;
    const int bad = factor16(D, I, g, n);
    if (bad >= 0 && lane == 0) { *flag = (int)(p0 + 16 * jb + bad) + 1; *sflag = 1; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = p0 + 16 * jb + g + 4 * r, col = p0 + 16 * jb + n;
        stg<AG>(Rg + row * Np + col, D[r]);                           // R_d, zeros below its diagonal
        if (Tg) stg<AG>(Tg + row * Np + col, I[r]);                   // T_d (lower)
        stg<AG>(Ug + col * Np + row, I[r]);                           // U_d = T_d^T
        Ud[n * 16 + g + 4 * r] = I[r];                                // LDS, k-major: Ud[k][m] = T_d[m][k]
    }
    return bad < 0;
}

// acc(r, c) -= R[jb, r]^T R[jb, c] from the panel `P` (both operands k-major reads of it)
__device__ __forceinline__ void pf16_tile_update(d4& acc, const double* __restrict__ P, int r, int c, int g, int n) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(4 * kk + g) * PFP + 16 * r + n], -P[(4 * kk + g) * PFP + 16 * c + n],
                                                   acc, 0, 0, 0);
}

template <int JB, bool AG>
__device__ __forceinline__ void pf16_panel(d4 (&accA)[8], d4 (&accB)[8], int cA, int cB, double* __restrict__ Pn,
                                           const double* __restrict__ Ud, int lane, int64_t p0,
                                           double* __restrict__ Rg, int64_t Np) {
    const int g = lane >> 4, n = lane & 15;
    double* P = Pn + (JB & 1) * 16 * PFP;
    double a[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a[kk] = Ud[(4 * kk + g) * 16 + n];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int c = side ? cB : cA;
        if (c > JB) {
            const d4 src = side ? accB[JB] : accA[JB];
            d4 pnl = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pnl = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], src[kk], pnl, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                P[(g + 4 * r) * PFP + 16 * c + n] = pnl[r];
                stg<AG>(Rg + (p0 + 16 * JB + g + 4 * r) * Np + p0 + 16 * c + n, pnl[r]);
            }
        }
    }
}

// returns false if the looked-ahead factorisation hit a non-positive pivot (sflag is set for the other waves)
// STEP (the task-graph kernel's role C): the workgroup publishes a flag after every step, for a consumer that follows the
// factorisation 16 rows at a time.  The flag of step JB promises panel JB (rows 16 JB .. 16 JB + 15 of R right of the diagonal
// tile) and the inverses T_d(0 .. JB): every wave leaves this phase with those stores drained -- the wave that has just
// factored the NEXT tile keeps only that tile's own stores in flight (vector-memory stores complete in order; they are
// covered by the next flag, or by the block's final one).
template <int JB, bool AG, bool STEP = false>
__device__ __forceinline__ void pf16_trail(d4 (&accA)[8], d4 (&accB)[8], int cA, int cB, double* __restrict__ Pn,
                                           double* __restrict__ Ud, volatile int* sflag, int w, int lane, int64_t p0,
                                           int* flag, double* __restrict__ Rg, double* __restrict__ Tg,
                                           double* __restrict__ Ug, int64_t Np) {
    const int g = lane >> 4, n = lane & 15;
    constexpr int NX = JB + 1;                                    // the tile row the next step factors
    constexpr int OWN_NX = (NX < 4) ? NX : 7 - NX;                // wave that owns tile (NX, NX)
    constexpr int OWN_JB = (JB < 4) ? JB : 7 - JB;                // ... and the one that deferred at step JB - 1
    const double* Pc = Pn + (JB & 1) * 16 * PFP;                  // this step's panel
    const double* Pp = Pn + ((JB + 1) & 1) * 16 * PFP;            // the previous one (for deferred updates)
    const bool pend = (JB >= 1) && (w == OWN_JB);                 // this wave skipped panel JB-1 for rows >= JB+1
    if (w == OWN_NX) {
        // the next diagonal tile first, factored at once
        d4& D = (NX < 4) ? accA[NX] : accB[NX];
        if (pend) pf16_tile_update(D, Pp, NX, NX, g, n);
        pf16_tile_update(D, Pc, NX, NX, g, n);
        pf16_factor<AG>(D, NX, g, n, Ud, sflag, p0, flag, Rg, Tg, Ug, Np, lane);
        // its other tile of row NX feeds the next panel; everything below waits one step
        const int co = (NX < 4) ? cB : cA;
        if (co > NX) {
            d4& E = (NX < 4) ? accB[NX] : accA[NX];
            if (pend) pf16_tile_update(E, Pp, NX, co, g, n);
            pf16_tile_update(E, Pc, NX, co, g, n);
        }
    } else {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int c = side ? cB : cA;
            if (c > JB) {
#pragma unroll
                for (int r = JB + 1; r < 8; ++r) {
                    if (r <= c) {
                        d4 acc = side ? accB[r] : accA[r];
                        if (pend) pf16_tile_update(acc, Pp, r, c, g, n);
                        pf16_tile_update(acc, Pc, r, c, g, n);
                        if (side) accB[r] = acc; else accA[r] = acc;
                    }
                }
            }
        }
    }
    if constexpr (STEP) {
        if (w == OWN_NX) {
#ifdef GPX_PF16_COUNTED_WAIT      // round 5: leave this wave's stores of the NEXT tile in flight behind the flag (counts them by hand)
            if (Tg) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

// DBG (scripts/potrf_bench.hip only): wall-clock stamps of the phases go to `dbg` (thread 0)
// Batched use (gpx_loglik_batch): blockIdx.z = batch element, S / R / U `bs` elements apart, one flag each;
// T may be NULL (the batched path needs U_d only and keeps it in the dead diagonal blocks of S).
// The body of k_potrf16 (one workgroup of 256 threads; Pn / Ud / sflag are the caller's LDS).  sflag != 0 afterwards:
// a non-positive pivot (recorded in *flag) or an earlier block's failure.
// STEP: after step JB thread 0 stores step0 + JB + 1 into *cstep (agent scope; see pf16_trail)
template <bool DBG, bool AG = false, bool STEP = false>
__device__ __forceinline__ void potrf16_body(const double* __restrict__ S, double* __restrict__ R,
                                             double* __restrict__ T, double* __restrict__ U, int64_t Np, int p,
                                             int* __restrict__ flag, long long* __restrict__ dbg,
                                             double* __restrict__ Pn, double* __restrict__ Ud, int& sflag,
                                             int* cstep = nullptr, int step0 = 0) {
    if (*flag != 0) {                     // an earlier block already failed (uniform)
        if (threadIdx.x == 0) sflag = 1;
        __syncthreads();
        return;
    }
    // The chain kernels run at the highest wave priority: they share compute units with the side streams'
    // trailing updates, whose waves raise their own priority to 1 around their MFMA phases -- a chain wave at
    // priority 0 on the same SIMD was measured to run 3x slower (k_potrf16 36 -> 107 us).
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int64_t p0 = (int64_t)p * NB;
    const int cA = w, cB = 7 - w;
    if (t == 0) sflag = 0;
    d4 accA[8], accB[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        accA[r] = (d4){0.0, 0.0, 0.0, 0.0};
        accB[r] = (d4){0.0, 0.0, 0.0, 0.0};
        if (r <= cA) {
#pragma unroll
            for (int q = 0; q < 4; ++q) accA[r][q] = ldg<AG>(S + (p0 + 16 * r + g + 4 * q) * Np + p0 + 16 * cA + n);
        }
        if (r <= cB) {
#pragma unroll
            for (int q = 0; q < 4; ++q) accB[r][q] = ldg<AG>(S + (p0 + 16 * r + g + 4 * q) * Np + p0 + 16 * cB + n);
        }
    }
    // the block below the diagonal tiles is zero in R and in T's upper / U's lower part: written here, while the
    // loads above are in flight (tiles (r, c) with r > c for R; T and U only need their off-diagonal tiles
    // defined once k_trtri_diag128 has run, which writes them in full).  The task-graph kernel's launcher zero-fills
    // them ahead of the launch instead (k_zero_diag_lower): nothing but results on its critical path.
    if constexpr (!AG)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx4 = t + 256 * q;
        const int r = idx4 >> 5, c = (idx4 & 31) * 4;
        if ((c >> 4) < (r >> 4)) *reinterpret_cast<d4*>(R + (p0 + r) * Np + p0 + c) = (d4){0.0, 0.0, 0.0, 0.0};
    }
    if (DBG && t == 0) dbg[0] = wall_clock64();
    __syncthreads();
    if (DBG && t == 0) dbg[1] = wall_clock64();
    // tile (0, 0) is factored before the loop; every later diagonal tile inside the previous step's trailing phase
    if (w == 0) pf16_factor<AG>(accA[0], 0, g, n, Ud, &sflag, p0, flag, R, T, U, Np, lane);
    __syncthreads();
#define GPX_PF_STEP(JB)                                                                                     \
    if (!sflag) {                                                                                           \
        pf16_panel<JB, AG>(accA, accB, cA, cB, Pn, Ud, lane, p0, R, Np);                                        \
        __syncthreads();                                                                                    \
        pf16_trail<JB, AG, STEP>(accA, accB, cA, cB, Pn, Ud, &sflag, w, lane, p0, flag, R, T, U, Np);           \
        __syncthreads();                                                                                    \
        if (STEP && t == 0) __hip_atomic_store(cstep, step0 + JB + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    }                                                                                                       \
    if (DBG && t == 0) dbg[2 + JB] = wall_clock64();
    GPX_PF_STEP(0)
    GPX_PF_STEP(1)
    GPX_PF_STEP(2)
    GPX_PF_STEP(3)
    GPX_PF_STEP(4)
    GPX_PF_STEP(5)
    GPX_PF_STEP(6)
#undef GPX_PF_STEP
    if (DBG && t == 0) dbg[9] = wall_clock64();
    if (DBG && t == 0) dbg[10] = wall_clock64();
}


// ------------------------------------------------------------------------------------------------
// R2b' (round 2): panel solve  R[p, J] = R_pp^-T S[p, J]  by blocked forward substitution over 16-row tiles,
// from the factor R_pp and its eight 16x16 diagonal inverses alone -- the inverse of the 128-block is NOT on the
// Cholesky's critical path any more (round 1 formed it inside the diagonal kernel because this step was a GEMM
// with T_pp).  One workgroup = 64 columns; each WAVE owns 16 of them and runs the whole substitution on its own:
// its 8 right-hand-side tiles live in accumulators, the solved tile x_jb feeds
// the next MFMAs straight from registers as the B operand,
//     x_jb = T_d(jb) s_jb        s_i -= R[jb, i]^T x_jb   (i > jb)        144 MFMAs per wave.
// ------------------------------------------------------------------------------------------------
// cb: the 64-column group, counted from block column p + 1
template <bool AG = false>
__device__ __forceinline__ void panel_solve16_body(const double* __restrict__ U, const double* __restrict__ S,
                                                   double* __restrict__ R, int64_t Np, int p, int cb) {
    // NO LDS and no barrier: the A fragments (tiles of R_pp, the 16x16 inverses) are read straight from global
    // memory / L2 in the k-major fragment layout (lane (g, n) <- row 4kk + g, column n: four 128-byte segments
    // per instruction), one step ahead of the MFMAs that consume them.  A 155 KB LDS image of R_pp was measured
    // first: alone it ran in 9.5 us, but it needs an EMPTY compute unit, and next to the side stream's trailing
    // updates (2 x 72 KB per CU) its launches waited up to 480 us for one.
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int64_t p0 = (int64_t)p * NB;
    const int64_t j0 = (int64_t)(p + 1) * NB + (int64_t)cb * 64 + 16 * w;
    const double* Rd = R + p0 * Np + p0;          // R_pp
    const double* Ud = U + p0 * Np + p0;          // diagonal 16-tiles hold T_d^T
    d4 X[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) X[r][q] = ldg<AG>(S + (p0 + 16 * r + g + 4 * q) * Np + j0 + n);
    double ti[8][4];                               // A fragments of the eight T_d
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ti[jb][kk] = ldg<AG>(Ud + (int64_t)(16 * jb + 4 * kk + g) * Np + 16 * jb + n);
    double acur[7][4], anxt[7][4];                 // A fragments of R[jb, i], i = jb+1 .. 7
#pragma unroll
    for (int i = 1; i < 8; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acur[i - 1][kk] = ldg<AG>(Rd + (int64_t)(4 * kk + g) * Np + 16 * i + n);
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        if (jb + 1 < 8) {
#pragma unroll
            for (int i = jb + 2; i < 8; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    anxt[i - jb - 2][kk] = ldg<AG>(Rd + (int64_t)(16 * (jb + 1) + 4 * kk + g) * Np + 16 * i + n);
        }
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ti[jb][kk], X[jb][kk], x, 0, 0, 0);
        X[jb] = x;
        const d4 xn = -x;
#pragma unroll
        for (int i = jb + 1; i < 8; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                X[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[i - jb - 1][kk], xn[kk], X[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acur[i][kk] = anxt[i][kk];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) stg<AG>(R + (p0 + 16 * r + g + 4 * q) * Np + j0 + n, X[r][q]);
}


// DB: `smem` holds two k-step images and the workgroup has its compute unit to itself (gemm_tile_128_d)
template <bool AG = false, int PRIO = 1, bool DB = false>
__device__ __forceinline__ void syrk_tile(const double* __restrict__ R, double* __restrict__ S, int64_t Np, int kb0,
                                          int kb1, int I, int J, double* smem) {
    const int64_t i0 = (int64_t)I * NB, j0 = (int64_t)J * NB;
    // accumulators start from the S tile (its loads overlap the first k-steps), A enters negated: acc = S - A B
    d4 acc[4][4];
    double* tile = S + i0 * Np + j0;
    tile128_load<AG>(acc, tile, Np);
#ifdef GPX_TG_REGLOOPS       // A/B builds only: the register-staged k-loops of rounds 2-5
    if constexpr (DB) gemm_tile_128_d<PRIO, true, AG ? 16 : 0>(acc, R + i0, Np, R + j0, Np, kb0 * NB, kb1 * NB, smem);
    else gemm_tile_128_g<PRIO, true>(acc, R + i0, Np, R + j0, Np, kb0 * NB, kb1 * NB, smem);
#else
    // operands by LDS-DMA (round 6): no staging registers, no ds_write phase; A is negated by the MFMA itself -- same bits
    if constexpr (DB) gemm_tile_128_ld<PRIO, true, AG ? 16 : 0>(acc, R + i0, Np, R + j0, Np, kb0 * NB, kb1 * NB, smem);
    else gemm_tile_128_l<32, (PRIO > 2 ? 2 : PRIO), 2, false, false, true>(acc, R + i0, Np, R + j0, Np, kb0 * NB, kb1 * NB, smem);
#endif
    tile128_store<AG>(acc, tile, Np);
}

}  // namespace gpx
