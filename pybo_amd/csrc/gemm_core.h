// gemm_core.h -- fp64 MFMA tile engine for gfx950 (CDNA4), shared by the Cholesky trailing update,
// the triangular inversion and the posterior sweep.
//
// One workgroup (256 threads = 4 wave64, arranged 2x2) produces a 128x128 fp64 tile
//     D(m,n) = sum_{k in [k_lo,k_hi)} A(m,k) * B(k,n)
// with BOTH operands supplied "k-major":  A(m,k) = A[k*lda + m],  B(k,n) = B[k*ldb + n].
// That is the layout every caller is arranged to have (upper Cholesky factor R row-major, U = R^-1
// row-major, cross-Gram Ks[obs][candidate]), so every global load is a full 1 KiB row per wave and
// every LDS store is lane-linear; no transposes on the load path.
//
// Per wave: a 64x64 sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators (16 x 8 VGPR = 128 VGPR).
// Operand fragments (one f64 per lane: A[m = lane&15][k = lane>>4], B[k = lane>>4][n = lane&15])
// come from LDS with ds_read_b64; the row pitch of 144 f64 (= 288 dwords = 32 mod 64 banks) makes
// the two 32-lane groups of a ds_read_b64 hit disjoint bank halves -> conflict-free.
// Result layout (f64 MFMA, NOT the f32 map): D[row = (lane>>4) + 4*r][col = lane&15], r = 0..3.
//
// The k-loop schedules of this file perform the same arithmetic in the same order (bit-identical results):
//   operands by LDS-DMA (round 6; buffer_load_dwordx4 ... lds, no staging registers):
//     gemm_tile_128_l   k-step 32 (or 16) through ONE LDS image: the sweep (two workgroups per CU), the triangular inverse, the
//                       factorisation's workers at two workgroups per CU; optional skip of a triangular last k-block (TRI)
//     gemm_tile_128_ld  two images, the next step's loads behind the first four MFMA groups: a workgroup alone on its CU
//     gemm_tile_128_w   every WAVE keeps its own operand halves: no barrier at all (a witness of the sweep: it moves twice the bytes)
//   operands through registers (rounds 1-5; kept as independently scheduled witnesses and for the Gram / RFF products):
//     gemm_tile_128_s   k-step 32, buffer loads, banded priority, second fragment set (round 5's sweep)
//     gemm_tile_128_g   k-step 32, flat loads (round 2);  gemm_tile_128_b  k-step 16 through a 2-deep ring (round 1)
//     gemm_tile_128_d   two images, LDS writes behind the MFMA groups (round 5's lone workgroup)
// fp64 MFMA is 64 cycles/instruction/SIMD: a 32-row k-step is 8192 matrix cycles per wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace gpx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int TB = 128;           // tile edge (BM = BN)
constexpr int BK = 16;            // k per pipeline step
constexpr int LDT = TB + 16;      // LDS row pitch in f64
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_LDS_F64 = 2 * 2 * BK * LDT;        // [buf][A|B][BK][LDT]
constexpr int GEMM_LDS_BYTES = GEMM_LDS_F64 * 8;      // 73,728 B -> 2 workgroups per CU

__device__ __forceinline__ void acc_zero(d4 (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
}

// A, B already point at (k = 0, m = tile origin) / (k = 0, n = tile origin); k_lo, k_hi are multiples of BK;
// all 256 threads of the workgroup must call this.
//
// k-loop on the 2 x 16-row LDS ring ("write-at-top", prefetch distance 2): the global loads of tile t+2 are
// issued right after the registers holding tile t+1 have been written to LDS at the TOP of step t, so the
// end-of-step barrier never waits on a fresh ds_write (its lgkmcnt is a whole MFMA phase old) and the
// vmcnt wait at the top is for loads issued one full step earlier.  PRIO: raise the wave priority around
// the MFMA phase so that, of the two workgroups sharing a CU, the one in its matrix phase owns the pipe
// and the other one's staging/sync phase fills the gaps.
// (Round 1 also carried a write-at-end ring, a register-double-buffered fragment pipeline and an LDS-DMA
// staging variant; all tied or lost against the two schedules kept here -- DESIGN.md section 4 -- and were
// removed.  This one stays as the second, independently scheduled witness of the bit-identity test.)
// ILV (all loops): row block i of a wave is tile rows (2 i + wm) * 16 .. + 15 instead of (4 wm + i) * 16 .. (acc_row_ilv)
template <bool PRIO, bool ILV = false>
__device__ __forceinline__ void gemm_tile_128_b(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ B, int64_t ldb, int k_lo,
                                                int k_hi, double* smem) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    double* As = smem;
    double* Bs = smem + 2 * BK * LDT;
    const int lrow = w;
    const int lcol = lane * 2;
    d2 ra[4], rb[4];
    const int nk = (k_hi - k_lo) / BK;
    if (nk <= 0) return;
    const double* Ap = A + (int64_t)(k_lo + lrow) * lda + lcol;
    const double* Bp = B + (int64_t)(k_lo + lrow) * ldb + lcol;
    auto gload = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = *reinterpret_cast<const d2*>(Ap + (int64_t)(4 * p) * lda);
            rb[p] = *reinterpret_cast<const d2*>(Bp + (int64_t)(4 * p) * ldb);
        }
        Ap += (int64_t)BK * lda;
        Bp += (int64_t)BK * ldb;
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<d2*>(As + buf * BK * LDT + (lrow + 4 * p) * LDT + lcol) = ra[p];
            *reinterpret_cast<d2*>(Bs + buf * BK * LDT + (lrow + 4 * p) * LDT + lcol) = rb[p];
        }
    };
    gload();            // tile 0
    swrite(0);
    if (nk > 1) gload();  // tile 1 stays in registers until the top of step 0
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            swrite(buf ^ 1);                 // tile kt+1 (loaded during step kt-1)
            if (kt + 2 < nk) gload();        // tile kt+2, consumed at the top of step kt+1
        }
        const double* as = As + buf * BK * LDT + (ILV ? wm * 16 : wm * 64) + fr;
        const double* bs = Bs + buf * BK * LDT + wn * 64 + fr;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int kr = kk * 4 + fk;
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = as[kr * LDT + i * (ILV ? 32 : 16)];
                b[i] = bs[kr * LDT + i * 16];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
}


// DEFAULT k-loop: k-step of 32 through a SINGLE LDS buffer (same 73,728 B, so still two workgroups per CU):
// 128 MFMAs per wave between staging phases instead of 64, i.e. half as many step boundaries per flop.  The
// staging phase (barrier - LDS write - barrier) is exposed inside a workgroup and relies on the co-resident
// workgroup to keep the matrix pipe busy meanwhile.
constexpr int BK32 = 32;

// PRIO: wave priority during the matrix phase (0 = leave it alone; 1 for throughput tiles, 2 for the task-graph
// kernel's urgent tiles; PRIO - 1 outside the phase).
// NEGA: the A operand is negated on its way into LDS, i.e. acc += -(A^T-panel) * B: the symmetric updates start
// their accumulators from the S tile they update (loaded while the first k-steps are in flight) and store
// S - sum A B directly, instead of a dependent read-modify-write round trip after the k-loop.
// REV (the loops that take it): the 32-row k-steps are accumulated in DESCENDING order, the rows inside a step ascending -- a
// different, fixed summation order; every loop produces the same bits for the same REV (gemm_rev32 gives it to the 16-row loops).
template <int PRIO, bool NEGA = false, bool ILV = false, bool REV = false>
__device__ __forceinline__ void gemm_tile_128_g(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ B, int64_t ldb, int k_lo,
                                                int k_hi, double* smem) {
    // k_hi - k_lo must be a multiple of 32 (every caller's extent is a multiple of 128)
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    double* As = smem;                 // [BK32][LDT]
    double* Bs = smem + BK32 * LDT;    // [BK32][LDT]
    const int lrow = w;
    const int lcol = lane * 2;
    d2 ra[8], rb[8];
    const int nk = (k_hi - k_lo) / BK32;
    if (nk <= 0) return;
    const double* Ap = A + (int64_t)((REV ? k_hi - BK32 : k_lo) + lrow) * lda + lcol;
    const double* Bp = B + (int64_t)((REV ? k_hi - BK32 : k_lo) + lrow) * ldb + lcol;
    auto gload = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            ra[p] = *reinterpret_cast<const d2*>(Ap + (int64_t)(4 * p) * lda);
            rb[p] = *reinterpret_cast<const d2*>(Bp + (int64_t)(4 * p) * ldb);
        }
        Ap += (REV ? -1 : 1) * (int64_t)BK32 * lda;
        Bp += (REV ? -1 : 1) * (int64_t)BK32 * ldb;
    };
    auto swrite = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *reinterpret_cast<d2*>(As + (lrow + 4 * p) * LDT + lcol) = NEGA ? -ra[p] : ra[p];
            *reinterpret_cast<d2*>(Bs + (lrow + 4 * p) * LDT + lcol) = rb[p];
        }
    };
    const int fr = lane & 15, fk = lane >> 4;
    gload();
    swrite();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload();
        const double* as = As + (ILV ? wm * 16 : wm * 64) + fr;
        const double* bs = Bs + wn * 64 + fr;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
        for (int kk = 0; kk < BK32 / 4; ++kk) {
            const int kr = kk * 4 + fk;
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = as[kr * LDT + i * (ILV ? 32 : 16)];
                b[i] = bs[kr * LDT + i * 16];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO - 1);
        __syncthreads();               // everyone has finished reading the buffer
        if (kt + 1 < nk) {
            swrite();
            __syncthreads();
        }
    }
}

// The SWEEP's k-loop (round 5; the same arithmetic in the same order as gemm_tile_128_g: bit-identical results).
// What the phase stamps of scripts/sweep_phase showed about two workgroups sharing a compute unit (one wave of each per SIMD):
//  * the MFMA pipe serves ONE wave's matrix phase at a time; the other wave's phase starts when that one ends.  A wave
//    that still has VALU instructions ahead of its first MFMA (the 64-bit address arithmetic of 16 global loads) gets
//    them issued only in the gaps of the streaming wave -- it is NOT ready at the hand-over: ~280 idle clocks, twice
//    per 17.4 k-clock period.  Here the loads are BUFFER loads: the step's base lives in SGPRs and is bumped by SALU,
//    the row of load p is an SGPR offset, the thread's position one constant VGPR offset per operand -- no VALU.
//  * with both waves ready early the pipe is shared fairly, the two workgroups fall into lock-step and then stage at the
//    same time.  The wave further into its matrix phase therefore outranks the other one (s_setprio 1 for the first four
//    MFMA groups, 2 for the last four): whoever leads finishes first and stages while the other streams -- anti-phase is
//    the attractor, and the waiting wave sits at its first MFMA with its fragments loaded.
//  * the fragments of MFMA group kk+1 are requested before the MFMAs of group kk (a second fragment set; the buffer loads
//    freed the registers).
// 61.8 -> 59.7 ms per 65536-column launch at N = 8192 (0.905 -> 0.937 of the fp64-MFMA peak), profiles/history/r05_sweep_idle_attribution.txt.
// PRIO / NEGA as in gemm_tile_128_g (PRIO = 0: no priority changes at all; otherwise PRIO for the first half of a step's
// MFMAs, PRIO + 1 for the second half and while the loads are issued, PRIO - 1 outside the matrix phase).
template <int PRIO = 1, bool NEGA = false, bool ILV = false, bool REV = false>
__device__ __forceinline__ void gemm_tile_128_s(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ B, int64_t ldb, int k_lo, int k_hi,
                                                double* smem) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    double* As = smem;                 // [BK32][LDT]
    double* Bs = smem + BK32 * LDT;    // [BK32][LDT]
    const int lrow = w;
    const int lcol = lane * 2;
    d2 ra[8], rb[8];
    const int nk = (k_hi - k_lo) / BK32;
    if (nk <= 0) return;
    const char* Abase = reinterpret_cast<const char*>(A + (int64_t)(REV ? k_hi - BK32 : k_lo) * lda);
    const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)(REV ? k_hi - BK32 : k_lo) * ldb);
    const int voA = (int)(((int64_t)lrow * lda + lcol) * 8), voB = (int)(((int64_t)lrow * ldb + lcol) * 8);
    const int soA = (int)(4 * lda * 8), soB = (int)(4 * ldb * 8);      // four rows on: the SGPR offset of load p is p * so
    auto gload = [&]() {
        // (no bounds: num_records = 2^32 - 1; the base moves with the step, so the offsets stay below 32 rows)
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            ra[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rA, voA, p * soA, 0));
            rb[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rB, voB, p * soB, 0));
        }
        Abase += (REV ? -1 : 1) * (int64_t)BK32 * lda * 8;
        Bbase += (REV ? -1 : 1) * (int64_t)BK32 * ldb * 8;
    };
    auto swrite = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *reinterpret_cast<d2*>(As + (lrow + 4 * p) * LDT + lcol) = NEGA ? -ra[p] : ra[p];
            *reinterpret_cast<d2*>(Bs + (lrow + 4 * p) * LDT + lcol) = rb[p];
        }
    };
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int IST = ILV ? 32 : 16;
    const double* as = As + (ILV ? wm * 16 : wm * 64) + fr;
    const double* bs = Bs + wn * 64 + fr;
    gload();
    swrite();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO + 1);              // the load issue itself ahead of the partner's stream
        if (kt + 1 < nk) gload();
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        double a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[0][i] = as[fk * LDT + i * IST];
            b[0][i] = bs[fk * LDT + i * 16];
        }
#pragma unroll
        for (int kk = 0; kk < BK32 / 4; ++kk) {
            if (PRIO && kk == 4) __builtin_amdgcn_s_setprio(PRIO + 1);
            if (kk + 1 < BK32 / 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[(kk + 1) & 1][i] = as[((kk + 1) * 4 + fk) * LDT + i * IST];
                    b[(kk + 1) & 1][i] = bs[((kk + 1) * 4 + fk) * LDT + i * 16];
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // the LDS reads of group kk+1 first,
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);     // then the 16 MFMAs of group kk
        }
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO - 1);
        __syncthreads();               // everyone has finished reading the buffer
        if (kt + 1 < nk) {
            swrite();
            __syncthreads();
        }
    }
}

// The k-loop for a workgroup that has its compute unit to ITSELF (round 5: the task-graph factorisation's workers up to 72
// blocks -- one workgroup per CU, LDS 2 x 72 KB).  A lone workgroup on the single-buffer loops above runs at 0.68 of the matrix
// peak (scripts/sweep_phase: mode 256): barrier - LDS write - barrier stands exposed in every k-step and nobody fills it.
// Here the k-step of 32 goes through TWO LDS buffers: while the MFMAs of step t read buffer t & 1, the registers holding tile
// t + 1 are written to the other buffer two 16-byte pieces per MFMA group, and each register is refilled at once with its piece
// of tile t + 2 -- every global load has a whole step to arrive, every LDS write hides behind 16 MFMAs, ONE barrier per step.
// Same arithmetic in the same order as the other loops: bit-identical results.  smem: 2 x GEMM_LDS_F64 doubles.
// AUX: cache policy of the operand loads (16 = sc1: past this CU's L1 -- operands another workgroup of the SAME launch has just
// stored write-through need no acquire fence then; every element is loaded once per workgroup, the L1 had nothing to give)
template <int PRIO = 1, bool NEGA = false, int AUX = 0>
__device__ __forceinline__ void gemm_tile_128_d(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ B, int64_t ldb, int k_lo, int k_hi,
                                                double* smem) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int lrow = w;
    const int lcol = lane * 2;
    d2 ra[8], rb[8];
    const int nk = (k_hi - k_lo) / BK32;
    if (nk <= 0) return;
    const char* Abase = reinterpret_cast<const char*>(A + (int64_t)k_lo * lda);
    const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)k_lo * ldb);
    const int voA = (int)(((int64_t)lrow * lda + lcol) * 8), voB = (int)(((int64_t)lrow * ldb + lcol) * 8);
    const int soA = (int)(4 * lda * 8), soB = (int)(4 * ldb * 8);
    const int fr = lane & 15, fk = lane >> 4;
    // tile 0 -> buffer 0, tile 1 -> registers
    {
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            ra[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rA, voA, p * soA, AUX));
            rb[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rB, voB, p * soB, AUX));
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *reinterpret_cast<d2*>(smem + (lrow + 4 * p) * LDT + lcol) = NEGA ? -ra[p] : ra[p];
            *reinterpret_cast<d2*>(smem + BK32 * LDT + (lrow + 4 * p) * LDT + lcol) = rb[p];
        }
        Abase += (int64_t)BK32 * lda * 8;
        Bbase += (int64_t)BK32 * ldb * 8;
        if (nk > 1) {
            __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
            __amdgpu_buffer_rsrc_t rB1 = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                ra[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rA1, voA, p * soA, AUX));
                rb[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rB1, voB, p * soB, AUX));
            }
            Abase += (int64_t)BK32 * lda * 8;
            Bbase += (int64_t)BK32 * ldb * 8;
        }
    }
    __syncthreads();
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    // one k-step; W1: tile kt + 1 exists (its registers go to the other buffer), W2: tile kt + 2 exists (registers refilled).
    // Compile-time flags: with run-time tests around the writes and loads every MFMA group became a basic block of its own
    // and nothing overlapped (0.65 instead of 0.73 of peak alone on a CU).
    auto step = [&](int kt, auto w1_, auto w2_) {
        constexpr bool W1 = decltype(w1_)::value, W2 = decltype(w2_)::value;
        double* cur = smem + (kt & 1) * GEMM_LDS_F64;
        double* nxt = smem + ((kt + 1) & 1) * GEMM_LDS_F64;
        const double* as = cur + wm * 64 + fr;
        const double* bs = cur + BK32 * LDT + wn * 64 + fr;
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
        double a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[0][i] = as[fk * LDT + i * 16];
            b[0][i] = bs[fk * LDT + i * 16];
        }
#pragma unroll
        for (int kk = 0; kk < BK32 / 4; ++kk) {
            if (kk + 1 < BK32 / 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[(kk + 1) & 1][i] = as[((kk + 1) * 4 + fk) * LDT + i * 16];
                    b[(kk + 1) & 1][i] = bs[((kk + 1) * 4 + fk) * LDT + i * 16];
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // the LDS reads of group kk + 1,
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);     // the 16 MFMAs of group kk,
            if (W1) {                                               // piece kk of tile kt + 1 into the other buffer ...
                *reinterpret_cast<d2*>(nxt + (lrow + 4 * kk) * LDT + lcol) = NEGA ? -ra[kk] : ra[kk];
                *reinterpret_cast<d2*>(nxt + BK32 * LDT + (lrow + 4 * kk) * LDT + lcol) = rb[kk];
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            if (W2) {                                               // ... and its registers refilled with tile kt + 2
                ra[kk] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rA, voA, kk * soA, AUX));
                rb[kk] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rB, voB, kk * soB, AUX));
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
        }
        if (W2) {
            Abase += (int64_t)BK32 * lda * 8;
            Bbase += (int64_t)BK32 * ldb * 8;
        }
        __syncthreads();               // buffer kt & 1 has been read by everyone, the other one is complete
    };
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(kt, std::true_type{}, std::true_type{});
    if (kt + 1 < nk) { step(kt, std::true_type{}, std::false_type{}); ++kt; }
    step(kt, std::false_type{}, std::false_type{});
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO - 1);
}

// The sweep's k-loop for THREE workgroups per compute unit (round 6).  What the round-5 stamps left standing: the matrix
// pipe of a SIMD serves one wave's matrix phase at a time, and a lone streaming wave leaves ~3 % of the pipe's issue slots
// empty (its own LDS reads, waits and priority changes); the partner cannot fill them because it is staging or parked
// behind a barrier.  A third resident wave per SIMD can -- but 3 x 73.7 KB of LDS and 3 x 248 VGPRs do not exist.  Here
// the operands travel global memory -> LDS WITHOUT registers (buffer_load_dwordx4 ... lds: every wave instruction lands
// one 1 KiB k-row, the row pitch of 144 doubles is set per instruction through M0), the k-step is BKL rows through ONE LDS
// buffer (BKL = 16: 36,864 B -> three workgroups per CU; BKL = 32: 73,728 B -> two), and with no staging registers the
// wave fits in 168 VGPRs (3 x 168 = 504 of a SIMD's 512).  The step: wait for the own loads, barrier (everyone's rows are
// in LDS), BKL / 4 groups of 16 MFMAs with the next group's fragments requested ahead, barrier (everyone has read), issue
// the next step's loads.  The load latency stands exposed inside a workgroup and is covered by the other workgroups of
// the CU -- the same bet as the single-buffer loop above, with no ds_write phase left to cover.
// Same arithmetic in the same order as every other loop here: bit-identical results.
typedef __attribute__((address_space(3))) void* gemm_lds_ptr;
template <int BKL>
constexpr int gemm_l_lds_f64() { return 2 * BKL * LDT; }

// TRI (needs ILV): A's last 128 k-rows [k_hi - 128, k_hi) are a lower-triangular block (A(m, k) = 0 for k > m, m and k
//      counted inside the block).  In its j-th 32-row quarter the row blocks 2 i + wm < 2 j hold only zeros for BOTH waves
//      when i < j: their MFMAs are skipped (an exact no-op: the skipped products are +0).  The interleaved rows are what
//      makes the skip worth it: both wave rows lose the same share (1.5 of the block's 4 quarters).
// NEGA: acc += -(A) B through the MFMA's own negation of its A operand (neg:[1,0,0]): the same bits as negating A on its way
//      into LDS (the register-staged loops), which a DMA cannot do.  AUX: cache policy of the operand loads (see gemm_tile_128_d).
// REV: the 32-row k-steps in DESCENDING order (see gemm_tile_128_g).
template <int BKL, int PRIO = 1, int NSET = 2, bool ILV = false, bool TRI = false, bool NEGA = false, int AUX = 0, bool REV = false>
__device__ __forceinline__ void gemm_tile_128_l(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ B, int64_t ldb, int k_lo, int k_hi,
                                                double* smem) {
    static_assert(!TRI || ILV, "TRI needs the interleaved row blocks");
    static_assert(!REV || BKL == 32, "REV is the order of the 32-row steps: gemm_rev32 for other step sizes");
    constexpr int G = BKL / 4;         // MFMA groups per step
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    double* As = smem;                 // [BKL][LDT]
    double* Bs = smem + BKL * LDT;     // [BKL][LDT]
    const int nk = (k_hi - k_lo) / BKL;
    if (nk <= 0) return;
    const char* Abase = reinterpret_cast<const char*>(A + (int64_t)(REV ? k_hi - BKL : k_lo) * lda);
    const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)(REV ? k_hi - BKL : k_lo) * ldb);
    const int64_t stepA = (REV ? -1 : 1) * (int64_t)BKL * lda * 8, stepB = (REV ? -1 : 1) * (int64_t)BKL * ldb * 8;
    const int voA = (int)(((int64_t)w * lda + 2 * lane) * 8), voB = (int)(((int64_t)w * ldb + 2 * lane) * 8);
    const int soA = (int)(4 * lda * 8), soB = (int)(4 * ldb * 8);      // four rows on: the SGPR offset of load p is p * so
    auto issue = [&]() {
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
#pragma unroll
        for (int p = 0; p < BKL / 4; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (gemm_lds_ptr)(As + (w + 4 * p) * LDT), 16, voA, p * soA, 0, AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (gemm_lds_ptr)(Bs + (w + 4 * p) * LDT), 16, voB, p * soB, 0, AUX);
        }
        Abase += stepA;
        Bbase += stepB;
    };
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int IST = ILV ? 32 : 16;                                  // row-block stride of a wave's A fragments
    const double* as = As + (ILV ? wm * 16 : wm * 64) + fr + fk * LDT;
    const double* bs = Bs + wn * 64 + fr + fk * LDT;
    // one k-step; I0: row blocks i < I0 are skipped (TRI); more: there is a next step whose loads go out at the end
    auto step = [&](auto i0c, bool more) {
        constexpr int I0 = decltype(i0c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's rows of this step are in LDS
        asm volatile("" ::: "memory");
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        double a[NSET][4], b[NSET][4];
        auto frag = [&](int set, int g) {
#pragma unroll
            for (int i = I0; i < 4; ++i) a[set][i] = as[g * 4 * LDT + i * IST];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[set][j] = bs[g * 4 * LDT + j * 16];
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            if (PRIO && kk == G / 2) __builtin_amdgcn_s_setprio(PRIO + 1);
            if (NSET == 2 && kk + 1 < G) {
                frag((kk + 1) & 1, kk + 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 8 - I0, 0);      // the LDS reads of group kk + 1 first,
            }
#pragma unroll
            for (int i = I0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk % NSET][i], b[kk % NSET][j], acc[i][j], 0, 0, NEGA ? 1 : 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * (4 - I0), 0);    // then the MFMAs of group kk
            if (NSET == 1 && kk + 1 < G) frag(0, kk + 1);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        // (lgkmcnt: the compiler may sink a step's last MFMAs below the barrier and leave their fragment reads in flight above
        //  it; no LDS read of this wave may still be pending when another wave's DMA starts to overwrite the buffer)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // everyone has finished reading the buffer
        if (more) issue();
    };
    issue();
    if (!TRI) {
        for (int kt = 0; kt < nk; ++kt) step(std::integral_constant<int, 0>{}, kt + 1 < nk);
    } else {
        // the triangular block's quarters: its first (lowest k) is full; BKL = 16 takes each quarter in two steps
        constexpr int Q = 32 / BKL;                     // steps per quarter
        const int nd = nk - 3 * Q;                      // steps that multiply all four row blocks
        if (!REV) {
            for (int kt = 0; kt < nd; ++kt) step(std::integral_constant<int, 0>{}, true);
            for (int q = 0; q < Q; ++q) step(std::integral_constant<int, 1>{}, true);
            for (int q = 0; q < Q; ++q) step(std::integral_constant<int, 2>{}, true);
            for (int q = 0; q < Q; ++q) step(std::integral_constant<int, 3>{}, q + 1 < Q);
        } else {
            for (int q = 0; q < Q; ++q) step(std::integral_constant<int, 3>{}, true);
            for (int q = 0; q < Q; ++q) step(std::integral_constant<int, 2>{}, true);
            for (int q = 0; q < Q; ++q) step(std::integral_constant<int, 1>{}, true);
            for (int kt = 0; kt < nd; ++kt) step(std::integral_constant<int, 0>{}, kt + 1 < nd);
        }
    }
}

// The register-free loop for a workgroup that has its compute unit to ITSELF (two k-step images of LDS, 2 x 73,728 B): while
// step t is multiplied out of image t & 1, the DMA of step t + 1 fills the other image -- issued four instructions at a time
// behind the step's first four MFMA groups -- and ONE barrier ends the step: behind it every wave has finished reading image
// t & 1 AND every wave's rows of step t + 1 have landed (each wave waits for its own loads first).  Against gemm_tile_128_d
// (registers, LDS writes behind the MFMA groups): no staging registers, no ds_write.  Same arithmetic, same order, same bits.
template <int PRIO = 1, bool NEGA = false, int AUX = 0>
__device__ __forceinline__ void gemm_tile_128_ld(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                                 const double* __restrict__ B, int64_t ldb, int k_lo, int k_hi,
                                                 double* smem) {
    constexpr int G = BK32 / 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int nk = (k_hi - k_lo) / BK32;
    if (nk <= 0) return;
    const char* Abase = reinterpret_cast<const char*>(A + (int64_t)k_lo * lda);
    const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)k_lo * ldb);
    const int voA = (int)(((int64_t)w * lda + 2 * lane) * 8), voB = (int)(((int64_t)w * ldb + 2 * lane) * 8);
    const int soA = (int)(4 * lda * 8), soB = (int)(4 * ldb * 8);
    // rows w + 4 p, p = P0 .. P1 - 1, of the step at Abase / Bbase into image img
    auto issue = [&](int img, auto p0c, auto p1c) {
        constexpr int P0 = decltype(p0c)::value, P1 = decltype(p1c)::value;
        double* As = smem + img * GEMM_LDS_F64;
        double* Bs = As + BK32 * LDT;
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
#pragma unroll
        for (int p = P0; p < P1; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (gemm_lds_ptr)(As + (w + 4 * p) * LDT), 16, voA, p * soA, 0, AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (gemm_lds_ptr)(Bs + (w + 4 * p) * LDT), 16, voB, p * soB, 0, AUX);
        }
    };
    auto bump = [&]() {
        Abase += (int64_t)BK32 * lda * 8;
        Bbase += (int64_t)BK32 * ldb * 8;
    };
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 8>;
    const int fr = lane & 15, fk = lane >> 4;
    const int aoff = wm * 64 + fr + fk * LDT, boff = BK32 * LDT + wn * 64 + fr + fk * LDT;
    issue(0, I0{}, I8{});
    bump();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    // One step.  FILL: the loads of the NEXT step go out behind this step's first four MFMA groups, four at a time, into the
    // image the previous step was read from (the barrier that ended it freed it); they have the step's other four groups
    // (4096 matrix clocks) to land.
    auto step = [&](int kt, auto fillc) {
        constexpr bool FILL = decltype(fillc)::value;
        const double* as = smem + (kt & 1) * GEMM_LDS_F64 + aoff;
        const double* bs = smem + (kt & 1) * GEMM_LDS_F64 + boff;
        const int img = (kt + 1) & 1;
        asm volatile("" ::: "memory");
        double a[2][4], b[2][4];
        auto frag = [&](int set, int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[set][i] = as[g * 4 * LDT + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[set][j] = bs[g * 4 * LDT + j * 16];
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            if (kk + 1 < G) {
                frag((kk + 1) & 1, kk + 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, NEGA ? 1 : 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            if (FILL && kk < 4) {
                if (kk == 0) issue(img, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                if (kk == 1) issue(img, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
                if (kk == 2) issue(img, std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
                if (kk == 3) issue(img, std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
            }
        }
        asm volatile("" ::: "memory");
        if (FILL) {
            bump();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's rows of the next step; its LDS reads of this one
            __builtin_amdgcn_s_barrier();                               // image kt & 1 is read, the other image is complete
        }
    };
    int kt = 0;
    for (; kt + 1 < nk; ++kt) step(kt, std::true_type{});
    step(kt, std::false_type{});
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO - 1);
    __builtin_amdgcn_s_barrier();                                       // the caller may reuse LDS
}

// The barrier-free loop: every WAVE keeps its own operands.  Wave (wm, wn) needs the 64 columns wm of A and the 64 columns wn
// of B of every k-row and nothing else; here it fetches exactly those itself (LDS-DMA) into a private LDS region -- 16 k-rows x
// (64 + 64) doubles = 16 KB per image -- and therefore never meets another wave at a barrier: it waits for its OWN loads
// (vmcnt), multiplies, and refills the image it has just read.  The price: each half of A and of B is fetched by the two waves
// that use it (twice the L2 -> LDS bytes of the shared-image loops).
//   NIMG = 2 (a workgroup alone on its CU: 128 KB): step t + 1 lands in the other image while step t is multiplied; its loads go
//            out four at a time behind the four MFMA groups of step t.
//   NIMG = 1 (two workgroups per CU: 64 KB each): the load latency stands exposed in the wave and is covered by the partner
//            workgroup's wave on the same SIMD.
// LDS image of a k-group (4 rows x 64 columns, 2 KB, two DMA instructions of 1 KB): the two rows a half-wave reads together sit
// 128 B apart inside 256 B lines -- [row r, columns 16 p .. 16 p + 15] at 256 p + 128 (r & 1) + 1024 (r >> 1) -- so a fragment
// read (lanes (fr, fk) <- row fk, column 16 i + fr) touches 256 contiguous bytes per half-wave: conflict-free without padding.
// The DMA writes lane-linearly; the permutation is applied to the per-lane GLOBAL address.  Same arithmetic, order and bits.
constexpr int GEMM_W_IMG_F64 = 2 * 16 * 64;          // one wave's image: A half + B half, 16 k-rows x 64 columns each

template <int NIMG, int PRIO = 1, bool NEGA = false, bool ILV = false, int AUX = 0, bool TRI = false>
__device__ __forceinline__ void gemm_tile_128_w(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ B, int64_t ldb, int k_lo, int k_hi,
                                                double* smem) {
    constexpr int BKW = 16, G = 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int nk = (k_hi - k_lo) / BKW;
    if (nk <= 0) return;
    double* mine = smem + w * (NIMG * GEMM_W_IMG_F64);
    const char* Abase = reinterpret_cast<const char*>(A + (int64_t)k_lo * lda);
    const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)k_lo * ldb);
    // lane L supplies 16 bytes at LDS offset 16 L of a 1 KB block: piece p = L >> 4, row-in-pair r = (L >> 3) & 1, column pair L & 7
    const int lp = lane >> 4, lr = (lane >> 3) & 1, lc = lane & 7;
    const int colA = (ILV ? (2 * lp + wm) * 16 : wm * 64 + 16 * lp) + 2 * lc, colB = wn * 64 + 16 * lp + 2 * lc;
    const int voA = (int)(((int64_t)lr * lda + colA) * 8), voB = (int)(((int64_t)lr * ldb + colB) * 8);
    const int soA = (int)(2 * lda * 8), soB = (int)(2 * ldb * 8);      // a block = two k-rows: block q of the step at q * so
    // blocks Q0 .. Q1 - 1 (of 8 per operand per step) of the step at Abase / Bbase into image img
    auto issue = [&](int img, auto q0c, auto q1c) {
        constexpr int Q0 = decltype(q0c)::value, Q1 = decltype(q1c)::value;
        double* As = mine + img * GEMM_W_IMG_F64;
        double* Bs = As + 16 * 64;
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
#pragma unroll
        for (int q = Q0; q < Q1; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (gemm_lds_ptr)(As + q * 128), 16, voA, q * soA, 0, AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (gemm_lds_ptr)(Bs + q * 128), 16, voB, q * soB, 0, AUX);
        }
    };
    auto bump = [&]() {
        Abase += (int64_t)BKW * lda * 8;
        Bbase += (int64_t)BKW * ldb * 8;
    };
    using Q0_ = std::integral_constant<int, 0>;
    using Q8_ = std::integral_constant<int, 8>;
    const int fr = lane & 15, fk = lane >> 4;
    const int foff = (fk >> 1) * 128 + (fk & 1) * 16 + fr;             // + 256 g (group) + 32 i (16-column piece), in doubles
    issue(0, Q0_{}, Q8_{});
    bump();
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    // I0: row blocks i < I0 are skipped (TRI, as in gemm_tile_128_l)
    auto step = [&](int kt, auto fillc, auto i0c) {
        constexpr bool FILL = decltype(fillc)::value;                   // the next step exists: fetch it (NIMG = 2: meanwhile)
        constexpr int I0 = decltype(i0c)::value;
        const int cur = (NIMG == 2) ? (kt & 1) : 0, nxt = (NIMG == 2) ? ((kt + 1) & 1) : 0;
        const double* as = mine + cur * GEMM_W_IMG_F64 + foff;
        const double* bs = as + 16 * 64;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // this wave's own rows of step kt (and no LDS read pending)
        double a[2][4], b[2][4];
        auto frag = [&](int set, int g) {
#pragma unroll
            for (int i = I0; i < 4; ++i) a[set][i] = as[g * 256 + i * 32];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[set][j] = bs[g * 256 + j * 32];
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            if (kk + 1 < G) {
                frag((kk + 1) & 1, kk + 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 8 - I0, 0);
            }
#pragma unroll
            for (int i = I0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, NEGA ? 1 : 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * (4 - I0), 0);
            if (FILL && NIMG == 2) {
                if (kk == 0) issue(nxt, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                if (kk == 1) issue(nxt, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
                if (kk == 2) issue(nxt, std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
                if (kk == 3) issue(nxt, std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
            }
        }
        if (FILL) {
            if (NIMG == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the image is read: refill it
                issue(0, Q0_{}, Q8_{});
            }
            bump();
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int kt = 0;
    if (!TRI) {
        for (; kt + 1 < nk; ++kt) step(kt, T_{}, std::integral_constant<int, 0>{});
        step(kt, F_{}, std::integral_constant<int, 0>{});
    } else {
        // the triangular block = the last eight steps; its quarter j (two steps) skips the row blocks i < j
        for (; kt < nk - 6; ++kt) step(kt, T_{}, std::integral_constant<int, 0>{});
        step(kt, T_{}, std::integral_constant<int, 1>{}); ++kt;
        step(kt, T_{}, std::integral_constant<int, 1>{}); ++kt;
        step(kt, T_{}, std::integral_constant<int, 2>{}); ++kt;
        step(kt, T_{}, std::integral_constant<int, 2>{}); ++kt;
        step(kt, T_{}, std::integral_constant<int, 3>{}); ++kt;
        step(kt, F_{}, std::integral_constant<int, 3>{});
    }
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // the caller may reuse LDS
}

// The REV order for a loop that does not take it natively (the 16-row-step witnesses): one call per 32-row step, highest first.
// Slow (a prologue per step) and exact: the accumulators see the same sequence of MFMAs as in the native loops.
template <class Loop>
__device__ __forceinline__ void gemm_rev32(int k_lo, int k_hi, Loop loop) {
    for (int k = k_hi - BK32; k >= k_lo; k -= BK32) loop(k, k + BK32);
}

// tile row of accumulator register acc[i][.][r] under the interleaved row blocks (ILV)
__device__ __forceinline__ int acc_row_ilv(int i, int r) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return (2 * i + (w >> 1)) * 16 + (lane >> 4) + 4 * r;
}

// element coordinates of accumulator register acc[i][j][r] inside the 128x128 tile
__device__ __forceinline__ int acc_row(int i, int r) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return (w >> 1) * 64 + i * 16 + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int j) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return (w & 1) * 64 + j * 16 + (lane & 15);
}

// ------------------------------------------------------------------------------------------------
// Latency-oriented sibling: one workgroup (4 waves, 2x2) produces a 64x64 tile, each wave a 32x32
// sub-tile = 2x2 MFMA accumulators, i.e. 16 instead of 64 matrix instructions per wave per k-step.  Used by
// the kernels on the Cholesky's serial chain (panel solve, in-panel row update), which have too few
// 128-tiles to fill the chip: four times as many workgroups, each a quarter as long.
// Same operand convention (k-major A and B), same LDS bank trick (pitch 80 f64 = 160 dwords = 32 mod 64).
// ------------------------------------------------------------------------------------------------
constexpr int T64 = 64;
constexpr int LDT64 = T64 + 16;
constexpr int GEMM64_LDS_F64 = 2 * BK32 * LDT64;    // [A|B][32][LDT64] = 40,960 B

// 64x64 tile engine of the Cholesky's chain kernels (row updates): K is stepped 32 at a time through a SINGLE
// 40 KB LDS buffer with a TWO-deep register prefetch: the global loads
// of step t+2 are issued at the top of step t and written to LDS at the end of step t+1.  The chain kernels that
// use it are bound by the latency of their dependent global loads, not by MFMA or bandwidth (round 1's 16-row
// ring with a one-step prefetch: K = 512 = 32 round trips of ~3.3 us next to the trailing updates): half as many
// round trips, each covered by two compute phases instead of one.  k_hi - k_lo must be a multiple of 32.
// Workgroup = 512 threads = 8 waves (2 x 4), wave tile 32 x 16 = 2 MFMA accumulators: the kernels that use this tile
// are latency-bound per k-step, so the MFMA work of a step is spread over twice the waves of the 128-tile engine.
constexpr int GEMM64_THREADS = 512;

template <bool NEGA = false>
__device__ __forceinline__ void gemm_tile_64_g(d4 (&acc)[2], const double* __restrict__ A, int64_t lda,
                                               const double* __restrict__ B, int64_t ldb, int k_lo, int k_hi,
                                               double* smem) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wm = w >> 2, wn = w & 3;
    double* As = smem;                    // [32][LDT64]
    double* Bs = smem + BK32 * LDT64;     // [32][LDT64]
    const int lrow = t >> 5;              // 0..15: two k-rows per wave instruction
    const int lcol = (t & 31) * 2;
    const int nk = (k_hi - k_lo) / BK32;
    if (nk <= 0) return;
    const double* Ap = A + (int64_t)(k_lo + lrow) * lda + lcol;
    const double* Bp = B + (int64_t)(k_lo + lrow) * ldb + lcol;
    d2 ra[2][2], rb[2][2];
#define GPX_GLOAD64(ST)                                                                   \
    {                                                                                     \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                   \
            ra[ST][p] = *reinterpret_cast<const d2*>(Ap + (int64_t)(16 * p) * lda);       \
            rb[ST][p] = *reinterpret_cast<const d2*>(Bp + (int64_t)(16 * p) * ldb);       \
        }                                                                                 \
        Ap += (int64_t)BK32 * lda;                                                        \
        Bp += (int64_t)BK32 * ldb;                                                        \
    }
#define GPX_SWRITE64(ST)                                                                  \
    {                                                                                     \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                   \
            *reinterpret_cast<d2*>(As + (lrow + 16 * p) * LDT64 + lcol) = NEGA ? -ra[ST][p] : ra[ST][p]; \
            *reinterpret_cast<d2*>(Bs + (lrow + 16 * p) * LDT64 + lcol) = rb[ST][p];      \
        }                                                                                 \
    }
    GPX_GLOAD64(0);                       // tile 0
    GPX_SWRITE64(0);
    if (nk > 1) GPX_GLOAD64(1);           // tile 1 waits in stage 1
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    // two steps per trip so that the register stages are compile-time indices
    for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k = kt + half;
            if (k >= nk) break;
            // stage `half` is free (its tile is in LDS): fetch tile k + 2 into it
            if (k + 2 < nk) {
                if (half == 0) GPX_GLOAD64(0) else GPX_GLOAD64(1)
            }
            const double* as = As + wm * 32 + fr;
            const double* bs = Bs + wn * 16 + fr;
#pragma unroll
            for (int kk = 0; kk < BK32 / 4; ++kk) {
                const int kr = kk * 4 + fk;
                const double b = bs[kr * LDT64];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[kr * LDT64 + i * 16], b, acc[i], 0, 0, 0);
            }
            __syncthreads();               // everyone has finished reading the buffer
            if (k + 1 < nk) {
                if (half == 0) GPX_SWRITE64(1) else GPX_SWRITE64(0)     // tile k + 1 sits in the OTHER stage
                __syncthreads();
            }
        }
    }
#undef GPX_GLOAD64
#undef GPX_SWRITE64
}

__device__ __forceinline__ int acc_row64(int i, int r) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return (w >> 2) * 32 + i * 16 + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col64() {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return (w & 3) * 16 + (lane & 15);
}

}  // namespace gpx
