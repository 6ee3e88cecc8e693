// kernels_chol_tg.hip -- the blocked Cholesky as ONE persistent kernel that walks the factorisation's task graph
// (option "chol_tg"; serves `model.add_data(X, Y)`, pybo/bayesopt.py:114,258,269 -- the fit behind every BO step).
//
// Why: the stream-scheduled factorisation in kernels_fit.hip launches ~250 kernels over four streams; its serial chain
// (diagonal block -> panel solve -> row update, per 128-block) is latency-bound, and every chain launch waits ~100 us
// to START behind the trailing updates' resident tiles (profiles/history/r03_chol_parts.txt).  Here nothing is launched after
// the first instruction: the workgroups stay resident, take roles, and hand tiles to each other through agent-scope
// flags in device memory.
//
//   role C (1 workgroup)    POTRF(p), p = 0 .. nP-1: the 16-wide MFMA-blocked diagonal factorisation (potrf16_body), with a
//                           counter published after each of its eight 16-row steps.
//   the shadows (9)         dedicated workgroups that FOLLOW role C (and each other) 16 rows at a time instead of waiting for
//                           whole tiles: the solves of tiles (p, p+1), (p, p+2), (p, p+3) [S1, S2, S3], the last two chunks of
//                           the next diagonal tile [U0, U], the final chunks of tiles (p+1, p+2), (p+1, p+3) [V, V2: two workgroups each] -- see "the
//                           shadows" below.  Between two diagonal blocks nothing starts from a flag and a cold tile any more.
//   role W (everyone else)  the throughput work, from ONE list: the links of the column chains right of that band -- the solve of
//                           tile (p, J) and, fused with it per 64-column half, the final chunk of tile (p+1, J) (tg_do_trsmu;
//                           two solve halves and a separate update where a workgroup has only one k-step image of LDS) -- and the
//                           tile updates  S_IJ -= sum_{k in [k0,k1)} R_kI^T R_kJ  on the 128 x 128 fp64-MFMA tile engine.
//
// Tile (I, J) receives its I block updates in CHUNKS of consecutive k, graded by distance from the pivot (default 1, 1, 2, 4,
// 8, 16, 16, ... blocks counted back from k = I): far from the pivot a chunk is long (arithmetic intensity), next to it the
// chunks are single blocks (latency).  Accumulators start from the S tile and k ascends within and across chunks, so every
// element sees the same sequence of FMAs as in the stream-scheduled kernels: the factor is BIT-IDENTICAL.
//
// Scheduling: the workers' list is built on the host in the order the tasks become available (a valid topological order),
// oldest first.  A workgroup without a task DRAWS the next ticket with one fetch-and-add and waits for THAT task's
// dependencies with the ticket in hand (bounded polling, 0.4 us between looks): nothing stands between "ready" and "running".
// Every ticket is held by a polling workgroup, tickets are drawn in topological order and a task only waits for earlier tasks
// or for the shadows (which wait only for role C, each other and earlier tasks), so the earliest incomplete task of the graph is
// always held by a workgroup that will run it: the list cannot dead-lock as long as role C, the shadows and one worker are
// resident, and roles are handed out in order of arrival.  Every spin is bounded (abort code 2 -> the launcher's caller
// re-runs the stream schedule).  Dependencies are counters: seq[I][J] = chunks applied to tile (I, J), solved[2J + h] =
// block rows solved in the 64-column half h of block column J, diag[p], quad[p] = the diagonal tile p is ready for role C,
// the step counters of role C / S1 / S2 / S3, and per half tile a flag "final" for the fused links.
// (History.  Round 4 had two solve halves and six update pieces per block row on a critical list served by eight side-kick
//  workgroups -- three hand-offs of 4-5 us between two diagonal blocks, block period 56 us; the A/B against it, before it was
//  removed: profiles/history/r05_chol_shadow_ab.txt.  Measured and removed earlier -- profiles/history/r04_chol_tg_*_ab.txt: claiming a head by
//  compare-and-swap, a peek before the draw, priority lists, strided sub-queues, an urgent-only pool, XCD-affine tickets,
//  column-major lists, the next solve's dependency cone on the side-kicks: every one slower or no faster.)
//
// Hand-offs follow MI355X_MICROARCH.md ("inter-workgroup visibility"): payloads are stored write-through at agent scope
// (fit_tiles.h, AG = true), every storing wave drains (s_waitcnt vmcnt(0)), barrier, ONE lane stores the flag; consumers
// poll relaxed from one wave, read tiles with sc1 loads (8-byte fragments) or after ONE agent acquire (the tile engine's
// 16-byte operand loads).
#include <algorithm>
#include <cstring>
#include <vector>

#include "fit_tiles.h"

namespace gpx {

enum { TG_TRSM = 1, TG_UPD = 2, TG_SHADOW = 4, TG_TRSMU = 5 };      // (3 was the piece of a diagonal tile's update: the shadows' U now)
struct TgTask { int16_t type, I, J, k0, k1, ord, aux, rsv; };     // 16 bytes; aux = column half (TRSM) / piece (UPDQ)

struct TgArgs {
    double *S, *R, *T, *U;
    int64_t Np;
    int nP;
    int* dflag;                  // [0] = failing pivot + 1
    int* ctl;                    // control block (zeroed before every launch), layout below
    const TgTask* q[2];          // 0: critical (role S), 1: the workers' list
    int n[2];
    int isolate;                 // the critical workgroups keep their compute units to themselves
    int nap;                     // longest pause between two looks at a waiting task's dependencies, in units of 64 clocks (8, 16, 32, 64 or 127)
    long long* trace;            // optional: [p][4] critical-path stamps, then [crit task][2]
    long long tmo;               // spin bound in wall-clock ticks (100 MHz)
    long long* tasklog;          // optional (trace level 2): per workgroup TG_LOG_CAP records {task (2 words), start, end}
};

// control block (ints): [0] arrivals, [32] abort (1 = not positive definite, 2 = a spin gave up), [64 + 32 q] list heads,
// then diag[nPad], quad[nPad], solved[2 nPad], seq[nP * nP], and per compute unit (key = xcc | se | sh | cu, 12 bits)
// the number of workgroups that have started there and the role of the first one.
// Default chunks (sweeps in profiles/history/r04_chol_taskgraph.txt): 1, 2, 4, 8, 16, 16, .. blocks counted back from the pivot.
constexpr int TG_DEFAULT_CHUNKS = 112489;     // 1, 1, 2, 4, 8, 16, 16, ..: the two chunks next to the pivot are single block rows (the shadows' U0 / U, V / V2); the step from 4 to 16 cost 10 % at N = 4096 (1.45 -> 1.29 ms) and 6 % at 8192
constexpr int TG_NEAR_CHUNKS = 11112489;     // ... and up to TG_NEAR_MAX blocks 1, 1, 1, 1, 2, 4, 8, 16, ..: at the chain-bound sizes every multi-block chunk next to the
constexpr int TG_NEAR_MAX = 36;              // pivot is a 40-us worker task in front of a shadow (N = 2048 0.589 -> 0.548 ms, 4096 1.22 -> 1.155; from N = 5000 on it is neutral, at 8192 it costs 2 %)
inline int tg_default_chunks(int nP) { return nP <= TG_NEAR_MAX ? TG_NEAR_CHUNKS : TG_DEFAULT_CHUNKS; }
constexpr int TG_LOG_CAP = 1024, TG_LOG_WGS = 1024;
constexpr int TG_NPIECE = 6;          // what role U stores into quad[p] when the diagonal tile p is ready for role C (the update once came in six pieces)
constexpr int TG_NSHADOW = 9;         // roles 1 .. 9: S1, S2, S3, U, U0, V (two column halves), V2 (two) -- role 0 is C; workers from 10 on
constexpr int TG_CTL_ABORT = 32, TG_CTL_HEAD = 64, TG_CTL_STEP = 128, TG_CTL_XSTEP = 160, TG_CTL_XSTEP2 = 192, TG_CTL_XSTEP3 = 224, TG_CTL_BASE = 256, TG_CU_KEYS = 4096;
__host__ __device__ inline int tg_npad(int nP) { return (nP + 31) / 32 * 32; }
__host__ __device__ inline int tg_ctl_hf(int nP) { return TG_CTL_BASE + 4 * tg_npad(nP) + nP * nP + 2 * TG_CU_KEYS; }      // half-tile flags [nP * nP * 2], then arrival counts [nP * nP]
__host__ __device__ inline int tg_ctl_ints(int nP) { return tg_ctl_hf(nP) + 3 * nP * nP; }

__device__ __forceinline__ int ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sti(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a worker's task may start: a solve when the diagonal block is factored and its tile has every chunk; an update when the
// tile's earlier chunks are applied and the block rows it applies are solved in both block columns; a fused link (see
// tg_do_trsmu) when its half of tile (I, J) is final -- the same half's link of the row above said so; the diagonal block and
// everything else are waited for inside the task, behind the loads of the right-hand sides
__device__ __forceinline__ bool tg_deps_met(const TgTask& t, const int* dd, const int* sv, const int* sq, int nP) {
    const int I = t.I, J = t.J;
    if (t.type == TG_TRSMU) return I == 0 || ldi(dd + tg_ctl_hf(nP) - TG_CTL_BASE + (I * nP + J) * 2 + t.aux) != 0;
    const int s = ldi(sq + I * nP + J);
    if (t.type == TG_TRSM) return (ldi(dd + I) != 0) && (s == t.ord);
    const int s0 = ldi(sv + 2 * I), s1 = ldi(sv + 2 * I + 1), s2 = ldi(sv + 2 * J), s3 = ldi(sv + 2 * J + 1);
    return (s == t.ord) && (min(min(s0, s1), min(s2, s3)) >= t.k1);
}

struct TgHeld { TgTask t; int have; };     // in LDS, one per workgroup: the ticket in hand (have: 0 none, 1 a task, 2 the list ran out)

// The next task of the workers' list for this workgroup.  Called by ONE full wave; lane 0 works, the
// result is the same in every lane: 1 (task in `out`), 0 (the list is exhausted) or -1 (abort).
// A workgroup without a ticket draws one at once (ONE fetch-and-add) and waits for that task's dependencies with the
// ticket in hand.  (History, profiles/history/r04_chol_taskgraph.txt: claiming a head with compare-and-swap after checking its
// dependencies serialised the chip -- 1.3 us per task with 123 workers, 5.9 us with 507, N = 8192 in 107 ms; peeking at the
// head before drawing made every idle workgroup rush for the one head that had just become ready: 1-2 % slower.)
__device__ __forceinline__ int tg_take(const TgArgs& a, TgTask& out, int lane, TgHeld* held) {
    const int nP = a.nP, npad = tg_npad(nP);
    int* ctl = a.ctl;
    const int* dd = ctl + TG_CTL_BASE;
    const int* sv = dd + 2 * npad;
    const int* sq = sv + 2 * npad;
    const long long t0 = wall_clock64();
    int nap = 0;                                   // polls since the draw: the pauses grow
    int* head = ctl + TG_CTL_HEAD + 32;
    const int nq = a.n[1];
    const TgTask* tq = a.q[1];
    for (unsigned spins = 0;; ++spins) {
        if (ldi(ctl + TG_CTL_ABORT) != 0) return -1;
        int code = 0;                              // 1: the held task is ready, 2: the list ran out
        union { TgTask t; int4 v; } u;
        u.v = make_int4(0, 0, 0, 0);
        if (lane == 0) {
            if (held->have == 0) {
                const int k = (nq > 0) ? atomicAdd(head, 1) : nq;
                if (k < nq) {
                    held->t = *reinterpret_cast<const TgTask*>(tq + k);
                    held->have = 1;
                } else {
                    held->have = 2;
                }
            }
            if (held->have == 2) {
                code = 2;
            } else {
                u.t = held->t;
                if (tg_deps_met(u.t, dd, sv, sq, nP)) {
                    held->have = 0;
                    code = 1;
                }
            }
        }
        code = __shfl(code, 0);
        if (code == 2) return 0;
        if (code == 1) {
            u.v.x = __shfl(u.v.x, 0); u.v.y = __shfl(u.v.y, 0);
            u.v.z = __shfl(u.v.z, 0); u.v.w = __shfl(u.v.w, 0);
            out = u.t;
            return 1;
        }
        // pauses between two looks: 256 clocks at first, then a.nap x 64 (default 16: 0.4 us; until late in round 4 the long
        // pause was 127 x 64 clocks = 3.4 us -- half of that, on average, between a dependency's arrival and the task's start:
        // N = 2048 0.945 -> 0.876 ms, 8192 5.30 -> 5.19; profiles/history/r04_chol_tg_polling_ab.txt)
        if (nap < 4) __builtin_amdgcn_s_sleep(4);
        else if (a.nap <= 8) __builtin_amdgcn_s_sleep(8);
        else if (a.nap <= 16) __builtin_amdgcn_s_sleep(16);
        else if (a.nap <= 32) __builtin_amdgcn_s_sleep(32);
        else if (a.nap <= 64) __builtin_amdgcn_s_sleep(64);
        else __builtin_amdgcn_s_sleep(127);
        ++nap;
        if ((spins & 63) == 63 && wall_clock64() - t0 > a.tmo) {
            if (lane == 0) { sti(ctl + TG_CTL_ABORT, 2); sti(a.dflag + 1, 2); }
            return -1;
        }
    }
}

// The workgroup's LDS (dynamic: the size is the launch's): 8 doubles of control words -- the task in hand (2), role / flags
// (2), the held ticket (20 bytes) -- then the tile engine's buffer: ONE k-step image (72 KB; two workgroups per CU; the
// diagonal kernel's panels fit inside) or TWO (144 KB: a workgroup alone on its CU runs the double-buffered k-loop,
// gemm_tile_128_d).  File scope, so that the role bodies below can be separate (non-inlined) functions with their own register
// allocation and still address it with ds_* instructions: inlined into one kernel body the roles spilled 319 VGPRs.
constexpr int TG_CTL_F64 = 32;        // control words: [0..1] the task in hand, [2..3] codes, [4..6] the held ticket, [8..15] the dispatcher's trace sums, [16..31] the argument block
extern __shared__ __attribute__((aligned(16))) double tg_smem[];
#define tg_buf (tg_smem + TG_CTL_F64)
// Calls of the role / task bodies must not carry LLVM's `tail` marker (the optimiser sets it on any call that cannot see the
// caller's frame): a function with a marked call site loses the no-callee-saved-registers treatment and saves ~100 VGPRs to
// scratch on entry.
#define TG_BODY __noinline__ __attribute__((not_tail_called))

// The launch's argument block as the role bodies see it: a copy in the workgroup's LDS (written once by the kernel's first
// instructions).  The role bodies are separate functions; handed the kernel's by-value block by reference they forced a copy of it
// into SCRATCH memory, and the dispatcher's live registers were spilled around every call (round 5: 672 bytes of scratch per
// lane, 112 scratch instructions in the dispatcher).  (__builtin_amdgcn_kernarg_segment_ptr() inside a callee folds to a null
// pointer with this compiler: not an option.)
static_assert(sizeof(TgArgs) <= 16 * sizeof(double), "TgArgs must fit its LDS slot");
__device__ __forceinline__ const TgArgs& tg_kargs() { return *reinterpret_cast<const TgArgs*>(tg_smem + 16); }

// The last dependency of a task in hand (every thread calls; thread 0 polls): both flags >= need.  false: abort.
__device__ __forceinline__ bool tg_wait_flags(const TgArgs& a, const int* f0, const int* f1, int need) {
    int* code = reinterpret_cast<int*>(tg_smem + 2);
    if (threadIdx.x == 0) {
        int ok = 1;
        const long long t0 = wall_clock64();
        for (unsigned spins = 0; min(ldi(f0), ldi(f1)) < need; ++spins) {
            if (ldi(a.ctl + TG_CTL_ABORT) != 0) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 255) == 255 && wall_clock64() - t0 > a.tmo) { sti(a.ctl + TG_CTL_ABORT, 2); sti(a.dflag + 1, 2); ok = 0; break; }
        }
        code[2] = ok;
    }
    __syncthreads();
    return code[2] != 0;
}

// The panel solve of fit_tiles.h with the factor's diagonal block staged in LDS.  panel_solve16_body reads its A fragments
// from global memory one step ahead: eight dependent round trips, ~1.4 us each through L1 -- but ~3 us each past it (the
// agent-scope loads this kernel needs), 25 us per solve on the critical path.  Here the 36 upper 16-tiles of R_pp (72 KB =
// the tile engine's LDS buffer, exactly) arrive with 18 16-byte loads per thread in flight at once, together with the
// right-hand sides and the eight 16 x 16 inverses (registers): ONE round trip, then the substitution runs from LDS.
// Tile (r, c), r <= c, sits at 256 * (8 r - r (r - 1) / 2 + c - r), row-major 16 x 16: a fragment read (lane (g, n) <-
// row 4 kk + g, column n) is 64 consecutive doubles.  Same MFMAs in the same order as panel_solve16_body.
__device__ __forceinline__ int tri_index(int r, int c) { return 8 * r - r * (r - 1) / 2 + c - r; }

__device__ __forceinline__ bool panel_solve16_lds(const TgArgs& a, const double* __restrict__ U, const double* __restrict__ S,
                                                  double* __restrict__ R, int64_t Np, int p, int cb, double* lds,
                                                  const int* diag) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int64_t p0 = (int64_t)p * NB;
    const int64_t j0 = (int64_t)(p + 1) * NB + (int64_t)cb * 64 + 16 * w;
    const double* Rd = R + p0 * Np + p0;          // R_pp
    const double* Ud = U + p0 * Np + p0;          // diagonal 16-tiles hold T_d^T
    // (0) the right-hand sides are final before the diagonal block is: their loads travel while it is awaited
    d4 X[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) X[r][q] = ldg<true>(S + (p0 + 16 * r + g + 4 * q) * Np + j0 + n);
    if (!tg_wait_flags(a, diag + p, diag + p, 1)) return false;
    // (1) everything else this workgroup reads, issued back to back
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Rd), 0, (int)(128 * Np * 8), 0x00020000);
    u4v stage[18];
    // piece e = t + 256 q of the 36 tiles x 128 16-byte pieces: tile e >> 7 = 2 q + (t >> 7), so its LDS position is
    // 512 q + (t >> 7) * 256 + row * 16 + 2 c2 -- one register for all 18 pieces (18 of them were spilled to scratch memory)
    const int piece = t & 127, row = piece >> 3, c2 = piece & 7;
    const int dst0 = (t >> 7) * 256 + row * 16 + 2 * c2;
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        int r = 0;
        int idx = 2 * q + (t >> 7);
        while (idx >= 8 - r) { idx -= 8 - r; ++r; }
        const int c = r + idx;
        stage[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((16 * r + row) * Np + 16 * c + 2 * c2) * 8), 0, 16);
    }
    double ti[8][4];                               // A fragments of the eight T_d
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ti[jb][kk] = ldg<true>(Ud + (int64_t)(16 * jb + 4 * kk + g) * Np + 16 * jb + n);
#pragma unroll
    for (int q = 0; q < 18; ++q) *reinterpret_cast<u4v*>(lds + dst0 + 512 * q) = stage[q];
    __syncthreads();
    // (2) the substitution
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ti[jb][kk], X[jb][kk], x, 0, 0, 0);
        X[jb] = x;
        const d4 xn = -x;
#pragma unroll
        for (int i = jb + 1; i < 8; ++i) {
            const double* tl = lds + 256 * tri_index(jb, i) + g * 16 + n;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) X[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(tl[kk * 64], xn[kk], X[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) stg<true>(R + (p0 + 16 * r + g + 4 * q) * Np + j0 + n, X[r][q]);
    __syncthreads();                               // the LDS image is free again
    return true;
}

// the strictly-lower 16-tiles of R's diagonal blocks are zero (potrf16_body writes them itself in the stream schedule)
__global__ __launch_bounds__(256) void k_zero_diag_lower(double* __restrict__ R, int64_t Np) {
    const int64_t p0 = (int64_t)blockIdx.x * NB;
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx4 = t + 256 * q;
        const int r = idx4 >> 5, c = (idx4 & 31) * 4;
        if ((c >> 4) < (r >> 4)) *reinterpret_cast<d4*>(R + (p0 + r) * Np + p0 + c) = (d4){0.0, 0.0, 0.0, 0.0};
    }
}


// a wave-uniform 64-bit value the compiler cannot prove uniform (fields of the argument block reached through a reference)
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <class T>
__device__ __forceinline__ T* uni(T* p) { return reinterpret_cast<T*>(uni64(reinterpret_cast<unsigned long long>(p))); }

__device__ __forceinline__ void tg_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ TG_BODY void tg_role_diag() {
    const TgArgs& a = tg_kargs();
    double* Pn = tg_buf;                           // 2 x 16 x PFP
    double* Ud = tg_buf + 2 * 16 * PFP;            // 256
    int* code = reinterpret_cast<int*>(tg_smem + 2);
    int& sflag = code[1];
    const int t = threadIdx.x;
    const int nP = a.nP, npad = tg_npad(nP);
    int* ctl = a.ctl;
    int* dd = ctl + TG_CTL_BASE;
    int* qd = dd + npad;
    for (int p = 0; p < nP; ++p) {
        if (t == 0) {
            int c = 1;
            if (a.trace) a.trace[4 * p] = wall_clock64();
            if (p > 0) {
                const long long t0 = wall_clock64();
                for (unsigned spins = 0; ldi(qd + p) != TG_NPIECE; ++spins) {
                    if (ldi(ctl + TG_CTL_ABORT) != 0) { c = -1; break; }
                    __builtin_amdgcn_s_sleep(2);
                    if ((spins & 255) == 255 && wall_clock64() - t0 > a.tmo) { sti(ctl + TG_CTL_ABORT, 2); sti(a.dflag + 1, 2); c = -1; break; }
                }
            }
            code[0] = c;
            if (a.trace) a.trace[4 * p + 1] = wall_clock64();
        }
        __syncthreads();
        if (code[0] != 1) break;
        potrf16_body<false, true, true>(a.S, a.R, a.T, a.U, a.Np, p, a.dflag, nullptr, Pn, Ud, sflag, ctl + TG_CTL_STEP, 8 * p);
        tg_drain();
        __syncthreads();
        if (sflag) {
            if (t == 0) sti(ctl + TG_CTL_ABORT, 1);
            break;
        }
        if (t == 0) {
            sti(dd + p, 1);
            sti(ctl + TG_CTL_STEP, 8 * p + 8);
            if (a.trace) a.trace[4 * p + 2] = wall_clock64();
        }
    }
}

// ---- the shadows of the diagonal factorisation: roles S1, S2, S3, U0, U, V, V2 -------------------------------------------
// Between POTRF(p) and POTRF(p+1) lie the solve of tile (p, p+1) and the update of the diagonal tile (p+1, p+1) with it.
// As tasks (two solve halves, six update pieces: round 4) they started from nothing when the diagonal block's flag went up and
// cost three hand-offs of 4-5 us on the critical path: 28 us of diagonal block + 26-28 us of hand-offs per block row.  The
// shadows do both WHILE the diagonal block is being factored: role C publishes a counter after each of its eight 16-row steps
// (potrf16_body, STEP), and dedicated workgroups follow it -- and each other -- one step at a time:
//   S1  advances the substitution of tile (p, p+1) by one step per count -- x_jb = T_d(jb) s_jb, s_i -= R[jb, i]^T x_jb, the
//       MFMAs of panel_solve16_lds in the same order -- and publishes its own counter per 16 solved rows;
//   U   applies every 16 rows S1 has solved to the diagonal tile (p+1, p+1) (rank-16 update; k ascends 4 at a time as in the
//       tile engines: bit-identical) and hands the tile to role C;
//   S2, S3  do S1's work for tiles (p, p+2) and (p, p+3), with counters of their own;
//   U0  applies block row p-1 to the diagonal tile (p+1, p+1) behind S2 of block row p-1: U finds the tile complete up to its
//       own block row instead of waiting for a worker's 24-us update that could only start when S2 was done;
//   V, V2  apply block row p-1 to tiles (p, p+1) and (p, p+2) -- the right-hand sides of S1 and S2 -- behind S1 and S2 (S3) of
//       block row p-1, for the same reason.
// What is left between two diagonal blocks when everything is on time: S1's last step (one poll, one 16 x 16 load, 8 MFMAs per
// wave, its stores), U's last step (one poll, one 16-row load, 36 MFMAs, the tile's store) and role C's load of the tile:
// 8-9 us (the first block rows of a factorisation run at 37-38 us per block row; measured: profiles/history/r05_chol_taskgraph.txt).
// What keeps everything on time is the speed of the column chains right of the band (tg_do_trsmu).
// Operands travel global memory -> LDS without registers (global_load_lds_dwordx4; rings of 16-row panels, up to three steps
// ahead of the arithmetic when a shadow starts late and the counts are already up); each wave counts the vector-memory
// operations it has issued since a panel's loads and waits with the matching vmcnt; all bookkeeping is wave-uniform (SGPRs).
// Wave w owns 32 columns of a solve (8 x 2 accumulator tiles), in U / U0 the diagonal tile's column tiles w and 7 - w (9
// tiles), in a V / V2 workgroup 16 columns of its 64-column half (8 tiles).
__device__ __forceinline__ int tg_peek(const int* f) { return __builtin_amdgcn_readfirstlane(ldi(f)); }
__device__ __forceinline__ bool tg_wave_wait_ge(const TgArgs& a, const int* f, int need) {
    const long long t0 = wall_clock64();
    for (unsigned spins = 0; tg_peek(f) < need; ++spins) {
        if (tg_peek(a.ctl + TG_CTL_ABORT) != 0) return false;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255) == 255 && wall_clock64() - t0 > a.tmo) { sti(a.ctl + TG_CTL_ABORT, 2); sti(a.dflag + 1, 2); return false; }
    }
    return true;
}
// wait until at most n (wave-uniform, never more than were really issued since) vector-memory operations are outstanding
__device__ __forceinline__ void tg_vmcnt_le(int n) {
    n = __builtin_amdgcn_readfirstlane(n);
    if (n >= 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else if (n >= 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else if (n >= 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 8-byte agent-scope accesses through a buffer descriptor: ONE lane-offset register per tile walk, everything else in SGPRs
// (with 64-bit flat addresses the compiler precomputed one address pair per access and spilled them)
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tg_rsrc(const double* base, int64_t Np) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, (int)(128 * Np * 8), 0x00020000);
}
__device__ __forceinline__ double tg_bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    union { u2v v; double d; } u;
    u.v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 16);
    return u.d;
}
__device__ __forceinline__ void tg_bstore(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double x) {
    union { u2v v; double d; } u;
    u.d = x;
    __builtin_amdgcn_raw_buffer_store_b64(u.v, rs, voff, soff, 16);
}

// tile i = 0 .. 8 of wave w in the diagonal tile: column tile w (rows 0 .. w), then column tile 7 - w (rows 0 .. 7 - w) -- the
// same nine registers in every wave, the tile's position a wave-uniform number
__device__ __forceinline__ void tg_sh_tile(int i, int w, int& r, int& c) {
    const bool first = i <= w;
    r = first ? i : i - w - 1;
    c = first ? w : 7 - w;
}
constexpr int TG_SH_PAN = 16 * PFP;            // one staged panel (f64)
typedef __attribute__((address_space(3))) void* lds_ptr;

// a follower's view of the counter it follows, per wave (all wave-uniform)
struct TgFollow {
    int avail;            // steps of this block known to be published (0 .. 8)
    int issued;           // steps whose loads have been issued
    int vmi;              // vector-memory operations issued by this wave (an under-count is safe, an over-count is not)
    int mark[8];          // vmi right after the loads of step jb
};

// S roles: panel jb = rows 16 jb .. 16 jb + 15 of R_pp (wave w: rows 4 w .. 4 w + 3, one 1024-byte row per instruction, lane l:
// columns 2 l, 2 l + 1; nothing right of the diagonal tile in the last panel) and the 16 x 16 inverse T_d(jb)^T (32 lanes)
__device__ __forceinline__ void tg_s_issue(int jb, const double* __restrict__ Rd, const double* __restrict__ Ud, int64_t Np,
                                           double* __restrict__ lds, TgFollow& f) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = jb % 3;
    double* pan = lds + b * TG_SH_PAN;
    double* td = lds + 3 * TG_SH_PAN + b * 256;
    if (jb < 7) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            __builtin_amdgcn_global_load_lds(Rd + (int64_t)(16 * jb + 4 * w + rr) * Np + 2 * lane, (lds_ptr)(pan + (4 * w + rr) * PFP), 16, 0, 16);
        f.vmi += 4;
    }
    if (lane < 32)
        __builtin_amdgcn_global_load_lds(Ud + (int64_t)(16 * jb + 4 * w + (lane >> 3)) * Np + 16 * jb + 2 * (lane & 7), (lds_ptr)(td + 64 * w), 16, 0, 16);
    f.vmi += 1;
    f.mark[jb] = f.vmi;
}
// U role: the 16 rows S1 solved in step jb (rows 16 jb .. of tile (p, p+1))
__device__ __forceinline__ void tg_u_issue(int jb, const double* __restrict__ Xg, int64_t Np, double* __restrict__ lds, TgFollow& f) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* xb = lds + (jb & 3) * TG_SH_PAN;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
        __builtin_amdgcn_global_load_lds(Xg + (int64_t)(16 * jb + 4 * w + rr) * Np + 2 * lane, (lds_ptr)(xb + (4 * w + rr) * PFP), 16, 0, 16);
    f.vmi += 4;
    f.mark[jb] = f.vmi;
}

// The loads of step JB are on their way (issued here if they were not: that may wait for the counter), then landed in every
// wave's share (barrier); then, the counter permitting, the next two steps' loads go out.  SROLE: tg_s_issue, else tg_u_issue.
// DRAIN: this wave's earlier stores are complete as well before the barrier (S1 publishes the previous step behind it).
// (Steps are issued in order, so the step to issue is always one of JB, JB + 1, JB + 2: compile-time slots of `mark`.)
// V role: the 16 rows S1 and S2 solved in step jb of the PREVIOUS block row (tiles (p-1, p) and (p-1, p+1)): two panels
// (RM + 1 = pairs of panels in the ring: 2 with one k-step image of LDS, 4 with two)
template <int RM>
__device__ __forceinline__ void tg_v_issue(int jb, const double* __restrict__ Ag, const double* __restrict__ Bg, int64_t Np,
                                           double* __restrict__ lds, TgFollow& f) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* pa = lds + (2 * (jb & RM)) * TG_SH_PAN;
    double* pb = pa + TG_SH_PAN;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        __builtin_amdgcn_global_load_lds(Ag + (int64_t)(16 * jb + 4 * w + rr) * Np + 2 * lane, (lds_ptr)(pa + (4 * w + rr) * PFP), 16, 0, 16);
        __builtin_amdgcn_global_load_lds(Bg + (int64_t)(16 * jb + 4 * w + rr) * Np + 2 * lane, (lds_ptr)(pb + (4 * w + rr) * PFP), 16, 0, 16);
    }
    f.vmi += 8;
    f.mark[jb] = f.vmi;
}

// KIND: 0 = a solve shadow (tg_s_issue), 1 = U (tg_u_issue), 2 / 3 = V with a ring of 2 / 4 panel pairs (tg_v_issue)
template <int J, int KIND>
__device__ __forceinline__ void tg_issue(TgFollow& f, double* __restrict__ lds, const double* __restrict__ A0,
                                         const double* __restrict__ A1, int64_t Np) {
    if constexpr (J < 8) {
        if constexpr (KIND == 0) tg_s_issue(J, A0, A1, Np, lds, f);
        else if constexpr (KIND == 1) tg_u_issue(J, A0, Np, lds, f);
        else if constexpr (KIND == 2) tg_v_issue<1>(J, A0, A1, Np, lds, f);
        else tg_v_issue<3>(J, A0, A1, Np, lds, f);
        f.issued = J + 1;
    }
}
// The loads of step JB are on their way (issued here if they were not: that may wait for the counter -- for both counters
// where a role follows two), then landed in every wave's share (barrier); then, the counter(s) permitting, the next steps'
// loads go out (as far ahead as the role's LDS ring is deep).  DRAIN: this wave's earlier stores are complete as well before
// the barrier (a publishing role announces the previous step behind it).
// (Steps are issued in order, so the step to issue is always one of JB .. JB + 3: compile-time slots of `mark`.)
__device__ __forceinline__ int tg_peek2(const int* c1, const int* c2) {
    const int v = tg_peek(c1);
    return c2 ? min(v, tg_peek(c2)) : v;
}
template <int JB, int KIND, bool DRAIN>
__device__ __forceinline__ bool tg_follow(const TgArgs& a, const int* counter, const int* counter2, int base, TgFollow& f,
                                          double* __restrict__ lds, const double* __restrict__ A0, const double* __restrict__ A1,
                                          int64_t Np) {
    // (the bookkeeping is wave-uniform: say so, or the compiler keeps it in vector registers and branches on exec masks)
    f.avail = __builtin_amdgcn_readfirstlane(f.avail); f.issued = __builtin_amdgcn_readfirstlane(f.issued);
    f.vmi = __builtin_amdgcn_readfirstlane(f.vmi);
    base = __builtin_amdgcn_readfirstlane(base);
    if (f.issued == JB) {
        if (f.avail <= JB) {
            if (!tg_wave_wait_ge(a, counter, base + JB + 1)) return false;
            if (counter2 && !tg_wave_wait_ge(a, counter2, base + JB + 1)) return false;
            f.avail = min(8, tg_peek2(counter, counter2) - base);
        }
        tg_issue<JB, KIND>(f, lds, A0, A1, Np);
    }
    if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else tg_vmcnt_le(f.vmi - __builtin_amdgcn_readfirstlane(f.mark[JB]));
    __syncthreads();
    constexpr int ahead = (KIND == 0) ? 2 : (KIND == 2 ? 1 : 3);     // rings of 3 (S: panel + inverse), 4 (U), 2 or 4 pairs (V) buffers
    constexpr int lim = (JB + 1 + ahead < 8) ? JB + 1 + ahead : 8;
    if (f.issued < lim) {
        if (f.avail < lim) f.avail = max(f.avail, min(8, tg_peek2(counter, counter2) - base));
        if (f.issued == JB + 1 && f.avail > JB + 1) tg_issue<JB + 1, KIND>(f, lds, A0, A1, Np);
        if (ahead >= 2 && f.issued == JB + 2 && f.avail > JB + 2) tg_issue<JB + 2, KIND>(f, lds, A0, A1, Np);
        if (ahead >= 3 && f.issued == JB + 3 && f.avail > JB + 3) tg_issue<JB + 3, KIND>(f, lds, A0, A1, Np);
    }
    return true;
}

// one step of a solve shadow: x_JB and the updates of the rows below it (panel_solve16_lds's MFMAs in the same order)
template <int JB, int XCTR>
__device__ __forceinline__ bool tg_s_step(const TgArgs& a, int p, TgFollow& f, d4 (&X)[8][2], double* __restrict__ lds,
                                          const double* __restrict__ Rd, const double* __restrict__ Ud,
                                          __amdgpu_buffer_rsrc_t rout, int vo, int64_t Np) {
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), g = lane >> 4, n = lane & 15;
    if (!tg_follow<JB, 0, true>(a, a.ctl + TG_CTL_STEP, nullptr, 8 * p, f, lds, Rd, Ud, Np)) return false;
    if (JB > 0 && t == 0) sti(a.ctl + XCTR, 8 * p + JB);      // rows 16 (JB - 1) .. are in memory
    const double* pan = lds + (JB % 3) * TG_SH_PAN;
    const double* td = lds + 3 * TG_SH_PAN + (JB % 3) * 256;
    double ti[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ti[kk] = td[(4 * kk + g) * 16 + n];
    d4 xn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ti[kk], X[JB][j][kk], x, 0, 0, 0);
        xn[j] = -x;
#pragma unroll
        for (int q = 0; q < 4; ++q) tg_bstore(rout, vo, (int)((((16 * JB + 4 * q) * Np) + 32 * w + 16 * j) * 8), x[q]);
    }
    f.vmi += 8;
    // (all fragments of the step first: left to itself the compiler waits for every LDS read right in front of its MFMA)
#pragma unroll
    for (int i0 = JB + 1; i0 < 8; i0 += 4) {
        double af[4][4];
#pragma unroll
        for (int i = i0; i < i0 + 4 && i < 8; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) af[i - i0][kk] = pan[(4 * kk + g) * PFP + 16 * i + n];
#pragma unroll
        for (int i = i0; i < i0 + 4 && i < 8; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) X[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i - i0][kk], xn[j][kk], X[i][j], 0, 0, 0);
    }
    return true;
}

// S1 (off = 1, PUB) and S2 (off = 2): the solve of tile (p, p + off), p = 0, 1, ..
template <int XCTR>
__device__ __forceinline__ void tg_role_solve(const TgArgs& a, int off) {
    constexpr int TSLOT = (XCTR == TG_CTL_XSTEP) ? 0 : (XCTR == TG_CTL_XSTEP2 ? 4 : -1);     // trace slots of S1 / S2
    double* lds = tg_buf;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), g = lane >> 4, n = lane & 15;
    const int nP = __builtin_amdgcn_readfirstlane(a.nP), npad = tg_npad(nP);
    const int64_t Np = (int64_t)uni64((unsigned long long)a.Np);
    double* R = uni(a.R);
    const double* S = uni(a.S);
    const double* U = uni(a.U);
    int* ctl = uni(a.ctl);
    int* dd = ctl + TG_CTL_BASE;
    int* sv = dd + 2 * npad;
    int* sq = sv + 2 * npad;
    __builtin_amdgcn_s_setprio(3);
    for (int p = 0; p + off < nP; ++p) {
        const int ord = __builtin_amdgcn_readfirstlane((int)a.q[0][p].ord);
        const int64_t p0 = (int64_t)p * NB, j0 = p0 + (int64_t)off * NB;
        long long ts0 = 0, ts1 = 0;
        if (a.trace && t == 0) ts0 = wall_clock64();
        // the right-hand sides: tile (p, p + off) with every chunk applied
        if (!tg_wave_wait_ge(a, sq + p * nP + p + off, ord)) return;
        const __amdgpu_buffer_rsrc_t rin = tg_rsrc(S + p0 * Np + j0, Np), rout = tg_rsrc(R + p0 * Np + j0, Np);
        const int vo = (int)((g * Np + n) * 8);
        d4 X[8][2];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) X[r][j][q] = tg_bload(rin, vo, (int)((((16 * r + 4 * q) * Np) + 32 * w + 16 * j) * 8));
        if (a.trace && t == 0) ts1 = wall_clock64();
        const double* Rd = R + p0 * Np + p0;
        const double* Ud = U + p0 * Np + p0;
        TgFollow f;
        f.avail = 0; f.issued = 0; f.vmi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) f.mark[i] = 0;
        if (!tg_s_step<0, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        if (!tg_s_step<1, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        if (!tg_s_step<2, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        if (!tg_s_step<3, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        if (!tg_s_step<4, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        if (!tg_s_step<5, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        if (!tg_s_step<6, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        if (!tg_s_step<7, XCTR>(a, p, f, X, lds, Rd, Ud, rout, vo, Np)) return;
        tg_drain();
        __syncthreads();
        if (t == 0) {
            sti(ctl + XCTR, 8 * p + 8);
            sti(sv + 2 * (p + off), p + 1);
            sti(sv + 2 * (p + off) + 1, p + 1);
            if (a.trace && TSLOT >= 0) {         // slots 0, 1 (S1) / 2, 3 (S2) of block row p: waiting for the right-hand sides, then the eight steps
                long long* o = a.trace + 4 * nP + 2 * 8 * p + TSLOT;
                o[0] = ts0; o[1] = ts1; o[2] = ts1; o[3] = wall_clock64();
            }
        }
    }
}
__device__ TG_BODY void tg_role_s1() { tg_role_solve<TG_CTL_XSTEP>(tg_kargs(), 1); }
__device__ TG_BODY void tg_role_s2() { tg_role_solve<TG_CTL_XSTEP2>(tg_kargs(), 2); }
__device__ TG_BODY void tg_role_s3() { tg_role_solve<TG_CTL_XSTEP3>(tg_kargs(), 3); }

// acc(r, c) -= x_r^T x_c as acc += x_r^T (-x_c): the same products with the same signs as the tile engines' (-A) B
template <int W>
__device__ __forceinline__ void tg_u_mfma(d4 (&acc)[9], const double (&fb)[8][4]) {
    double nb[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { nb[0][kk] = -fb[W][kk]; nb[1][kk] = -fb[7 - W][kk]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int r = (i <= W) ? i : i - W - 1, cs = (i <= W) ? 0 : 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[r][kk], nb[cs][kk], acc[i], 0, 0, 0);
    }
}

template <int JB>
__device__ __forceinline__ bool tg_u_step(const TgArgs& a, const int* counter, int base, TgFollow& f, d4 (&acc)[9],
                                          double* __restrict__ lds, const double* __restrict__ Xg, int64_t Np) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, n = lane & 15;
    if (!tg_follow<JB, 1, false>(a, counter, nullptr, base, f, lds, Xg, nullptr, Np)) return false;
    const double* xb = lds + (JB & 3) * TG_SH_PAN + g * PFP + n;
    // every fragment of the 16 rows once (the A fragment of row tile r IS the B fragment of column tile r), then the 36 MFMAs;
    // one body per wave so that a tile's fragments are registers chosen at compile time
    double fb[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb[r][kk] = xb[4 * kk * PFP + 16 * r];
    if (w == 0) tg_u_mfma<0>(acc, fb);
    else if (w == 1) tg_u_mfma<1>(acc, fb);
    else if (w == 2) tg_u_mfma<2>(acc, fb);
    else tg_u_mfma<3>(acc, fb);
    return true;
}

// U (LAST): the final chunk of the diagonal tile (p+1, p+1) -- block row p, 16 rows at a time behind S1 of block row p -- handed to
// role C.  U0 (!LAST): the chunk before it -- block row p-1 (and, where the row's first chunks were merged, the rows before it) --
// behind S2 of block row p-1, handed to U: as a worker task it started when S2 was done and took 24 us.
template <bool LAST>
__device__ __forceinline__ void tg_role_diagupd(const TgArgs& a) {
    double* lds = tg_buf;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), g = lane >> 4, n = lane & 15;
    const int nP = __builtin_amdgcn_readfirstlane(a.nP), npad = tg_npad(nP);
    const int64_t Np = (int64_t)uni64((unsigned long long)a.Np);
    const double* R = uni(a.R);
    double* S = uni(a.S);
    int* ctl = uni(a.ctl);
    int* dd = ctl + TG_CTL_BASE;
    int* qd = dd + npad;
    int* sv = qd + npad;
    int* sq = sv + 2 * npad;
    __builtin_amdgcn_s_setprio(3);
    for (int p = LAST ? 0 : 1; p + 1 < nP; ++p) {
        union { TgTask t; int4 v; } u;
        u.v = *reinterpret_cast<const int4*>(a.q[0] + p);
        u.v.x = __builtin_amdgcn_readfirstlane(u.v.x); u.v.y = __builtin_amdgcn_readfirstlane(u.v.y);
        u.v.z = __builtin_amdgcn_readfirstlane(u.v.z); u.v.w = __builtin_amdgcn_readfirstlane(u.v.w);
        const TgTask d = u.t;
        if (!LAST && d.rsv < 0) continue;              // the tile has no chunk of its own before the final one
        // the chunk this role applies: block rows [ck0, ck1), the last of them (kb) from the shadow it follows
        const int ck0 = LAST ? d.k0 : d.rsv, kb = LAST ? p : p - 1, need = LAST ? d.aux : d.aux - 1;
        const int64_t p0 = (int64_t)kb * NB, i0 = (int64_t)(p + 1) * NB;
        long long ts0 = 0, tsq = 0;
        if (a.trace && t == 0) ts0 = wall_clock64();
        // the diagonal tile (p+1, p+1) with its earlier chunks applied
        if (!tg_wave_wait_ge(a, sq + (p + 1) * nP + p + 1, need)) return;
        if (a.trace && t == 0) tsq = wall_clock64();
        const __amdgpu_buffer_rsrc_t rs = tg_rsrc(S + i0 * Np + i0, Np);
        const int vo = (int)((g * Np + n) * 8);
        d4 acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int r, c;
            tg_sh_tile(i, w, r, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = tg_bload(rs, vo, (int)((((16 * r + 4 * q) * Np) + 16 * c) * 8));
        }
        // block rows of the chunk before the followed one (only where the first chunks of a row were merged: block rows 2 and
        // 3): operands straight from global memory
        if (ck0 < kb) {
            if (!tg_wave_wait_ge(a, sv + 2 * (p + 1), kb) || !tg_wave_wait_ge(a, sv + 2 * (p + 1) + 1, kb)) return;
            for (int kr = ck0 * NB; kr < kb * NB; kr += 4) {
                const double* Rk = R + (int64_t)(kr + g) * Np + i0 + n;
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    int r, c;
                    tg_sh_tile(i, w, r, c);
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ldg<true>(Rk + 16 * r), ldg<true>(Rk + 16 * c), acc[i], 0, 0, 0);
                }
            }
        }
        long long ts1 = 0;
        if (a.trace && t == 0) ts1 = wall_clock64();
        const double* Xg = R + p0 * Np + i0;
        const int* counter = ctl + (LAST ? TG_CTL_XSTEP : TG_CTL_XSTEP2);
        const int base = 8 * kb;
        TgFollow f;
        f.avail = 0; f.issued = 0; f.vmi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) f.mark[i] = 0;
        if (!tg_u_step<0>(a, counter, base, f, acc, lds, Xg, Np)) return;
        if (!tg_u_step<1>(a, counter, base, f, acc, lds, Xg, Np)) return;
        if (!tg_u_step<2>(a, counter, base, f, acc, lds, Xg, Np)) return;
        if (!tg_u_step<3>(a, counter, base, f, acc, lds, Xg, Np)) return;
        if (!tg_u_step<4>(a, counter, base, f, acc, lds, Xg, Np)) return;
        if (!tg_u_step<5>(a, counter, base, f, acc, lds, Xg, Np)) return;
        if (!tg_u_step<6>(a, counter, base, f, acc, lds, Xg, Np)) return;
        if (!tg_u_step<7>(a, counter, base, f, acc, lds, Xg, Np)) return;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int r, c;
            tg_sh_tile(i, w, r, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) tg_bstore(rs, vo, (int)((((16 * r + 4 * q) * Np) + 16 * c) * 8), acc[i][q]);
        }
        tg_drain();
        __syncthreads();
        if (t == 0) {
            if (LAST) sti(qd + p + 1, TG_NPIECE);
            else sti(sq + (p + 1) * nP + p + 1, d.aux);
            if (LAST && a.trace) {
                const long long te = wall_clock64();
                long long* o = a.trace + 4 * nP + 2 * 8 * p;
                o[8] = ts0; o[9] = tsq;            // slots 4, 5 (U): waiting for the tile's earlier chunks; loaded .. stored
                o[10] = ts1; o[11] = te;
            }
        }
    }
}

__device__ TG_BODY void tg_role_u() { tg_role_diagupd<true>(tg_kargs()); }
__device__ TG_BODY void tg_role_u0() { tg_role_diagupd<false>(tg_kargs()); }

// V (off = 1) / V2 (off = 2): the final chunk of tile (p, p + off) -- block row p-1, 16 rows at a time behind S1 and S2 (S3) of
// the previous block row -- so that the right-hand sides of S1 (S2) are complete a few microseconds after those two are, not
// one worker task (24 us) later.  A full 128 x 128 x 128 update is 13.7 us of ONE compute unit's matrix pipe -- too long a link of
// that chain (measured: V took 24 us per tile and S1 started 19 us into its diagonal block) -- so each of the two is TWO
// workgroups, one per 64-column half (wave w: 16 columns, all eight tile rows: 8 accumulators); the second half to finish counts
// the tile's chunk.
template <int JB, bool DB>
__device__ __forceinline__ bool tg_v_step(const TgArgs& a, const int* cB, int p, int half, TgFollow& f, d4 (&acc)[8], double* __restrict__ lds,
                                          const double* __restrict__ Ag, const double* __restrict__ Bg, int64_t Np) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, n = lane & 15;
    if (!tg_follow<JB, DB ? 3 : 2, false>(a, a.ctl + TG_CTL_XSTEP, cB, 8 * (p - 1), f, lds, Ag, Bg, Np)) return false;
    const double* pa = lds + (2 * (JB & (DB ? 3 : 1))) * TG_SH_PAN + g * PFP + n;
    const double* pb = pa + TG_SH_PAN + 64 * half + 16 * w;
    double fa[8][4], nb[4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fa[r][kk] = pa[4 * kk * PFP + 16 * r];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) nb[kk] = -pb[4 * kk * PFP];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[r][kk], nb[kk], acc[r], 0, 0, 0);
    return true;
}

template <bool DB>
__device__ __forceinline__ void tg_role_offupd(const TgArgs& a, int off, int half) {
    double* lds = tg_buf;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), g = lane >> 4, n = lane & 15;
    const int nP = __builtin_amdgcn_readfirstlane(a.nP), npad = tg_npad(nP);
    const int64_t Np = (int64_t)uni64((unsigned long long)a.Np);
    const double* R = uni(a.R);
    double* S = uni(a.S);
    int* ctl = uni(a.ctl);
    int* dd = ctl + TG_CTL_BASE;
    int* sv = dd + 2 * npad;
    int* sq = sv + 2 * npad;
    __builtin_amdgcn_s_setprio(3);
    off = __builtin_amdgcn_readfirstlane(off); half = __builtin_amdgcn_readfirstlane(half);
    int* hc = ctl + tg_ctl_hf(nP) + 2 * nP * nP;
    const int* cB = ctl + (off == 1 ? TG_CTL_XSTEP2 : TG_CTL_XSTEP3);
    for (int p = 1; p + off < nP; ++p) {
        union { TgTask t; int4 v; } u;
        u.v = *reinterpret_cast<const int4*>(a.q[0] + p - 1);         // row p's final chunk: [k0, p), ordinal aux
        u.v.x = __builtin_amdgcn_readfirstlane(u.v.x); u.v.y = __builtin_amdgcn_readfirstlane(u.v.y);
        u.v.z = __builtin_amdgcn_readfirstlane(u.v.z); u.v.w = __builtin_amdgcn_readfirstlane(u.v.w);
        const TgTask d = u.t;
        const int64_t p0 = (int64_t)p * NB, q0 = p0 - NB, j0 = p0 + (int64_t)off * NB;
        long long ts0 = 0, ts1 = 0;
        if (a.trace && t == 0) ts0 = wall_clock64();
        if (!tg_wave_wait_ge(a, sq + p * nP + p + off, d.aux)) return;
        if (a.trace && t == 0) ts1 = wall_clock64();
        const __amdgpu_buffer_rsrc_t rs = tg_rsrc(S + p0 * Np + j0, Np);
        const int vo = (int)((g * Np + n) * 8);
        const int co = 64 * half + 16 * w;            // this wave's 16 columns of the tile
        d4 acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = tg_bload(rs, vo, (int)((((16 * r + 4 * q) * Np) + co) * 8));
        // block rows of the final chunk before row p-1 (block row 2 only: its first two chunks are merged): from global memory
        if (d.k0 < p - 1) {
            if (!tg_wave_wait_ge(a, sv + 2 * p, p - 1) || !tg_wave_wait_ge(a, sv + 2 * p + 1, p - 1) ||
                !tg_wave_wait_ge(a, sv + 2 * (p + off), p - 1) || !tg_wave_wait_ge(a, sv + 2 * (p + off) + 1, p - 1)) return;
            for (int kr = d.k0 * NB; kr < (p - 1) * NB; kr += 4) {
                const double* Ra = R + (int64_t)(kr + g) * Np + p0 + n;
                const double b0 = -ldg<true>(R + (int64_t)(kr + g) * Np + j0 + co + n);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(ldg<true>(Ra + 16 * r), b0, acc[r], 0, 0, 0);
            }
        }
        const double* Ag = R + q0 * Np + p0;          // tile (p-1, p): S1's
        const double* Bg = R + q0 * Np + j0;          // tile (p-1, p+off): S2's (S3's)
        TgFollow f;
        f.avail = 0; f.issued = 0; f.vmi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) f.mark[i] = 0;
        if (!tg_v_step<0, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
        if (!tg_v_step<1, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
        if (!tg_v_step<2, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
        if (!tg_v_step<3, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
        if (!tg_v_step<4, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
        if (!tg_v_step<5, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
        if (!tg_v_step<6, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
        if (!tg_v_step<7, DB>(a, cB, p, half, f, acc, lds, Ag, Bg, Np)) return;
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) tg_bstore(rs, vo, (int)((((16 * r + 4 * q) * Np) + co) * 8), acc[r][q]);
        tg_drain();
        __syncthreads();
        if (t == 0) {
            if (atomicAdd(hc + p * nP + p + off, 1) == 1) sti(sq + p * nP + p + off, d.aux + 1);     // the second half to arrive counts the chunk
            if (a.trace && off == 1 && half == 0) {         // slots 6, 7 of the block row it follows: V waiting from / its tile's earlier chunks in at, .. / stored at
                long long* o = a.trace + 4 * nP + 2 * 8 * (p - 1) + 12;
                o[0] = ts0; o[1] = ts1; o[2] = ts1; o[3] = wall_clock64();
            }
        }
    }
}
template <bool DB> __device__ TG_BODY void tg_role_v(int half) { tg_role_offupd<DB>(tg_kargs(), 1, half); }
template <bool DB> __device__ TG_BODY void tg_role_v2(int half) { tg_role_offupd<DB>(tg_kargs(), 2, half); }


template <bool DB>
__device__ TG_BODY void tg_do_upd(int k0, int k1, int I, int J) {
    const TgArgs& a = tg_kargs();
    // (arguments of a non-inlined function travel in VGPRs: tell the compiler they are wave-uniform)
    k0 = __builtin_amdgcn_readfirstlane(k0); k1 = __builtin_amdgcn_readfirstlane(k1);
    I = __builtin_amdgcn_readfirstlane(I); J = __builtin_amdgcn_readfirstlane(J);
    syrk_tile<true, 2, DB>(uni(a.R), uni(a.S), (int64_t)uni64((unsigned long long)a.Np), k0, k1, I, J, tg_buf);
}
__device__ TG_BODY bool tg_do_trsm(int p, int cb) {
    const TgArgs& a = tg_kargs();
    p = __builtin_amdgcn_readfirstlane(p); cb = __builtin_amdgcn_readfirstlane(cb);
    __builtin_amdgcn_s_setprio(3);
    const int* diag = uni(a.ctl) + TG_CTL_BASE;
    return panel_solve16_lds(a, uni(a.U), uni(a.S), uni(a.R), (int64_t)uni64((unsigned long long)a.Np), p, cb, tg_buf, diag);
}
// ---- TG_TRSMU: one link of a column's chain in ONE task ----------------------------------------------------------------
// Below the shadows' band every tile column walks down the block rows alone: the solve of tile (p, J) needs the last update
// of the same tile, which needs the solve of tile (p-1, J).  As two worker tasks (solve 20 us, one-block update 24 us: each
// loads its tile, stores it, publishes, is polled for) a link cost 44 us -- more than the 38 us per diagonal block the shadows
// allow -- and the columns fell behind the diagonal.  Here a link is one task per 64-column half: the solve (panel_solve16_lds's
// MFMAs in its order), then, with the solved rows still in registers as the B operand, the final chunk of the same half of tile
// (p+1, J):  S(p+1, J) -= R(p, p+1)^T R(p, J), R(p, p+1) staged whole in LDS (LDS-direct loads; it needs the two k-step images of
// the one-workgroup-per-CU launch).  Same FMAs per element as the tile engine (accumulators from S, k ascending 4 at a time).
// The half publishes itself (solved[], a flag for the same half's next link); the second half to arrive counts the tile's chunk.
__device__ TG_BODY bool tg_do_trsmu(int p, int J, int h, int k0, int ordn) {
    const TgArgs& a = tg_kargs();
    p = __builtin_amdgcn_readfirstlane(p); J = __builtin_amdgcn_readfirstlane(J); h = __builtin_amdgcn_readfirstlane(h);
    k0 = __builtin_amdgcn_readfirstlane(k0); ordn = __builtin_amdgcn_readfirstlane(ordn);
    __builtin_amdgcn_s_setprio(3);
    double* lds = tg_buf;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), g = lane >> 4, n = lane & 15;
    const int nP = __builtin_amdgcn_readfirstlane(a.nP), npad = tg_npad(nP);
    const int64_t Np = (int64_t)uni64((unsigned long long)a.Np);
    double* R = uni(a.R);
    double* S = uni(a.S);
    const double* U = uni(a.U);
    int* ctl = uni(a.ctl);
    int* dd = ctl + TG_CTL_BASE;
    int* sv = dd + 2 * npad;
    int* sq = sv + 2 * npad;
    int* hf = ctl + tg_ctl_hf(nP);
    int* hc = hf + 2 * nP * nP;
    const int64_t p0 = (int64_t)p * NB, i0 = p0 + NB, j0 = (int64_t)J * NB + 64 * h + 16 * w;
    const double* Rd = R + p0 * Np + p0;
    const double* Ud = U + p0 * Np + p0;
    // (0) the right-hand sides: final when this task was taken
    const __amdgpu_buffer_rsrc_t rx = tg_rsrc(S + p0 * Np + j0, Np), ro = tg_rsrc(R + p0 * Np + j0, Np);
    const int vo = (int)((g * Np + n) * 8);
    d4 X[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) X[r][q] = tg_bload(rx, vo, (int)(((16 * r + 4 * q) * Np) * 8));
    if (!tg_wave_wait_ge(a, dd + p, 1)) return false;
    // (1) the factor's diagonal block (36 upper 16-tiles, tile-major) and the eight 16 x 16 inverses: ONE round trip
    {
        const __amdgpu_buffer_rsrc_t rs = tg_rsrc(Rd, Np);
        u4v stage[18];
        const int piece = t & 127, row = piece >> 3, c2 = piece & 7;       // (as in panel_solve16_lds: one position register)
        const int dst0 = (t >> 7) * 256 + row * 16 + 2 * c2;
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            int r = 0;
            int idx = 2 * q + (t >> 7);
            while (idx >= 8 - r) { idx -= 8 - r; ++r; }
            const int c = r + idx;
            stage[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((16 * r + row) * Np + 16 * c + 2 * c2) * 8), 0, 16);
        }
        double ti[8][4];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) ti[jb][kk] = ldg<true>(Ud + (int64_t)(16 * jb + 4 * kk + g) * Np + 16 * jb + n);
#pragma unroll
        for (int q = 0; q < 18; ++q) *reinterpret_cast<u4v*>(lds + dst0 + 512 * q) = stage[q];
        __syncthreads();
        // (2) the substitution
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ti[jb][kk], X[jb][kk], x, 0, 0, 0);
            X[jb] = x;
            const d4 xn = -x;
#pragma unroll
            for (int i = jb + 1; i < 8; ++i) {
                const double* tl = lds + 256 * tri_index(jb, i) + g * 16 + n;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) X[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(tl[kk * 64], xn[kk], X[i], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) tg_bstore(ro, vo, (int)(((16 * r + 4 * q) * Np) * 8), X[r][q]);
    // (3) the same half of tile (p+1, J): every earlier chunk applied (long ago, as a rule), then R(p, p+1) whole into LDS
    if (!tg_wave_wait_ge(a, sq + (p + 1) * nP + J, ordn)) return false;
    const __amdgpu_buffer_rsrc_t rt = tg_rsrc(S + i0 * Np + j0, Np);
    d4 acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = tg_bload(rt, vo, (int)(((16 * r + 4 * q) * Np) * 8));
    // the solved rows are announced NOW, not with the update: the workers' chunk of tile (p+2, J) that ends with block row p
    // takes 24 us and the next link of this column wants it ~12 us after this one ends
    tg_drain();
    __syncthreads();                               // (every wave is also through with the diagonal block's image)
    if (t == 0) sti(sv + 2 * J + h, p + 1);
    if (k0 < p) {              // block rows of the chunk before row p (block row 2 only: merged first chunks): from global memory
        if (!tg_wave_wait_ge(a, sv + 2 * (p + 1), p) || !tg_wave_wait_ge(a, sv + 2 * (p + 1) + 1, p) ||
            !tg_wave_wait_ge(a, sv + 2 * J + h, p)) return false;
        for (int kr = k0 * NB; kr < p * NB; kr += 4) {
            const double* Ra = R + (int64_t)(kr + g) * Np + i0 + n;
            const double bq = -ldg<true>(R + (int64_t)(kr + g) * Np + j0 + n);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(ldg<true>(Ra + 16 * r), bq, acc[r], 0, 0, 0);
        }
    }
    if (!tg_wave_wait_ge(a, sv + 2 * (p + 1), p + 1) || !tg_wave_wait_ge(a, sv + 2 * (p + 1) + 1, p + 1)) return false;
    {
        const double* Ag = R + p0 * Np + i0;       // tile (p, p+1)
#pragma unroll
        for (int rr = 0; rr < 32; ++rr)
            __builtin_amdgcn_global_load_lds(Ag + (int64_t)(32 * w + rr) * Np + 2 * lane, (lds_ptr)(lds + (32 * w + rr) * PFP), 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (4) acc(i) -= R(p, p+1)[:, tile i]^T x  as  acc(i) += A_i (-x)
#pragma unroll
    for (int r = 0; r < 8; ++r) X[r] = -X[r];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        double af[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) af[i][kk] = lds[(16 * kt + 4 * kk + g) * PFP + 16 * i + n];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i][kk], X[kt][kk], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) tg_bstore(rt, vo, (int)(((16 * r + 4 * q) * Np) * 8), acc[r][q]);
    tg_drain();
    __syncthreads();
    if (t == 0) {
        sti(hf + ((p + 1) * nP + J) * 2 + h, 1);
        if (atomicAdd(hc + (p + 1) * nP + J, 1) == 1) sti(sq + (p + 1) * nP + J, ordn + 1);
    }
    return true;
}

// (the task goes to the workgroup's LDS slot, not through a reference to the caller's registers: that was scratch memory too)
__device__ TG_BODY int tg_take_call(int lane) {
    const TgArgs& a = tg_kargs();
    TgTask tk;
    tk.type = 0;
    const int c = tg_take(a, tk, lane, reinterpret_cast<TgHeld*>(tg_smem + 4));
    if (lane == 0) *reinterpret_cast<TgTask*>(tg_smem) = tk;
    return c;
}

// DB: the launch gives every workgroup two k-step images of LDS (one workgroup per CU): the workers' tile updates run the
// double-buffered k-loop.
template <bool DB>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_chol_tg(const TgArgs a) {
    TgTask* cur = reinterpret_cast<TgTask*>(tg_smem);           // 16 bytes
    int* code = reinterpret_cast<int*>(tg_smem + 2);             // [0] take result / role, [1] potrf's sflag
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) *reinterpret_cast<TgArgs*>(tg_smem + 16) = a;     // the role bodies' view of the arguments (tg_kargs)
    const int nP = a.nP, npad = tg_npad(nP);
    int* ctl = a.ctl;
    int* dd = ctl + TG_CTL_BASE;
    int* qd = dd + npad;
    int* sv = qd + npad;
    int* sq = sv + 2 * npad;
    if (t == 0) {
        // Roles in order of arrival.  The critical workgroups (role C and the shadows) keep their compute unit
        // to themselves: next to a worker's matrix phases the diagonal block took 60-80 us instead of 26 and the
        // critical solves twice their time (profiles/history/r04_chol_taskgraph.txt).  The second workgroup to start on a CU
        // looks up what the first one became and leaves at once if that is a critical role (a grid of two workgroups
        // per CU has no third one waiting to take the slot).
        reinterpret_cast<TgHeld*>(tg_smem + 4)->have = 0;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
        const int key = (int)((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf));
        int* cu_cnt = sq + nP * nP;
        int* cu_role = cu_cnt + TG_CU_KEYS;
        int r = -1;
        if (a.isolate && atomicAdd(cu_cnt + key, 1) > 0) {
            int first = 0;
            for (unsigned spins = 0; (first = ldi(cu_role + key)) == 0 && spins < 100000; ++spins) __builtin_amdgcn_s_sleep(1);
            if (first != 0 && first - 1 <= TG_NSHADOW) r = -2;          // leave
        }
        if (r != -2) {
            r = atomicAdd(ctl, 1);
            if (a.isolate) atomicCAS(cu_role + key, 0, r + 1);
        }
        code[0] = r;
    }
    __syncthreads();
    const int role = code[0];
    __syncthreads();
    if (role < 0) return;
    if (role == 0) {            // role C: the diagonal blocks, one after the other
        tg_role_diag();
        return;
    }
    if (role <= TG_NSHADOW) {        // the shadows of role C
        if (role == 1) tg_role_s1();
        else if (role == 2) tg_role_s2();
        else if (role == 3) tg_role_s3();
        else if (role == 4) tg_role_u();
        else if (role == 5) tg_role_u0();
        else if (role <= 7) tg_role_v<DB>(role - 6);
        else tg_role_v2<DB>(role - 8);
        return;
    }
    // the dispatcher keeps nothing in vector registers across a task (the callee would have to save it): the task is read back
    // from LDS, the trace sums (thread 0, trace runs only) live in LDS
    volatile long long* prof = reinterpret_cast<volatile long long*>(tg_smem + 8);      // [0..5] sums, [6] tprev, [7] ts
    if (t == 0) {
        for (int i = 0; i < 6; ++i) prof[i] = 0;
        prof[6] = a.trace ? wall_clock64() : 0;
    }
    const volatile TgTask* vc = cur;
    for (;;) {
        if (w == 0) {
            __builtin_amdgcn_s_setprio(0);
            const int c = tg_take_call(lane);
            if (lane == 0) {
                code[0] = c;
                // the single-buffer tile engine reads its operand panels with plain 16-byte loads: drop this CU's stale L1 lines
                // (the double-buffered one loads them sc1, past the L1: no fence, 1.7 us per task)
                if (!DB && c == 1 && vc->type == TG_UPD) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        if (code[0] != 1) break;
        const int type = __builtin_amdgcn_readfirstlane(vc->type);
        if (a.trace && t == 0) { const long long ts = wall_clock64(); prof[7] = ts; prof[1] += ts - prof[6]; }
        if (type == TG_UPD) {
            tg_do_upd<DB>(vc->k0, vc->k1, vc->I, vc->J);
        } else if (type == TG_TRSM) {
            if (!tg_do_trsm(vc->I, 2 * (vc->J - vc->I - 1) + vc->aux)) break;
        } else {
            if (!tg_do_trsmu(vc->I, vc->J, vc->aux, vc->k0, vc->rsv)) break;
        }
        tg_drain();
        __syncthreads();
        if (t == 0) {
            TgTask tk;
            tk.type = vc->type; tk.I = vc->I; tk.J = vc->J; tk.k0 = vc->k0; tk.k1 = vc->k1; tk.ord = vc->ord; tk.aux = vc->aux; tk.rsv = vc->rsv;
            long long te = 0;
            if (a.trace) {
                te = wall_clock64();
                const long long ts = prof[7];
                if (a.tasklog && role < TG_LOG_WGS && prof[0] < TG_LOG_CAP) {
                    long long* rec = a.tasklog + ((long long)role * TG_LOG_CAP + prof[0]) * 4;
                    union { TgTask t; long long w[2]; } u;
                    u.t = tk;
                    rec[0] = u.w[0]; rec[1] = u.w[1]; rec[2] = ts; rec[3] = te;
                }
                prof[0] += 1;
                prof[tk.type == TG_UPD ? 2 : 3] += te - ts;
                if (tk.type == TG_UPD) prof[5] += tk.k1 - tk.k0;
            }
            if (tk.type == TG_UPD) sti(sq + tk.I * nP + tk.J, tk.ord + 1);
            else if (tk.type == TG_TRSM) sti(sv + 2 * tk.J + tk.aux, tk.I + 1);          // (a fused link has published itself)
            if (a.trace) { const long long tp = wall_clock64(); prof[6] = tp; prof[4] += tp - te; }
        }
    }
    if (a.trace && t == 0 && role < 1024) {
        long long* o = a.trace + 20 * (long long)nP + 8 * role;
        for (int i = 0; i < 6; ++i) o[i] = prof[i];
        o[6] = 2;
        o[7] = wall_clock64();
    }
}

// ---- host: the task lists -----------------------------------------------------------------------------------------

// chunk boundaries of block row I: k = I - d for the distances d = 0, c1, c1 + c2, ... (chunk sizes counted back from
// the pivot, the last size repeated), plus 0
static std::vector<int> tg_boundaries(int I, const std::vector<int>& sizes) {
    std::vector<int> b;
    int d = 0;
    size_t i = 0;
    b.push_back(I);
    while (true) {
        d += sizes[std::min(i, sizes.size() - 1)];
        ++i;
        if (I - d <= 0) break;
        b.push_back(I - d);
    }
    b.push_back(0);
    std::reverse(b.begin(), b.end());                 // ascending: 0 = b[0] < ... < b[nb] = I
    b.erase(std::unique(b.begin(), b.end()), b.end());
    return b;
}

struct TgTables { std::vector<TgTask> q[2]; };

// The lists, in generation order = a topological order of the graph.  List 0 is not drawn from: it holds one descriptor per
// block row p for the shadows -- ord: the chunks of every tile of row p (what a solve of that row waits for); [k0, k1) / aux:
// the final chunk of row p+1 and its ordinal; rsv: the first block row of the chunk before it on the diagonal tile (p+1, p+1)
// (role U0's), or -1.  List 1, the workers': per step p (the block row that POTRF(p) releases) first the links of the column
// chains right of the shadows' band (J >= p+4; fused: solve + final chunk of the tile below per 64-column half; else the two
// solve halves), then the chunks that end at boundary p + 1, row by row, nearest the pivot first -- without those that belong
// to the shadows (the final chunks of tiles (I, I .. I+2), the diagonal tile's chunk before its final one) and, with fused
// links, without any final chunk.  (Column-major inside the step -- every update right behind the last solve it needs -- puts
// long far chunks in front of later solves: 7.8 against 5.5 ms at N = 8192, removed.)
static void tg_build(int nP, int chunk_code, TgTables& out, bool fuse) {
    std::vector<int> sizes;
    {
        std::vector<int> dg;
        for (int c = chunk_code; c > 0; c /= 10) dg.push_back(c % 10);
        std::reverse(dg.begin(), dg.end());
        for (int v : dg) if (v > 0) sizes.push_back(v == 9 ? 16 : v);       // (digit 9 stands for a chunk of 16 blocks)
        if (sizes.empty() || sizes[0] != 1) sizes.insert(sizes.begin(), 1);      // the final chunk is one block
    }
    std::vector<std::vector<int>> bnd(nP);
    std::vector<std::vector<int>> ends(nP + 1);        // ends[b] = rows with a chunk ending at boundary b
    for (int I = 0; I < nP; ++I) {
        bnd[I] = tg_boundaries(I, sizes);
        for (size_t j = 1; j < bnd[I].size(); ++j) ends[bnd[I][j]].push_back(I);
    }
    auto push = [&](int q, int type, int I, int J, int k0, int k1, int ord, int aux) {
        TgTask t;
        t.type = (int16_t)type; t.I = (int16_t)I; t.J = (int16_t)J; t.k0 = (int16_t)k0; t.k1 = (int16_t)k1;
        t.ord = (int16_t)ord; t.aux = (int16_t)aux; t.rsv = 0;
        out.q[q].push_back(t);
    };
    // the chunk before the final one of row I, if the final one is the single block row I-1: its first block row, else -1
    auto u0_start = [&](int I) {
        const std::vector<int>& b = bnd[I];
        const int nb = (int)b.size() - 1;
        return (nb >= 2 && b[nb - 1] == I - 1) ? b[nb - 2] : -1;
    };
    for (int q = 0; q < 2; ++q) out.q[q].clear();
    for (int p = 0; p < nP; ++p) {
        const int nch = (int)bnd[p].size() - 1;        // chunks of every tile of row p (0 for row 0)
        for (int J = p + 1; J < nP; ++J) {
            if (J == p + 1) {                          // one descriptor per block row, completed below (k0, k1, aux)
                push(0, TG_SHADOW, p, J, 0, 0, (p == 0) ? 0 : nch, 0);
                // rsv: where the chunk BEFORE the final one of the diagonal tile (p+1, p+1) starts (role U0's), -1 without one
                out.q[0].back().rsv = (int16_t)u0_start(p + 1);
                continue;
            }
            if (J <= p + 3) continue;                  // roles S2's and S3's
            if (fuse) {                                // one link of column J's chain: the solve and the final chunk of tile (p+1, J), per half
                const std::vector<int>& b = bnd[p + 1];
                const int nb = (int)b.size() - 1;      // the final chunk of row p+1 is [b[nb-1], p+1), its ordinal nb-1
                for (int h = 0; h < 2; ++h) {
                    push(1, TG_TRSMU, p, J, b[nb - 1], p + 1, (p == 0) ? 0 : nch, h);
                    out.q[1].back().rsv = (int16_t)(nb - 1);
                }
                continue;
            }
            for (int h = 0; h < 2; ++h) push(1, TG_TRSM, p, J, 0, 0, (p == 0) ? 0 : nch, h);
        }
        for (int I : ends[p + 1]) {                    // rows ascending: nearest the pivot first
            if (I == 0) continue;
            size_t j = 1;
            while (bnd[I][j] != p + 1) ++j;
            const int k0 = bnd[I][j - 1], k1 = p + 1, ord = (int)j - 1;
            for (int J = I; J < nP; ++J) {
                if (I == k1 && J == I) {                   // role U's: the final chunk of the diagonal tile
                    TgTask& d = out.q[0][(size_t)p];
                    d.k0 = (int16_t)k0; d.k1 = (int16_t)k1; d.aux = (int16_t)ord;
                } else if (I == k1 && (J <= I + 2 || fuse)) {
                    continue;                              // roles V's and V2's: the final chunks of tiles (I, I+1), (I, I+2); beyond: the fused links'
                } else if (J == I && k1 == I - 1 && u0_start(I) == k0) {
                    continue;                              // role U0's: the chunk before the final one of the diagonal tile
                } else {
                    push(1, TG_UPD, I, J, k0, k1, ord, 0);
                }
            }
        }
    }
}

// host-only view of the lists for the CPU tests (gpx_chol_tasks): 8 int16 per task, the two lists back to back
int64_t tg_tasks_copy(int nP, int chunks, int16_t* out, int64_t cap, int64_t* counts) {
    if (nP < 1 || nP > 2047) return -1;
    TgTables tb;
    if (chunks <= 0) chunks = tg_default_chunks(nP);
    tg_build(nP, chunks, tb, true);
    int64_t tot = 0;
    for (int q = 0; q < 2; ++q) { counts[q] = (int64_t)tb.q[q].size(); tot += counts[q]; }
    if (out && cap >= tot) {
        int64_t o = 0;
        for (int q = 0; q < 2; ++q) {
            if (!tb.q[q].empty()) std::memcpy(out + 8 * o, tb.q[q].data(), tb.q[q].size() * sizeof(TgTask));
            o += counts[q];
        }
    }
    return tot;
}

// The gate of the inversion's leading part (launch_trtri_ahead): ONE wave on a side stream that leaves when block rows
// 0 .. top - 1 of R are final -- the diagonal block top - 1 is factored and block row top - 1 is solved in every block column to
// its right -- i.e. when everything the kernels queued behind it read has been stored (write-through) by the running
// factorisation.  An abort ends it at once (the caller discards what follows); its own time-out raises the abort word, so that a
// gate that gave up can never let kernels through onto a factor that is not there yet.
__global__ __launch_bounds__(64) void k_tg_gate(int* ctl, int* dflag, int nP, int top, long long tmo) {
    const int npad = tg_npad(nP);
    const int* dd = ctl + TG_CTL_BASE;
    const int* sv = dd + 2 * npad;
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        if (ldi(ctl + TG_CTL_ABORT) != 0) return;
        bool ok = ldi(dd + top - 1) != 0;
        for (int j = 2 * top + lane; ok && j < 2 * nP; j += 64) ok = ldi(sv + j) >= top;
        if (__all(ok)) return;
        __builtin_amdgcn_s_sleep(64);
        if ((spins & 63) == 63 && wall_clock64() - t0 > tmo) {
            if (lane == 0) { sti(ctl + TG_CTL_ABORT, 2); sti(dflag + 1, 2); }
            return;
        }
    }
}


struct TgCache {                 // per handle (gpx_handle::tg): device copies of the tables and the control block
    int nP = 0, chunks = 0, fuse = -1;
    TgTask* dq = nullptr;
    int64_t cap_q = 0;
    int n[2] = {0, 0};
    int64_t off[2] = {0, 0};
    int* dctl = nullptr;
    int64_t cap_ctl = 0;
    long long* dtrace = nullptr;
    int64_t cap_trace = 0;
    int max_resident = 0, max_resident_db = 0;
};

void tg_free(gpx_handle* h) {
    TgCache* c = static_cast<TgCache*>(h->tg);
    if (!c) return;
    if (c->dq) (void)hipFree(c->dq);
    if (c->dctl) (void)hipFree(c->dctl);
    if (c->dtrace) (void)hipFree(c->dtrace);
    delete c;
    h->tg = nullptr;
}

// S -> R (and the 16 x 16 inverses in the diagonal tiles of T / U) by the persistent kernel.  Returns false when the
// launch could not be prepared (the caller runs the stream schedule instead).  After the stream has drained,
// tg_abort_code() tells whether a spin gave up (2): the caller then rebuilds the Gram matrix and re-runs the stream schedule.
bool launch_cholesky_tg(gpx_handle* h) {
    const int64_t Np = h->Np;
    const int nP = (int)(Np / NB);
    if (nP > 2047) return false;                       // int16 task fields
    if (!h->tg) h->tg = new TgCache();
    TgCache* c = static_cast<TgCache*>(h->tg);
    hipStream_t s = h->stream;
    const int chunks = h->tg_chunks > 0 ? h->tg_chunks : tg_default_chunks(nP);
    static_assert(3 * TG_SH_PAN + 3 * 256 <= GEMM_LDS_F64 && 4 * TG_SH_PAN <= GEMM_LDS_F64 && 8 * TG_SH_PAN <= 2 * GEMM_LDS_F64, "the shadows' LDS rings fit the tile engine's buffer(s)");
    // (decided here because the lists depend on it) one workgroup per CU with two k-step images of LDS: see below
    const bool db = (h->tg_db < 0) ? (nP <= h->tg_db_max) : (h->tg_db != 0);
    const bool fuse = db && h->tg_fuse != 0;       // a fused link stages a whole 128 x 128 tile in LDS
    static_assert(NB * PFP <= 2 * GEMM_LDS_F64, "tile (p, p+1) fits the two k-step images");
    if (c->nP != nP || c->chunks != chunks || c->fuse != (int)fuse || !c->dq) {
        TgTables tb;
        tg_build(nP, chunks, tb, fuse);
        const int64_t tot = (int64_t)tb.q[0].size() + (int64_t)tb.q[1].size() + 2;
        if (tot > c->cap_q) {
            if (c->dq) (void)hipFree(c->dq);
            c->dq = nullptr; c->cap_q = 0;
            if (hipMalloc((void**)&c->dq, (size_t)tot * sizeof(TgTask)) != hipSuccess) { (void)hipGetLastError(); return false; }
            c->cap_q = tot;
        }
        int64_t o = 0;
        (void)hipStreamSynchronize(s);                 // (an earlier launch may still be reading the old tables)
        for (int q = 0; q < 2; ++q) {
            c->off[q] = o;
            c->n[q] = (int)tb.q[q].size();
            if (c->n[q] > 0 &&
                (hipMemcpyAsync(c->dq + o, tb.q[q].data(), tb.q[q].size() * sizeof(TgTask), hipMemcpyHostToDevice, s) != hipSuccess ||
                 hipStreamSynchronize(s) != hipSuccess)) {
                (void)hipGetLastError();
                c->nP = 0;
                return false;
            }
            o += c->n[q] + 1;
        }
        c->nP = nP; c->chunks = chunks; c->fuse = (int)fuse;
    }
    const int64_t nctl = tg_ctl_ints(nP);
    if (nctl > c->cap_ctl) {
        if (c->dctl) (void)hipFree(c->dctl);
        c->dctl = nullptr; c->cap_ctl = 0;
        if (hipMalloc((void**)&c->dctl, (size_t)nctl * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return false; }
        c->cap_ctl = nctl;
    }
    const int nside = TG_NSHADOW;
    const int64_t nlog = (h->tg_trace >= 2) ? (int64_t)TG_LOG_WGS * TG_LOG_CAP * 4 : 0;
    const int64_t ntrace = 20 * (int64_t)nP + 8 * 1024 + 16 + nlog;
    if (h->tg_trace && ntrace > c->cap_trace) {
        if (c->dtrace) (void)hipFree(c->dtrace);
        c->dtrace = nullptr; c->cap_trace = 0;
        if (hipMalloc((void**)&c->dtrace, (size_t)ntrace * 8) != hipSuccess) { (void)hipGetLastError(); return false; }
        c->cap_trace = ntrace;
    }
    // DB: every workgroup gets two k-step images of LDS (one workgroup per CU by its LDS alone) and the workers run the
    // double-buffered k-loop -- the choice of the sizes that ran one workgroup per CU anyway (up to 72 blocks: latency-bound;
    // option chol_tg_db: -1 auto, 0 / 1 force)
    const size_t lds_bytes = (size_t)(TG_CTL_F64 + (db ? 2 : 1) * GEMM_LDS_F64) * sizeof(double);
    int& max_res = db ? c->max_resident_db : c->max_resident;
    if (max_res == 0) {
        int nb = 0;
        hipDeviceProp_t prop;
        const void* fn = db ? (const void*)k_chol_tg<true> : (const void*)k_chol_tg<false>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            (db ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_chol_tg<true>, GEMM_THREADS, lds_bytes)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_chol_tg<false>, GEMM_THREADS, lds_bytes)) != hipSuccess ||
            nb < 1 || hipGetDeviceProperties(&prop, h->device) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        max_res = nb * prop.multiProcessorCount;
    }
    (void)hipMemsetAsync(h->dflag, 0, 2 * sizeof(int), s);      // [0] failing pivot + 1, [1] 2 = a spin gave up (what tg_abort_code reads, next to the flag)
    (void)hipMemsetAsync(c->dctl, 0, (size_t)nctl * sizeof(int), s);
    if (h->tg_trace) (void)hipMemsetAsync(c->dtrace, 0, (size_t)ntrace * 8, s);
    hipLaunchKernelGGL(k_zero_diag_lower, dim3((unsigned)nP), dim3(256), 0, s, h->dR, Np);
    // one worker per tile of the matrix can be busy at most (plus the solves of a block row)
    int64_t want = 1 + nside + (int64_t)nP * (nP + 1) / 2 + 2 * nP;
    // up to ~72 blocks the factorisation is bound by the latency of its dependent tasks, not by throughput: ONE workgroup
    // per compute unit runs every task ~1.6x faster (N = 4096: 1.91 against 2.23 ms, N = 8192: 5.40 against 5.98; from
    // N = 12288 on two per CU win: 13.6 against 14.4 ms)
    if (!db && nP <= 72) want = std::min<int64_t>(want, max_res / 2 + 8);
    if (h->tg_grid > 0) want = h->tg_grid;
    const int grid = (int)std::max<int64_t>(2 + nside, std::min<int64_t>(want, max_res));
    const bool grid_is_full = !db && grid >= max_res;      // two workgroups per CU everywhere: isolation has a meaning
    TgArgs a;
    std::memset(&a, 0, sizeof a);
    a.S = h->dS; a.R = h->dR; a.T = h->dT; a.U = h->dU;
    a.Np = Np; a.nP = nP; a.dflag = h->dflag; a.ctl = c->dctl;
    for (int q = 0; q < 2; ++q) { a.q[q] = c->dq + c->off[q]; a.n[q] = c->n[q]; }
    // -1 = by size: where two workgroups per CU are the default (more than 112 blocks) the diagonal waits for its inputs anyway
    // (150 us per block at N = 16384) and ten more CUs of workers are worth more than a faster critical path: 24.8 -> 24.45 ms at
    // N = 16384, 46.0 -> 45.3 at 20480; below (chol_tg_db = 0 forced) the critical workgroups keep their CUs
    const bool iso = h->tg_isolate < 0 ? nP <= 112 : h->tg_isolate != 0;
    a.isolate = (iso && grid_is_full) ? 1 : 0;
    a.nap = h->tg_nap > 0 ? h->tg_nap : 16;
    a.trace = h->tg_trace ? c->dtrace : nullptr;
    a.tasklog = nlog ? c->dtrace + 20 * (int64_t)nP + 8 * 1024 + 16 : nullptr;
    a.tmo = (long long)(h->tg_tmo_ms > 0 ? h->tg_tmo_ms : 2000) * 100000LL;
    // the inversion's leading part rides on the side stream, behind a gate (one workgroup per CU only: a tile-engine workgroup
    // of another kernel cannot move in next to the critical roles -- 144 + 72 KB of LDS do not fit a CU)
    const int top = trtri_top(nP);
    const bool ahead = h->want_ahead && h->trtri_ahead != 0 && db && nP >= h->trtri_ahead_min && h->trtri_left == 0 && h->stream2 &&
                       h->dT != nullptr && hipEventRecord(h->ev_far, s) == hipSuccess;
    if (db) hipLaunchKernelGGL(k_chol_tg<true>, dim3((unsigned)grid), dim3(GEMM_THREADS), lds_bytes, s, a);
    else hipLaunchKernelGGL(k_chol_tg<false>, dim3((unsigned)grid), dim3(GEMM_THREADS), lds_bytes, s, a);
    h->diag_inv_pending = true;
    h->tg_launched = true;
    h->ahead_top = 0;
    if (ahead) {
        (void)hipStreamWaitEvent(h->stream2, h->ev_far, 0);        // the control block is zeroed
        hipLaunchKernelGGL(k_tg_gate, dim3(1), dim3(64), 0, h->stream2, c->dctl, h->dflag, nP, top, a.tmo);
        launch_trtri_ahead(h, h->stream2, top);
        (void)hipEventRecord(h->ev_rest, h->stream2);
        h->ahead_top = top;
    }
    return true;
}

// the abort word of the last launch (valid once the stream has drained): 0 ok, 1 not positive definite, 2 a spin gave up
int tg_abort_code(gpx_handle* h) {
    TgCache* c = static_cast<TgCache*>(h->tg);
    if (!c || !c->dctl) return 0;
    int v = 0;
    // (never the NULL stream: see gpx_create)
    if (hipMemcpyAsync(&v, c->dctl + TG_CTL_ABORT, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { (void)hipGetLastError(); return 2; }
    return v;
}

// diagnostic (option "chol_tg_trace"): the stamps of the last launch, 100 MHz ticks: out[4 p + {0, 1, 2}] = role C started
// waiting for / started / finished diagonal block p; then per critical task (5 per block row) start / end
int64_t tg_trace_copy(gpx_handle* h, long long* out, int64_t n) {
    TgCache* c = static_cast<TgCache*>(h->tg);
    if (!c || !c->dtrace || !h->tg_trace) return 0;
    const int64_t have = std::min<int64_t>(n, std::min<int64_t>(c->cap_trace, 20 * (int64_t)c->nP + 8 * 1024 + 16 + (h->tg_trace >= 2 ? (int64_t)TG_LOG_WGS * TG_LOG_CAP * 4 : 0)));
    if (have <= 0) return 0;
    if (hipMemcpyAsync(out, c->dtrace, (size_t)have * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return have;
}

}  // namespace gpx
