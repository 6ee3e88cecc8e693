// sweep_probe.hip -- stand-alone A/B bench + phase instrumentation of the dominant kernel (k_sweep_trmm, paired super-tile map,
// k-step 32 through a single LDS buffer), VERDICT round 4 item 1: where do the matrix pipe's idle cycles go?
//
// The kernel body is the library's (gemm_core.h engine, the same tile map); this file adds, behind template flags,
//   STAMP   s_memtime stamps of wave 0 at the four phase boundaries of every k-step (top of step, end of the matrix phase,
//           after the first barrier, after LDS write + second barrier) and the workgroup's HW_ID / XCC_ID
//   DEPH    de-phasing of the two workgroups that share a compute unit: the second arrival (per-CU arrival counter keyed
//           by HW_ID / XCC_ID) pauses `dsleep` x 64 clocks before its first k-step
//   PRIOA   asymmetric wave priority: odd arrivals run their matrix phase at priority 2, even ones at 1
// Usage: sweep_probe.bin <mode> [dsleep=64] [reps=5] [N=8192] [cols=65536] [dump_prefix]
//   bit 9: priority 2 during the global-load issue, bit 10: global loads issued after the first MFMA group, bit 11: priority 1 from the top of the step, bit 12: no s_setprio at all
//   mode bit 0: STAMP, bit 1: DEPH, bit 2: PRIOA, bit 4: no LDS writes, bit 5: no global loads, bit 6 / 7: LDS writes paced by s_sleep 1 / 2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../pybo_amd/csrc/gemm_core.h"

using namespace gpx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int MAXSTEP = 272, NSLOT = 12;     // k-steps of a tile pair at N = 8192: (64 + 1) * 128 / 32 = 260

struct ProbeArgs {
    unsigned* stamps;    // [nblk][MAXSTEP][NSLOT]  low words of s_memtime: 0 top of step, 1..8 after the k-th group of 16 MFMAs, 9 after barrier 1, 10 after LDS write + barrier 2
    unsigned* meta;      // [nblk][8]: hw_id, xcc, arrival, t_start lo, t_start hi, t_end lo, t_end hi, nsteps
    int* cu_cnt;         // [4096] arrivals per compute unit
    int dsleep;
    unsigned* whw;       // [nblk][4] HW_ID of every wave
    long long* rt;       // [nblk][2] s_memrealtime (100 MHz) at start / end: the frequency of the s_memtime counter
};

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

template <int MODE>
__device__ __forceinline__ void kloop(d4 (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                      const double* __restrict__ B, int64_t ldb, int k_lo, int k_hi, double* smem,
                                      unsigned* st, int& step, int prio) {
    constexpr bool STAMP = MODE & 1;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    double* As = smem;
    double* Bs = smem + BK32 * LDT;
    const int lrow = w;
    const int lcol = lane * 2;
    d2 ra[8], rb[8];
    const int nk = (k_hi - k_lo) / BK32;
    if (nk <= 0) return;
    const double* Ap = A + (int64_t)(k_lo + lrow) * lda + lcol;
    const double* Bp = B + (int64_t)(k_lo + lrow) * ldb + lcol;
    // bit 13: buffer loads -- the step's base in SGPRs (bumped by SALU), the row of load p as an SGPR offset, the thread's
    // position as ONE constant VGPR offset per operand: no VALU instruction in the load issue
    const char* Abase = reinterpret_cast<const char*>(A + (int64_t)k_lo * lda);
    const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)k_lo * ldb);
    const int voA = (int)(((int64_t)lrow * lda + lcol) * 8), voB = (int)(((int64_t)lrow * ldb + lcol) * 8);
    const int soA = (int)(4 * lda * 8), soB = (int)(4 * ldb * 8);
    auto gload = [&]() {
        if (MODE & 8192) {
            __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, -1, 0x00020000);
            __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bbase, 0, -1, 0x00020000);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                ra[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rA, voA, p * soA, 0));
                rb[p] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rB, voB, p * soB, 0));
            }
            Abase += (int64_t)BK32 * lda * 8;
            Bbase += (int64_t)BK32 * ldb * 8;
            return;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            ra[p] = *reinterpret_cast<const d2*>(Ap + (int64_t)(4 * p) * lda);
            rb[p] = *reinterpret_cast<const d2*>(Bp + (int64_t)(4 * p) * ldb);
        }
        Ap += (int64_t)BK32 * lda;
        Bp += (int64_t)BK32 * ldb;
    };
    auto swrite = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *reinterpret_cast<d2*>(As + (lrow + 4 * p) * LDT + lcol) = ra[p];
            *reinterpret_cast<d2*>(Bs + (lrow + 4 * p) * LDT + lcol) = rb[p];
            if (MODE & 64) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); }
            if (MODE & 128) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(2); __builtin_amdgcn_sched_barrier(0); }
        }
    };
    const int fr = lane & 15, fk = lane >> 4;
    typedef __attribute__((address_space(3))) double lds_double;
    const lds_double* asb[8];
    const lds_double* bsb[8];
    if (MODE & (1 << 21)) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            unsigned oa = (unsigned)(size_t)(lds_double*)(As + wm * 64 + fr + (kk * 4 + fk) * LDT);
            unsigned ob = (unsigned)(size_t)(lds_double*)(Bs + wn * 64 + fr + (kk * 4 + fk) * LDT);
            asm volatile("" : "+v"(oa));
            asm volatile("" : "+v"(ob));
            asb[kk] = (const lds_double*)(size_t)oa;
            bsb[kk] = (const lds_double*)(size_t)ob;
        }
    }
    gload();
    swrite();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        unsigned tk[9];
        if (STAMP) tk[0] = (unsigned)now();
        if (MODE & 512) __builtin_amdgcn_s_setprio(2);          // the loads' address arithmetic ahead of the partner's MFMA stream
        if (MODE & 2048) __builtin_amdgcn_s_setprio(1);
        if (kt + 1 < nk && !(MODE & 32) && !(MODE & 1024)) gload();
        const double* as = As + wm * 64 + fr;
        const double* bs = Bs + wn * 64 + fr;
        if (MODE & 4) { if (prio == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
        else if (!(MODE & 4096)) __builtin_amdgcn_s_setprio(1);
        if (MODE & 16384) {
            // fragment reads paired over TWO k-groups per LDS instruction (ds_read2st64_b64: rows 4 apart are 9 x 512 B apart),
            // one base register per 16-wide fragment: no address arithmetic inside the matrix phase
            const double* asi[4];
            const double* bsi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asi[i] = as + fk * LDT + i * 16;
                bsi[i] = bs + fk * LDT + i * 16;
            }
#pragma unroll
            for (int k2 = 0; k2 < BK32 / 8; ++k2) {
                double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a0[i] = asi[i][(2 * k2) * 4 * LDT];
                    a1[i] = asi[i][(2 * k2 + 1) * 4 * LDT];
                    b0[i] = bsi[i][(2 * k2) * 4 * LDT];
                    b1[i] = bsi[i][(2 * k2 + 1) * 4 * LDT];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
            }
        } else if ((MODE & (1 << 22)) && (MODE & (1 << 21))) {
            // bits 21 + 22: the double-buffered fragments read through one opaque LDS base register per k-group and operand
            double a[2][4], b[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[0][i] = asb[0][i * 16];
                b[0][i] = bsb[0][i * 16];
            }
#pragma unroll
            for (int kk = 0; kk < BK32 / 4; ++kk) {
                if ((MODE & 32768) && kk == 4) __builtin_amdgcn_s_setprio(2);
                if (kk + 1 < BK32 / 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[(kk + 1) & 1][i] = asb[kk + 1][i * 16];
                        b[(kk + 1) & 1][i] = bsb[kk + 1][i * 16];
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        } else if (MODE & (1 << 23)) {
            // bit 23: fragments two groups ahead (three register sets)
            double a[3][4], b[3][4];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[g][i] = as[(g * 4 + fk) * LDT + i * 16];
                    b[g][i] = bs[(g * 4 + fk) * LDT + i * 16];
                }
#pragma unroll
            for (int kk = 0; kk < BK32 / 4; ++kk) {
                if ((MODE & 32768) && kk == 4) __builtin_amdgcn_s_setprio(2);
                if (kk + 2 < BK32 / 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[(kk + 2) % 3][i] = as[((kk + 2) * 4 + fk) * LDT + i * 16];
                        b[(kk + 2) % 3][i] = bs[((kk + 2) * 4 + fk) * LDT + i * 16];
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk % 3][i], b[kk % 3][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        } else if (MODE & (1 << 22)) {
            // bit 22: fragments of group kk+1 requested before the MFMAs of group kk (explicit double buffer in registers)
            double a[2][4], b[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[0][i] = as[fk * LDT + i * 16];
                b[0][i] = bs[fk * LDT + i * 16];
            }
#pragma unroll
            for (int kk = 0; kk < BK32 / 4; ++kk) {
                if ((MODE & 32768) && kk == 4) __builtin_amdgcn_s_setprio(2);
                if (kk + 1 < BK32 / 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[(kk + 1) & 1][i] = as[((kk + 1) * 4 + fk) * LDT + i * 16];
                        b[(kk + 1) & 1][i] = bs[((kk + 1) * 4 + fk) * LDT + i * 16];
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // the DS reads first
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);    // then the 16 MFMAs
            }
        } else if (MODE & (1 << 21)) {
            // bit 21: one LDS base register per k-group and operand, made opaque to the compiler (so that it keeps them in
            // registers instead of re-deriving them with a VALU add inside the matrix phase)
#pragma unroll
            for (int kk = 0; kk < BK32 / 4; ++kk) {
                if ((MODE & 32768) && kk == 4) __builtin_amdgcn_s_setprio(2);
                const lds_double* ak = asb[kk];
                const lds_double* bk = bsb[kk];
                double a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = ak[i * 16];
                    b[i] = bk[i * 16];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int kk = 0; kk < BK32 / 4; ++kk) {
            // bit 15: the wave that is further into its matrix phase outranks the other one (anti-phase becomes the attractor)
            if ((MODE & 32768) && kk == (((MODE >> 17) & 7) ? ((MODE >> 17) & 7) : 4)) __builtin_amdgcn_s_setprio(2);
            if ((MODE & 65536) && kk == 2) __builtin_amdgcn_s_setprio(2);
            if ((MODE & 65536) && kk == 5) __builtin_amdgcn_s_setprio(3);
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
            if ((MODE & 1024) && kk == 0 && kt + 1 < nk) gload();       // loads issued from inside the wave's own MFMA stream
            if (STAMP) tk[1 + kk] = (unsigned)now();
        }
        if (MODE & (1 << 20)) __builtin_amdgcn_s_setprio(3); else if (!(MODE & 4096)) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (STAMP && lane == 0) {
            st[step * NSLOT + 9] = (unsigned)now();
#pragma unroll
            for (int i = 0; i < 9; ++i) st[step * NSLOT + i] = tk[i];
        }
        if (kt + 1 < nk) {
            if (!(MODE & 16)) swrite();
            __syncthreads();
        }
        if (STAMP && lane == 0) st[step * NSLOT + 10] = (unsigned)now();
        ++step;
    }
}

template <int MODE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_probe(const double* __restrict__ U, int64_t Np,
                                                           const double* __restrict__ Ks, int64_t ldk, int NT,
                                                           const double* __restrict__ avec, double* __restrict__ Qp,
                                                           double* __restrict__ Pp, int64_t ldp, int sm, ProbeArgs pa) {
    __shared__ __attribute__((aligned(16))) double smem[GEMM_LDS_F64];
    __shared__ int s_arr;
    __shared__ double s_extra[(MODE & 256) ? 3072 : 1];     // bit 8: 24 KB more LDS -> ONE workgroup per compute unit
    if ((MODE & 256) && Np < 0) s_extra[threadIdx.x] = 1.0, Qp[0] = s_extra[threadIdx.x ^ 1];
    const int nP = (int)(Np / TB);
    int mt, nt, mt2 = -1;
    {
        const int b = blockIdx.x;
        const int x = b & 7, q = b >> 3;
        const int SN = 64 / sm;
        const int per = (NT + 7) / 8;
        const int hper = (per + SN - 1) / SN;
        const int s = q >> 6, r = q & 63;
        const int G = s / hper, H = s - G * hper;
        const int i = G * sm + r / SN;
        const int ln = H * SN + (r - (r / SN) * SN);
        nt = x * per + ln;
        mt = nP - 1 - i;
        if (ln >= per || nt >= NT || i > mt) return;
        if (i < mt) mt2 = i;
    }
    int arrival = 0;
    unsigned long long tstart = 0;
    if (MODE & (1 | 2 | 4 | 8)) {
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
            const int key = (int)((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf));
            const int arr = atomicAdd(pa.cu_cnt + key, 1);
            s_arr = arr;
            if (MODE & 1) {
                unsigned* m = pa.meta + (size_t)blockIdx.x * 8;
                tstart = now();
                m[0] = hw; m[1] = xcc; m[2] = (unsigned)arr; m[3] = (unsigned)tstart; m[4] = (unsigned)(tstart >> 32);
                pa.rt[(size_t)blockIdx.x * 2] = wall_clock64();
            }
        }
        __syncthreads();
        arrival = s_arr;
        if ((MODE & 2) && (arrival & 1)) {
            for (int i = 0; i < pa.dsleep; i += 64) __builtin_amdgcn_s_sleep(64);
        }
    }
    unsigned* st = (MODE & 1) ? pa.stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * MAXSTEP * NSLOT : nullptr;
    if ((MODE & 1) && (threadIdx.x & 63) == 0) pa.whw[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    int step = 0;
    const int prio = (arrival & 1) ? 2 : 1;
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
        kloop<MODE>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, smem, st, step, prio);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int wm = w >> 1, wn = w & 1;
        double av[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) av[i][r] = avec[m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * r];
        double qs[4], ps[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double q = 0.0, p = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = acc[i][j][r];
                    q = fma(v, v, q);
                    p = fma(v, av[i][r], p);
                }
            q += __shfl_xor(q, 16);
            p += __shfl_xor(p, 16);
            q += __shfl_xor(q, 32);
            p += __shfl_xor(p, 32);
            qs[j] = q;
            ps[j] = p;
        }
        double* red = smem;
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
            Qp[(int64_t)mt * ldp + n0 + c] = red[c * 2] + red[(TB + c) * 2];
            Pp[(int64_t)mt * ldp + n0 + c] = red[c * 2 + 1] + red[(TB + c) * 2 + 1];
        }
    }
    if ((MODE & 1) && threadIdx.x == 0) {
        unsigned* m = pa.meta + (size_t)blockIdx.x * 8;
        const unsigned long long te = now();
        m[5] = (unsigned)te; m[6] = (unsigned)(te >> 32); m[7] = (unsigned)step;
        pa.rt[(size_t)blockIdx.x * 2 + 1] = wall_clock64();
    }
}

// the library's double-buffered loop for a workgroup alone on its compute unit (2 x 72 KB of LDS), same tile map
template <int WHICH>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_probe_lone(const double* __restrict__ U, int64_t Np,
                                                                const double* __restrict__ Ks, int64_t ldk, int NT,
                                                                const double* __restrict__ avec, double* __restrict__ Qp,
                                                                double* __restrict__ Pp, int64_t ldp, int sm) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    const int nP = (int)(Np / TB);
    int mt, nt, mt2 = -1;
    {
        const int b = blockIdx.x;
        const int x = b & 7, q = b >> 3;
        const int SN = 64 / sm;
        const int per = (NT + 7) / 8;
        const int hper = (per + SN - 1) / SN;
        const int s = q >> 6, r = q & 63;
        const int G = s / hper, H = s - G * hper;
        const int i = G * sm + r / SN;
        const int ln = H * SN + (r - (r / SN) * SN);
        nt = x * per + ln;
        mt = nP - 1 - i;
        if (ln >= per || nt >= NT || i > mt) return;
        if (i < mt) mt2 = i;
    }
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
        if (WHICH == 0) gemm_tile_128_d<1>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, dsm);
        else if (WHICH == 1) gemm_tile_128_g<1>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, dsm);
        else gemm_tile_128_s<1>(acc, U + m0, Np, Ks + (int64_t)nt * Np * TB, TB, 0, (mt + 1) * TB, dsm);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int wm = w >> 1, wn = w & 1;
        double qs[4], ps[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double q = 0.0, p = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = acc[i][j][r];
                    q = fma(v, v, q);
                    p = fma(v, avec[m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * r], p);
                }
            q += __shfl_xor(q, 16); p += __shfl_xor(p, 16);
            q += __shfl_xor(q, 32); p += __shfl_xor(p, 32);
            qs[j] = q; ps[j] = p;
        }
        double* red = dsm;
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
            Qp[(int64_t)mt * ldp + n0 + c] = red[c * 2] + red[(TB + c) * 2];
            Pp[(int64_t)mt * ldp + n0 + c] = red[c * 2 + 1] + red[(TB + c) * 2 + 1];
        }
    }
}

__global__ void k_fill(double* p, size_t n, unsigned seed, double scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * scale;
    }
}

template <int MODE>
static void launch(unsigned nblk, const double* U, int64_t Np, const double* Ks, int64_t ldk, int NT, const double* a,
                   double* Qp, double* Pp, int64_t ldp, ProbeArgs pa) {
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nblk), dim3(GEMM_THREADS), 0, 0, U, Np, Ks, ldk, NT, a, Qp, Pp, ldp, 8, pa);
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int dsleep = argc > 2 ? atoi(argv[2]) : 64;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const int64_t N = argc > 4 ? atoll(argv[4]) : 8192;
    const int64_t cols = argc > 5 ? atoll(argv[5]) : 65536;
    const char* dump = argc > 6 ? argv[6] : nullptr;
    const int64_t Np = (N + 127) / 128 * 128;
    const int nP = (int)(Np / 128), NT = (int)(cols / 128);
    double *U, *Ks, *a, *Qp, *Pp;
    CK(hipMalloc(&U, Np * Np * 8));
    CK(hipMalloc(&Ks, cols * Np * 8));
    CK(hipMalloc(&a, Np * 8));
    CK(hipMalloc(&Qp, (size_t)nP * cols * 8));
    CK(hipMalloc(&Pp, (size_t)nP * cols * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, U, (size_t)(Np * Np), 1u, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Ks, (size_t)(cols * Np), 2u, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, a, (size_t)Np, 3u, 1.0);
    const int sm = 8, SN = 64 / sm, per = (NT + 7) / 8, hper = (per + SN - 1) / SN, gm = ((nP + 1) / 2 + sm - 1) / sm;
    const unsigned nblk = (unsigned)(8 * 64 * hper * gm);
    ProbeArgs pa;
    pa.dsleep = dsleep;
    CK(hipMalloc(&pa.cu_cnt, 4096 * 4));
    pa.stamps = nullptr; pa.meta = nullptr; pa.rt = nullptr;
    if (mode & 1) {
        CK(hipMalloc(&pa.stamps, (size_t)nblk * 4 * MAXSTEP * NSLOT * 4));
        CK(hipMalloc(&pa.meta, (size_t)nblk * 8 * 4));
        CK(hipMemset(pa.stamps, 0, (size_t)nblk * 4 * MAXSTEP * NSLOT * 4));
        CK(hipMemset(pa.meta, 0, (size_t)nblk * 8 * 4));
        CK(hipMalloc(&pa.whw, (size_t)nblk * 16));
        CK(hipMalloc(&pa.rt, (size_t)nblk * 16));
        CK(hipMemset(pa.rt, 0, (size_t)nblk * 16));
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms_all;
    printf("mode %d dsleep %d N %lld cols %lld nblk %u\n", mode, dsleep, (long long)N, (long long)cols, nblk);
    for (int rep = 0; rep < reps + 2; ++rep) {
        CK(hipMemsetAsync(pa.cu_cnt, 0, 4096 * 4, 0));
        hipEventRecord(e0);
        if (mode >= 900000) {       // 900000 + which: the library loops with ONE workgroup per CU (which: 0 double-buffered, 1 single-buffer BK32, 2 the sweep's)
            const size_t lb = (size_t)2 * GEMM_LDS_F64 * 8;
            const int which = mode - 900000;
            if (which == 0) { hipFuncSetAttribute((const void*)k_probe_lone<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL(k_probe_lone<0>, dim3(nblk), dim3(GEMM_THREADS), lb, 0, U, Np, Ks, cols, NT, a, Qp, Pp, cols, 8); }
            else if (which == 1) { hipFuncSetAttribute((const void*)k_probe_lone<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL(k_probe_lone<1>, dim3(nblk), dim3(GEMM_THREADS), lb, 0, U, Np, Ks, cols, NT, a, Qp, Pp, cols, 8); }
            else { hipFuncSetAttribute((const void*)k_probe_lone<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL(k_probe_lone<2>, dim3(nblk), dim3(GEMM_THREADS), lb, 0, U, Np, Ks, cols, NT, a, Qp, Pp, cols, 8); }
        } else
        switch (mode) {
#define C(M) case M: launch<M>(nblk, U, Np, Ks, cols, NT, a, Qp, Pp, cols, pa); break;
            C(0) C(1) C(512) C(8192) C(8704) C(32768) C(40960) C(41472) C(4194304) C(4202496) C(4203008) C(4235776) C(4235777) C(48) C(304)
#undef C
            default: fprintf(stderr, "mode not built\n"); return 1;
        }
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) ms_all.push_back(ms);
        printf("  launch %d: %.3f ms  %.2f TFLOP/s algorithmic\n", rep, ms, (double)N * N * cols / ms / 1e9);
    }
    std::sort(ms_all.begin(), ms_all.end());
    const double med = ms_all[ms_all.size() / 2];
    printf("RESULT mode %d dsleep %d median %.3f ms  %.2f TFLOP/s  frac %.4f\n", mode, dsleep, med,
           (double)N * N * cols / med / 1e9, (double)N * N * cols / med / 1e9 / 78.6);
    // checksum of the results (variants must agree bit for bit)
    {
        std::vector<double> h((size_t)cols);
        CK(hipMemcpy(h.data(), Qp + (size_t)(nP - 1) * cols, cols * 8, hipMemcpyDeviceToHost));
        unsigned long long x = 0;
        for (size_t i = 0; i < h.size(); ++i) { unsigned long long b; memcpy(&b, &h[i], 8); x = x * 1099511628211ull ^ b; }
        printf("checksum Qp[last row block] %016llx\n", x);
    }
    if ((mode & 1) && dump) {
        std::vector<unsigned> meta((size_t)nblk * 8), st((size_t)nblk * 4 * MAXSTEP * NSLOT), whw((size_t)nblk * 4);
        CK(hipMemcpy(whw.data(), pa.whw, whw.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(meta.data(), pa.meta, meta.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(st.data(), pa.stamps, st.size() * 4, hipMemcpyDeviceToHost));
        {
            std::vector<long long> rt((size_t)nblk * 2);
            CK(hipMemcpy(rt.data(), pa.rt, rt.size() * 8, hipMemcpyDeviceToHost));
            double sc = 0, sr = 0;
            for (unsigned b = 0; b < nblk; ++b) {
                if (meta[(size_t)b * 8 + 7] == 0) continue;
                const unsigned long long a0 = meta[(size_t)b * 8 + 3] | ((unsigned long long)meta[(size_t)b * 8 + 4] << 32);
                const unsigned long long a1 = meta[(size_t)b * 8 + 5] | ((unsigned long long)meta[(size_t)b * 8 + 6] << 32);
                sc += (double)(a1 - a0);
                sr += (double)(rt[(size_t)b * 2 + 1] - rt[(size_t)b * 2]);
            }
            printf("s_memtime ticks per s_memrealtime tick (100 MHz): %.4f -> s_memtime runs at %.1f MHz\n", sc / sr, sc / sr * 100.0);
        }
        // keep the workgroups of XCD 0 (all CUs): 1/8 of the launch
        FILE* f = fopen(dump, "wb");
        unsigned hdr[4] = {0, (unsigned)MAXSTEP, (unsigned)NSLOT, 4};
        std::vector<unsigned> keep;
        for (unsigned b = 0; b < nblk; ++b)
            if (meta[(size_t)b * 8 + 7] != 0 && meta[(size_t)b * 8 + 1] < 1 && ((meta[(size_t)b * 8] >> 13) & 7) < 1 && ((meta[(size_t)b * 8] >> 8) & 15) < 4) keep.push_back(b);
        hdr[0] = (unsigned)keep.size();
        fwrite(hdr, 4, 4, f);
        for (unsigned b : keep) {
            fwrite(&b, 4, 1, f);
            fwrite(&meta[(size_t)b * 8], 4, 8, f);
            fwrite(&whw[(size_t)b * 4], 4, 4, f);
            fwrite(&st[(size_t)b * 4 * MAXSTEP * NSLOT], 4, (size_t)4 * MAXSTEP * NSLOT, f);
        }
        fclose(f);
        printf("dumped %zu workgroups to %s\n", keep.size(), dump);
    }
    return 0;
}
