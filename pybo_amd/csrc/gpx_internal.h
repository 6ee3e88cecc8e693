// gpx_internal.h -- handle layout and host-side launcher prototypes (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/gpx.h"

namespace gpx {

constexpr int NB = 128;       // factorisation block = GEMM tile edge
constexpr int TBH = 128;      // host-side copy of the GEMM tile edge (gemm_core.h TB)
constexpr int DMAX = 1024;    // max input dimension (buffer sizing only: the kernels walk coordinates 16 at a time)
constexpr int DMAX_RFF = DMAX;          // max input dimension of the Thompson / RFF kernels (round 3: the projection walks d in chunks)
constexpr int DMAX_RFF_RESIDENT = 64;   // up to here the whole k-range of a feature tile [d][144] lives in LDS at once
constexpr int TOPK_MAX = 4096;   // max k of the device top-k (served TOPK_PASS entries per pass)
constexpr int TOPK_PASS = 64;
constexpr int PEND_MAX = 8;   // appended observations per pass of the sweep-cache correction

enum Timer {
    T_GRAM = 0, T_CHOL, T_TRTRI, T_ALPHA, T_XGRAM, T_TRMM, T_ACQ, T_RFF, T_NLAUNCH, T_FLOP, T_COPY, T_APPEND,
    T_RANK1, T_RFFSWEEP, T_RFFOPS, T_TGFALL, T_SCLK, T_RFFCLK, T_AHEAD, T_COUNT
};

struct EventPair { hipEvent_t a, b; int slot; };

}  // namespace gpx

struct CholGraphKey {             // what a captured factorisation depends on (compared bytewise: zero-filled before use)
    int64_t Np;
    int w, rl, merge, fuse;
    const void *S, *R, *T, *U, *flag;
    hipStream_t s2, s3, s4;
};

struct gpx_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream2 = nullptr;   // library-owned side stream (Cholesky lookahead: far trailing updates, low priority)
    hipStream_t stream3 = nullptr;   // library-owned side stream (rows 2..4 of the next panel's near update, normal priority)
    hipStream_t stream4 = nullptr;   // library-owned side stream (mid(P): the next-but-one panel's rows, low priority)
    hipEvent_t ev_chain = nullptr, ev_far = nullptr, ev_rest = nullptr;
    hipEvent_t ev_row[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int x_skip = 0;               // diagnostic: time parts of the factorisation alone (see launch_cholesky)
    int x_bg = 0, x_bg_lds = 72, x_bg_iters = 10000;   // diagnostic: synthetic MFMA background load (k_bg_mfma)
    hipStream_t stream_bg = nullptr;
    int chol_merge = 1;           // two-panel accumulation of the far updates while >= this many rest block rows (0 = off)
    int chol_fuse = 0;            // diagonal block + panel solve in one launch (k_potrf_solve16)
    int chol_graph = 0;           // replay the factorisation's launches from a captured hipGraph (per size / options / buffers)
    hipGraphExec_t chol_exec = nullptr;
    CholGraphKey chol_key;
    int chol_rl = 1;              // in-panel updates right-looking (1, default) or left-looking (0)
    int chol_w = 0;               // outer panel width of the factorisation in 128-blocks (2..8; 0 = by size)
    // the factorisation as ONE persistent task-graph kernel (kernels_chol_tg.hip)
    int chol_tg = 1;              // 1 (default): task-graph kernel for fits of >= tg_min blocks; 0: the stream schedule
    int tg_min = 2;               // smallest number of 128-blocks the task-graph kernel is used for (round 5, with the shadows: it wins from two blocks on, by the kernel's clock -- N = 256: 0.082 against 0.091 ms -- and, since the abort word travels with the pivot flag in ONE copy, by the host's: fit up to the factor 0.172 against 0.184 ms at N = 200, 0.213 against 0.236 at 300, 0.25 against 0.29 at 500, 0.41 against 0.54 at 1000)
    int tg_max = 160;             // ... and the largest (from N = 24576 on the stream schedule is 1-2 % faster: both throughput-bound)
    int tg_chunks = 0;            // chunk sizes counted back from the pivot, as decimal digits (0 = default by size: 112489 = 1, 1, 2, 4, 8, 16, 16, .., up to 36 blocks 11112489)
    int tg_nap = 0;               // longest polling pause of a waiting workgroup in units of 64 clocks (0 = default 16; round 4 until late: 127)
    int tg_db = -1;               // -1 (default): the double-buffered workers (one workgroup per CU, 2 x 72 KB of LDS) up to tg_db_max blocks; 0 / 1: never / always
    int tg_fuse = 1;            // a column's solve and the final chunk of the tile below it as one task (needs the shadows and one workgroup per CU)
    int tg_db_max = 112;          // (N = 12288: 13.04 against 13.38 ms; N = 16384: 27.6 against 27.2: two workgroups per CU win there)
    int tg_grid = 0;              // workgroups launched (0 = by size, bounded by residency)
    int tg_isolate = -1;          // the critical workgroups keep their compute units to themselves (full grids only): -1 by size (up to 112 blocks), 0 / 1
    int tg_trace = 0;             // diagnostic: stamp the critical path with the kernel's own clock (gpx_chol_trace)
    int tg_tmo_ms = 0;            // bound of every spin in milliseconds (0 = default 2000)
    bool tg_launched = false;     // the last factorisation ran on the task-graph kernel
    int tg_fallbacks = 0;         // launches that gave up (spin bound) and were re-run on the stream schedule
    void* tg = nullptr;           // its cached tables / control block (gpx::TgCache)
    std::string err;

    // model state
    bool fitted = false;
    int stage = 0;               // 0 none, 1 gram, 2 chol (fitted; T/U/a/alpha not formed yet), 3 + inverse
    int ahead_top = 0;               // != 0: the leading part of the inversion has been enqueued on stream2 behind the factorisation's gate (ev_rest marks its end)
    bool want_ahead = false;         // fit_core -> launch_cholesky_tg: the inverse follows at once (stage 3 / eager_inverse)
    int trtri_ahead = 1;             // option: allow that (task-graph factorisation with one workgroup per CU only)
    int trtri_ahead_min = 8;         // ... from this many blocks on (N = 1024: 0.43 -> 0.41 ms factor + inverse, 1536: 0.65 -> 0.58, 2560: 1.17 -> 1.05: below ~24 blocks the factorisation does not fill the chip and the side stream finds free compute units at once)
    bool diag_inv_pending = false;   // the diagonal blocks of T / U hold 16x16 inverses only (k_trtri_diag128 due)
    bool eager_inverse = false;  // option: form the inverse inside the fit (timing experiments)
    int grad_form = 0;           // option: predict-with-gradients form (0 auto: one pass for a single point, 1 two passes, 2 one pass)
    int grad_rb_cs = 0;          // option (experiment): columns per segment of k_tri_matvec_rb (0 = 2048)
    int grad_kernel = -1;        // option: triangular matvec of the two-pass form (-1 auto, 0 one wave per row, 1 register-blocked)
    int trtri_left = 0;          // option: the triangular inverse's recursion as -(T22 L21) T11 instead of -T22 (L21 T11) (measured: no better)
    bool refine_inverse = false; // option: one Newton step on the triangular inverse (left residual)
    double* drefine = nullptr;   // its Np^2 scratch (allocated on first use, capacity cap_np^2)
    int64_t cap_refine = 0;
    int64_t N = 0, Np = 0, d = 0;
    int kernel_id = 0;
    double rho = 1, sn2 = 0, bias = 0;
    std::vector<double> ell;
    int64_t fail_pivot = -1;

    // device buffers (capacity tracked in elements)
    int64_t cap_np = 0, cap_d = 0;
    double* dXs = nullptr;    // (Np, d) observed points scaled by 1/ell, padded rows = 0 (allocated (cap_np, cap_d))
    double* dXraw = nullptr;  // (N, d) observed points as given
    double* dy = nullptr;     // (Np,) y (padded 0)
    double* dS = nullptr;     // (Np,Np) working Gram matrix (upper), later trtri workspace
    double* dR = nullptr;     // (Np,Np) upper Cholesky factor, row-major: K = R^T R
    double* dT = nullptr;     // (Np,Np) T = R^-T (lower), row-major
    double* dU = nullptr;     // (Np,Np) U = R^-1 = T^T (upper), row-major
    double* da = nullptr;     // (Np,) a = T (y - bias)
    double* dalpha = nullptr; // (Np,) alpha = U a
    double* dinvell = nullptr;// (DMAX,) 1/ell
    double* hinv = nullptr;   // pinned host staging of 1/ell (DMAX doubles): its H2D copy needs no host synchronisation
    hipEvent_t ev_inv = nullptr;     // completion of that copy (waited for before the buffer is rewritten)
    bool inv_inflight = false;
    double* hpin = nullptr;   // pinned host staging of predict-with-gradients (GB points)
    int64_t cap_hpin = 0;
    double* hpin_dev = nullptr;   // the same buffer as the device sees it (single-point results are written there directly)
    char* dsmall = nullptr;   // ONE allocation behind dflag / dscal / dinvell / dclk
    unsigned long long* dclk = nullptr;   // {sum of s_memtime ticks, sum of 100 MHz ticks} over the sweep kernel's workgroups: the sustained shader clock
    int* dflag = nullptr;     // [0] = failing pivot + 1 (0 = ok)
    double* dscal = nullptr;  // small scalar scratch (16 doubles)

    // sweep workspace
    int64_t chunk = 0;        // option: candidate columns per chunk (multiple of 128); 0 = by size: 65536, 131072 up to N = 4096 (half the launches: config B 68.4 -> 68.0 ms)
    int super_m = 8;          // rows (mt) of an XCD super-tile of 64 workgroups: 8 -> 8x8, 4 -> 4x16, 2 -> 2x32
    int tile_order = -1;      // -1: by size (7 below 32 block rows, 19 from there on); else bits 0-1 tile map (3 = XCD 8x8 super-tiles of PAIRED tiles), bits 2-4 k-loop schedule (launch_sweep_trmm)
    int64_t cap_ks = 0;       // elements of dKs
    double* dKs = nullptr;    // cross-Gram chunk, tile-blocked [chunk/128][Np][128]
    double* dQp = nullptr;    // (Np/128, chunk) per-row-block partials of colsum(V^2)
    double* dPp = nullptr;    // (Np/128, chunk) per-row-block partials of V^T a
    int64_t cap_part = 0;
    double* dXc = nullptr;    // staging for host candidates (M, d)
    int64_t cap_xc = 0;
    double* dout = nullptr;   // staging for host outputs 3*(M) (acq, mu, s2)
    int64_t cap_out = 0;
    double* dblkv = nullptr;  // per-block top-k values
    int64_t* dblki = nullptr; // per-block top-k indices
    int64_t cap_blk = 0;
    int64_t cap_blki = 0;
    double* dtopv = nullptr;  // final top-k (S*k)
    int64_t* dtopi = nullptr;
    int64_t cap_top = 0;
    const double* last_topv = nullptr;   // device (value, index) pairs of the last sweep's top-k, read by
    const int64_t* last_topi = nullptr;  // gpx_topk_allgather
    int64_t last_topn = 0;
    double* drff = nullptr;   // RFF parameter staging
    int64_t cap_rff = 0;
    double* drffs = nullptr;  // RFF feature-Gram scratch (Phi slabs + split-K partials)
    int64_t cap_rffs = 0;
    double* dgrad = nullptr;  // predict-with-gradient scratch
    int64_t cap_grad = 0;
    double* dens = nullptr;   // ensemble sweep: accumulators + member outputs (5 M)
    int64_t cap_ens = 0;

    // sweep cache (warm BO step): candidates and their reduced sums q = colsum(V^2), p = V^T a of the last full
    // sweep, kept current by gpx_append's rank-1 correction and re-scored by gpx_sweep_update
    const double* app_w = nullptr;   // scratch of the last gpx_append: w = K^-1 k(X, x_new)
    bool cache_on = false;       // option "sweep_cache"
    bool cache_valid = false;
    int64_t cache_M = 0;
    double* dcZ = nullptr;       // (M, d) candidates
    double* dcq = nullptr;       // (M,) q, then p (one allocation: dcp = dcq + cap_cq / 2)
    double* dcp = nullptr;
    int64_t cap_cz = 0, cap_cq = 0;
    // appended observations whose cache correction is still due (applied together, at most PEND_MAX per pass)
    double* dpend = nullptr;     // [PEND_MAX][pend_ld] weight rows, then [PEND_MAX][2] {1/d, a_new}
    int64_t pend_ld = 0;
    int npend = 0;

    // announced observation (gpx_append_begin): the y-independent part of the next append and the correction pass of
    // the sweep cache run ahead, on the third stream, while the caller evaluates its objective
    bool spec_active = false;    // an announcement is waiting for its gpx_append
    bool spec_launched = false;  // the third stream may still be reading the announcement's buffers
    bool spec_used = false;      // the last append_host consumed the announcement (api.hip finishes the cache update)
    bool apply_pending = false;  // ... whose q += v^2, p += v a is still due (applied by flush_pending)
    uint64_t gen = 0, spec_gen = 0;      // model generation (fit / append / grow) now and at the announcement
    std::vector<double> spec_x;  // the announced point (host copy, compared bit for bit)
    double* dspec = nullptr;     // [x xpad][xs xpad][ks Np][g Np][r Np][tu Np][row ldw][pscal 2 + scal 16][v M]
    int64_t cap_spec = 0, spec_ld = 0, spec_M = 0;
    hipEvent_t ev_spec_go = nullptr, ev_spec_done = nullptr;

    double* dbatch = nullptr;    // gpx_loglik_batch: Gram / factor / scaled inputs / a of the batch (one allocation)
    int64_t cap_batch = 0;

    // timers
    std::vector<gpx::EventPair> pending;
    std::vector<hipEvent_t> pool;
    double tacc[gpx::T_COUNT] = {0};
};

namespace gpx {

// launchers (kernels_fit.hip)
void launch_scale_x(hipStream_t s, const double* X, int64_t n, int64_t np, int d, const double* invell,
                    double* Xs);
void launch_gram_sym(hipStream_t s, const double* Xs, int64_t N, int64_t Np, int d, int kernel_id,
                     double rho, double sn2, double* S);
int ensure_side_streams(gpx_handle* h);   // api.hip: streams 2 / 3 + events, on first use
void launch_cholesky(gpx_handle* h);   // S -> R, diag blocks of T/U; sets dflag
bool launch_cholesky_tg(gpx_handle* h);    // the same by the persistent task-graph kernel (kernels_chol_tg.hip); false: not launched
int tg_abort_code(gpx_handle* h);          // after the stream has drained: 0 ok, 1 not PD, 2 a spin gave up
int64_t tg_trace_copy(gpx_handle* h, long long* out, int64_t n);
void tg_free(gpx_handle* h);
int64_t tg_tasks_copy(int nP, int chunks, int16_t* out, int64_t cap, int64_t* counts);
void launch_trtri(gpx_handle* h);      // R, diag blocks -> T, U (uses S as workspace)
void launch_trtri_ahead(gpx_handle* h, hipStream_t s, int top);   // the part that needs the leading `top` block rows only
int trtri_top(int nP);
void launch_refine_inverse(gpx_handle* h, double* tmp);   // option refine_inverse: one Newton step on T / U (tmp: Np^2 scratch)
void launch_alpha(gpx_handle* h);      // a = T (y - bias); alpha = U a
void launch_kinv_diag(gpx_handle* h, double* out);   // out[i] = [K^-1]_ii = sum_m U[i][m]^2
void launch_transpose_lower(hipStream_t s, const double* R, int64_t Np, double* out, int64_t N);
void launch_posterior_wide(hipStream_t s, const double* A, const double* v, const double* z, int n, int64_t np, double sc,
                           double sn2, double* B, double* R, double* work, int* flag, double* theta);   // n >= 128 features

// launchers (kernels_sweep.hip)
void launch_cross_gram(hipStream_t s, const double* Xs, int64_t Np, int64_t N, int d, const double* Xc,
                       int64_t m0, int64_t M, int64_t cols, const double* invell, int kernel_id,
                       double rho, double* Ks, int64_t ldk);
void launch_sweep_trmm(hipStream_t s, const double* U, int64_t Np, const double* Ks, int64_t ldk,
                       int64_t cols, const double* a, double* Qp, double* Pp, int64_t ldp,
                       int tile_order, int super_m, unsigned long long* clk);
// reduce partials, form mu/s2/acq for columns [0,cols) of this chunk -> global candidate m0+..
// nrb = 0: Qp/Pp are reduced per-candidate sums (the sweep cache); qsum/psum (optional) receive the reduced sums
void launch_acq(hipStream_t s, const double* Qp, const double* Pp, int64_t ldp, int nrb, int64_t m0,
                int64_t cols_valid, double rho, double bias, int acq_id, double p0, double* acq_out,
                double* mu_out, double* s2_out, double* qsum, double* psum);
// correction of the cached sums after q <= 8 appended observations in one pass (see kernels_sweep.hip)
void launch_pend_store(hipStream_t s, const double* w, int64_t Nj, int64_t ldw, const double* scal, double* row,
                       double* pscal_j);
void launch_cache_apply(hipStream_t s, const double* v, const double* scal, int64_t M, double* qsum, double* psum);
void launch_scale_point(hipStream_t s, const double* x, const double* invell, int d, double* xs);
void launch_sweep_rank1_v(hipStream_t s, const double* Xs, int64_t Ntot, int d, const double* Wq, int64_t ldw,
                          const double* pscal, const double* Z, int64_t M, const double* invell, int kernel_id,
                          double rho, const double* xlast, double* vout);
void launch_sweep_rankq(hipStream_t s, const double* Xs, int64_t Ntot, int d, const double* Wq, int64_t ldw, int q,
                        const double* pscal, const double* Z, int64_t M, const double* invell, int kernel_id,
                        double rho, double* qsum, double* psum);
// block-local top-k over vals[0..M) then merge -> topv/topi (k entries)
void launch_topk(hipStream_t s, const double* vals, int64_t M, int k, double* blkv, int64_t* blki,
                 int64_t nblk, double* topv, int64_t* topi);
// merge n candidate (value, index) pairs (index == INT64_MAX: no entry; consumed in place) into the k best
void launch_topk_merge(hipStream_t s, double* vals, int64_t* idx, int64_t n, int k, double* topv, int64_t* topi);
void launch_topk_rows(hipStream_t s, const double* vals, int64_t M, int64_t S, int k, double* blkv, int64_t* blki,
                      int64_t nblk, double* topv, int64_t* topi);
int64_t topk_blocks(int64_t M);

// predict with gradients (small M path)
int ensemble_predict_grad_host(gpx_handle* const* mem, int n, const double* Xc, int64_t M, double* mu, double* s2,
                               double* dmu, double* ds2);
int predict_mean_host(gpx_handle* h, const double* Xc, int64_t M, double* mu, double* dmu);
int predict_grad_host(gpx_handle* h, const double* Xc, int64_t M, double* mu, double* s2, double* dmu,
                      double* ds2);

// form T, U, a, alpha if the current fit has not done so yet (api.hip)
int ensure_inverse(gpx_handle* h);

int loglik_host(gpx_handle* h, double* out);
int loglik_batch_host(gpx_handle* h, int64_t B, const double* hyp, double* out);   // kernels_fit.hip
int append_host(gpx_handle* h, const double* x, double ynew);
// the y-independent kernels of an append, enqueued on s: k* = k(X, x), r = T k*, {d, 1/d, (resid - r.a)/d, d^2} -> scal
// (a non-positive d^2 sets *flag), tu = U r.  dx: the point on the device.
int grow_factor_if_full(gpx_handle* h);
void launch_append_prepare(gpx_handle* h, hipStream_t s, const double* dx, double* dks, double* dg, double* dr,
                           double* dtu, double resid, double* scal, int* flag);
// layout of the announcement scratch (pointers into h->dspec)
struct SpecBuf { double *x, *xs, *ks, *g, *r, *tu, *row, *pscal, *scal, *v; };
SpecBuf spec_layout(const gpx_handle* h);
int rff_grad_host(gpx_handle* h, const double* W, const double* b, const double* theta, int64_t n, int64_t d,
                  double bias, const double* Xc, int64_t M, double* f, double* g);

// launchers (kernels_rff.hip)
extern int g_rff_variant;
void launch_rff_mfma(hipStream_t s, const double* Wt, const double* bt, const double* tt, int S, int nfb, int n, int d,
                     int dp, double bias, const double* Xc, int64_t M, double* vals, unsigned long long* clk);
int64_t rff_gram_batch_scratch(int64_t S, int64_t Np);
void launch_rff_posterior(hipStream_t s, const double* A, const double* v, const double* z, int S, int n, double sc,
                          double sn2, double* theta, int* flag);
void launch_rff_gram_batch(hipStream_t s, const double* Xraw, int64_t N, int64_t Np, int d, int dp,
                           const double* Wt, const double* bt, int S, int n, const double* y, double bias,
                           double* scratch, double* A, double* v);
void launch_rff_gram(hipStream_t s, const double* Xraw, const double* Ft_scratch, int64_t N, int d,
                     const double* W, const double* b, int n, const double* y, double bias, double* A,
                     double* v);

// launchers (kernels_ens.hip)
void launch_ens_accum(hipStream_t s, double* acc0, double* acc1, const double* t0, const double* t1, int64_t M,
                      int mode, int first);
void launch_ens_finish(hipStream_t s, const double* acc0, const double* acc1, int64_t M, int mode, double n,
                       double beta, double* out, double* mu_out, double* s2_out);
void launch_grid_sobol(hipStream_t s, const uint32_t* sv, int bits, int64_t first, int64_t M, int d,
                       const double* bounds, double* X);
void launch_grid_uniform(hipStream_t s, uint64_t seed, int64_t first, int64_t M, int d, const double* bounds,
                         double* X);

}  // namespace gpx
