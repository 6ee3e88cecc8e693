/*
 * gpx.h -- C-ABI of libgpx.so: the MI355X (gfx950) GP-posterior + acquisition engine.
 *
 * The reference (mwhoffman/pybo) has NO native boundary: its GP arithmetic is a set of Python
 * method calls on a duck-typed `model` object (the un-vendored `reggie` package).  Each entry point
 * below therefore cites the *call site in pybo* whose work it performs; the Python model/policy/
 * solver plugins in pybo_amd/ bind these through ctypes.
 *
 * Conventions
 *   - every function returns an int status: 0 = GPX_OK, negative = error class;
 *     gpx_last_error(h) returns a NUL-terminated description (handle-local; NULL handle ->
 *     thread-local string of the last failed gpx_create).
 *   - caller owns every host buffer; arrays are row-major C-contiguous float64 / int64.
 *   - `_dev` variants take DEVICE pointers (e.g. torch.Tensor.data_ptr()) valid on the handle's
 *     device; they enqueue on the handle's stream and are synchronous on return only where an
 *     output lands in a host buffer (top-k, status).
 *   - a handle is not re-entrant; distinct handles may be driven from distinct threads.
 *   - a handle owns up to four HIP streams (its own or the caller's + three side streams created on first need); a process
 *     that keeps several handles alive should export GPU_MAX_HW_QUEUES=8 BEFORE its first HIP call, or work meant to overlap
 *     inside a handle may share a hardware queue and serialise (+2.3 ms per warm iteration at N = 8192).  Results never depend on it.
 *   - no C++ exception crosses this boundary.
 */
#ifndef GPX_H
#define GPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpx_handle gpx_handle;

enum gpx_status {
    GPX_OK = 0,
    GPX_EARG = -1,    /* bad argument */
    GPX_ENOTPD = -2,  /* K + sn2*I not positive definite; see gpx_fail_pivot */
    GPX_EHIP = -3,    /* HIP runtime error */
    GPX_EOOM = -4,    /* device allocation failed */
    GPX_ESTATE = -5,  /* call order error (e.g. sweep before fit) */
    GPX_ERCCL = -6    /* RCCL could not be loaded, or a collective failed; see gpx_comm_last_error */
};

/* covariance functions; parametrisation (sn2, rho, ell[d], bias) = the arguments of
 * reggie.make_gp(sn2, rho, ell, bias)                       [pybo/bayesopt.py:98-105] */
enum gpx_kernel {
    GPX_KERN_SE_ARD = 0,   /* rho * exp(-r2/2),                      r2 = sum_k ((x_k-z_k)/ell_k)^2 */
    GPX_KERN_MATERN52 = 1, /* rho * (1 + sqrt5 r + 5 r2/3) exp(-sqrt5 r) */
    GPX_KERN_MATERN32 = 2, /* rho * (1 + sqrt3 r) exp(-sqrt3 r) */
    GPX_KERN_MATERN12 = 3  /* rho * exp(-r) */
};

/* acquisition functions evaluated by the sweep */
enum gpx_acq {
    GPX_ACQ_EI = 0,   /* model.get_improvement(target, X)  [pybo/policies/simple.py:25]  params = {target} */
    GPX_ACQ_PI = 1,   /* model.get_tail(target, X)         [pybo/policies/simple.py:39]  params = {target} */
    GPX_ACQ_UCB = 2,  /* mu + sqrt(beta*s2)                [pybo/policies/simple.py:62-73] params = {beta} */
    GPX_ACQ_MEAN = 3  /* posterior mean                    [pybo/recommenders.py:19-25]  no params */
};

/* ---- lifetime --------------------------------------------------------------------------- */
/* `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for a
 * library-owned stream. */
int gpx_create(int device, void *stream, gpx_handle **out);
int gpx_destroy(gpx_handle *h);
const char *gpx_last_error(const gpx_handle *h);
int gpx_version(void);
/* Options (int64 values).  Every option below selects among schedules that give BIT-IDENTICAL results unless it says
 * otherwise; defaults in brackets.  The measurements behind the defaults: DESIGN.md section 4.
 *   "chunk"          candidate columns per sweep chunk, a multiple of 128 [by size: 65536; 131072 up to N = 4096]
 *   "tile_order"     sweep-kernel schedule: bits 0-1 blockIdx->tile map (0 linear heavy-first, 1 per-XCD candidate slices, 2 per-XCD
 *                    8x8 super-tiles, 3 the same with every workgroup computing the PAIR of tiles (nP-1-i, nt), (i, nt): equal work),
 *                    bits 2-4 k-loop (4 = operands by LDS-DMA, k-step 32, two workgroups per CU, the all-zero quarter-rows of T's
 *                    diagonal block skipped; 3 = the same without the skip; 1 = barrier-free, every wave fetching its own operand
 *                    halves; 7 = k-step 16, three workgroups per CU; 6 / 5 / 2 = the
 *                    register-staged loops of rounds 5 / 2 / 1: independently scheduled witnesses) [-1 = by size: 7 below 32 block rows, else 19]
 *   "super_m"        rows of the XCD super-tile of 64 workgroups: 1, 2, 4, 8, 16 [8 -> 8 x 8]
 *   "sweep_cache"    1: full sweeps keep candidates and reduced sums for gpx_sweep_update; 0: leave a live cache alone; -1: drop it [0]
 *   "eager_inverse"  1: form the triangular inverse inside gpx_fit instead of on first use [0]
 *   "trtri_ahead"    1: when the inverse is certain or likely to follow a fit, its part that needs only the factor's leading block
 *                    rows runs on a side stream behind the factorisation's tail, from "trtri_ahead_min" (8) blocks on [1]
 *   "trtri_left"     1: T21 = -(T22 L21) T11 instead of -T22 (L21 T11); agrees to rounding, NOT bit for bit [0]
 *   "refine_inverse" 1: one Newton step on the inverse, T <- (2I - T R^T) T (squares the left residual; for cond(K) >~ 1e9) [0]
 *   "grad_form"      gpx_predict WITH gradients: 0 auto (one point: one pass over T with 1 + d right-hand sides; batches: two
 *                    passes), 1 always two passes, 2 always one; the forms agree to rounding [0]
 *   "grad_kernel"    triangular matvec of the two-pass form: -1 auto, 0 one wave per row, 1 register-blocked [-1]
 *   "chol_tg"        1: the factorisation is ONE persistent task-graph kernel for fits of "chol_tg_min" (2) .. "chol_tg_max" (160)
 *                    128-blocks; 0: the stream schedule (~250 launches over four streams) [1].  "chol_tg_db" -1 / 0 / 1: one workgroup per
 *                    CU with two k-step images of LDS up to "chol_tg_db_max" (112) blocks / never / always [-1]; "chol_tg_fuse" 1: a
 *                    column's solve and the final chunk of the tile below it are one task [1]; "chol_tg_tmo_ms": bound of every spin
 *                    -- on expiry the fit re-runs on the stream schedule and says so on stderr [2000]
 *   "chol_w", "chol_rl", "chol_merge", "chol_fuse", "chol_graph"   the stream schedule: outer panel width in blocks (2..8, 0 = by
 *                    size), right- / left-looking in-panel updates [1], far updates of two panels in one pass [1], diagonal block
 *                    factored inside the panel solve [0], replay from a captured hipGraph [0]
 * Diagnostic knobs ("chol_tg_chunks", "chol_tg_nap", "chol_tg_grid", "chol_tg_isolate", "chol_tg_trace", "grad_rb_cs", "x_rff",
 * "x_bg*", "x_skip" -- the last one leaves parts of the factorisation OUT) exist only in a library built with -DGPX_DIAGNOSTICS
 * (pybo_amd/csrc/libgpx_diag.so; pybo_amd/csrc/gpx_diag.h); the shipping library answers them with GPX_EARG.
 * Environment: GPX_OPTIONS="name=value,name=value" applies options to every handle at creation (A/B runs through a plug-in layer
 * whose handles the caller never sees); a bad entry fails gpx_create with GPX_EARG. */
int gpx_set_option(gpx_handle *h, const char *name, int64_t value);

/* ---- GP fit = model.add_data(X, Y)            [pybo/bayesopt.py:114,258,269] ------------- */
/* Limits: 1 <= d <= 1024 for every entry point (the Thompson / RFF kernels keep a whole feature tile in LDS up to
 * d = 64 and walk the coordinates 32 at a time beyond); top-k requests k <= 4096 (64 per pass over the values).
 * Gram build K = k(X,X) + sn2 I and Cholesky K = R^T R.  The triangular inverse T = R^-T, a = T (y - bias)
 * and alpha follow on FIRST USE (sweep, predict, mean_at_obs, loglik, append, introspection): the Thompson
 * entry points never read them.  X is (N,d), y is (N,), ell is (d,) on the HOST in both variants. */
int gpx_fit(gpx_handle *h, const double *X, int64_t N, int64_t d, const double *y, int kernel_id,
            const double *ell, double rho, double sn2, double bias);
int gpx_fit_dev(gpx_handle *h, const double *dX, int64_t N, int64_t d, const double *dy,
                int kernel_id, const double *ell, double rho, double sn2, double bias);
/* log marginal likelihood of the fitted model, -1/2 a.a - sum log R_ii - N/2 log 2pi: what a
 * hyper-parameter sampler (reggie.MCMC, pybo/bayesopt.py:115) evaluates once per proposal. */
int gpx_loglik(gpx_handle *h, double *out);
/* The same for B hyper-parameter vectors at once on the handle's RESIDENT data, without touching the handle's own fit: hypers
 * (B, d + 3) row-major [sn2, rho, ell_1..d, bias] (the argument order of reggie.make_gp, pybo/bayesopt.py:105), out (B,); -inf
 * where K + sn2 I is not positive definite.  One batched launch chain and one host synchronisation per 64 vectors: what
 * reggie.MCMC(model, n=10, burn=100) [pybo/bayesopt.py:115] repeats per proposal. */
int gpx_loglik_batch(gpx_handle *h, int64_t B, const double *hypers, double *out);
/* Incremental fit: absorb ONE more observation x (d,), y into the current factorisation in O(N^2) (two memory-bound passes over
 * T and U) instead of refitting -- the per-iteration `model.add_data(x, y)` of the BO loop [pybo/bayesopt.py:269].  A full
 * 128-block is extended inside buffers allocated with head-room (device copy, no refit).  GPX_ENOTPD like gpx_fit.  A live sweep
 * cache (below) is corrected for the new observation in the same call (one N*M pass). */
int gpx_append(gpx_handle *h, const double *x, double y);
/* ANNOUNCE the next observation's location before its value exists (between `x, _ = solver(index, bounds)` and
 * `y = objective(x)`, pybo/bayesopt.py:265-268: the time the reference spends idle): everything of the coming gpx_append(h, x, y)
 * that does not depend on y is enqueued now without a host synchronisation -- k(X, x), the two triangular passes and, on a side
 * stream, the N*M covariance evaluations of the sweep-cache correction.  A following gpx_append with the bit-identical x then costs
 * O(N + M); any other x, or a model change, and the announcement is ignored.  Results bit-identical to an unannounced append.
 * GPX_ESTATE (nothing started, not an error) without a live sweep cache or when the next append has to add a 128-block first. */
int gpx_append_begin(gpx_handle *h, const double *x);
/* 0-based index of the failing pivot of the last GPX_ENOTPD fit, else -1. */
int64_t gpx_fail_pivot(const gpx_handle *h);

/* introspection for parity tests (host outputs): which = 0: L (N,N) lower Cholesky factor, row-major (K + sn2 I = L L^T);
 * 1: T = L^-1 (N,N) lower; 2: K + sn2 I, upper triangle (only after gpx_fit_stage(.., 1): the factorisation consumes it) */
int gpx_get_matrix(gpx_handle *h, int which, double *out);
/* a = L^-1 (y - bias) (N,) and alpha = (K + sn2 I)^-1 (y - bias) (N,) */
int gpx_get_vectors(gpx_handle *h, double *a, double *alpha);
/* debugging/parity: run the fit but stop after the Gram build (stage=1) or Cholesky (stage=2) */
int gpx_fit_stage(gpx_handle *h, const double *X, int64_t N, int64_t d, const double *y,
                  int kernel_id, const double *ell, double rho, double sn2, double bias, int stage);

/* posterior (latent) mean at the N observed points, closed form y - sn2*alpha = model.predict(X_obs)[0] [pybo/policies/simple.py:21,35] */
int gpx_mean_at_obs(gpx_handle *h, double *mu_host, double *mu_max);
/* posterior (latent) variance at the N observed points, closed form sn2 - sn2^2 [K^-1]_ii with
 * [K^-1]_ii = sum_m U[i][m]^2 (one HBM-read pass over U = R^-1; a sweep over X_obs is an N x N x N product).
 * = model.predict(X_obs)[1]                     [pybo/policies/simple.py:21,35; pybo/recommenders.py:34] */
int gpx_var_at_obs(gpx_handle *h, double *s2_host);
/* rows (a multiple of 128) the handle's factor buffers are allocated for: 4 matrices of capacity^2 doubles stay
 * on the device until gpx_destroy -- what a pool of handles should be sized by. */
int64_t gpx_capacity(const gpx_handle *h);

/* ---- posterior moments = model.predict(X, grad) [pybo/policies/simple.py:64] ------------- */
/* Xc (M,d) -> mu (M,), s2 (M,) latent variance; dmu, ds2 (M,d) optional (NULL to skip).  With gradients a single point
 * (M = 1) is answered in the one-pass form, batches in the two-pass form: option "grad_form" above. */
int gpx_predict(gpx_handle *h, const double *Xc, int64_t M, double *mu, double *s2, double *dmu, double *ds2);
/* The mean alone, mu (M,) and optionally dmu (M,d) (NULL to skip): mu = bias + k(x, X).alpha reads neither the
 * factor nor its inverse -- what the latent recommender maximises, model.predict(X, True)[0::2]
 * [pybo/recommenders.py:17-24], without the two triangular passes per point the variance costs. */
int gpx_predict_mean(gpx_handle *h, const double *Xc, int64_t M, double *mu, double *dmu);

/* ---- acquisition sweep + top-k = the batched index call and argsort of the solver
 *      finit = f(xgrid); idx = argsort(finit)[::-1]       [pybo/solvers/lbfgs.py:50-51] ---- */
/* Evaluates acq over M candidates, returns the k best (value desc, then index asc) in host
 * buffers top_val[k], top_idx[k] (indices are LOCAL to Xc; add your shard offset).  acq_all, mu,
 * s2 (each (M,), host in gpx_sweep / device in gpx_sweep_dev) are optional (NULL to skip). */
int gpx_sweep(gpx_handle *h, int acq_id, const double *params, int nparams, const double *Xc,
              int64_t M, int64_t k, double *top_val, int64_t *top_idx, double *acq_all, double *mu,
              double *s2);
int gpx_sweep_dev(gpx_handle *h, int acq_id, const double *params, int nparams, const double *dXc,
                  int64_t M, int64_t k, double *top_val, int64_t *top_idx, double *d_acq_all,
                  double *d_mu, double *d_s2);

/* ---- warm BO step: the NEXT iteration's `index(xgrid)` over the SAME grid with the SAME hyper-parameters
 *      [pybo/bayesopt.py:262-269 with a fixed `xgrid=` in pybo/solvers/lbfgs.py:42-50].  The reference pays a full refit and a full
 *      solve again; with option "sweep_cache" = 1 a full gpx_sweep* keeps the candidates and their reduced sums q = colsum(V^2),
 *      p = V^T a in HBM (8 (d + 2) M bytes), every gpx_append adds the one new row of V to them (N*M covariance evaluations; up to 8
 *      appended points share one pass), and gpx_sweep_update re-scores the whole grid in O(M) (the target / beta may change from call
 *      to call) + top-k.  Outputs as in gpx_sweep*.  GPX_ESTATE without a valid cache (never swept, or refitted since). */
int gpx_sweep_update(gpx_handle *h, int acq_id, const double *params, int nparams, int64_t k, double *top_val,
                     int64_t *top_idx, double *acq_all, double *mu, double *s2);
int gpx_sweep_update_dev(gpx_handle *h, int acq_id, const double *params, int nparams, int64_t k,
                         double *top_val, int64_t *top_idx, double *d_acq_all, double *d_mu, double *d_s2);
/* number of candidates in the live sweep cache (0: none) */
int64_t gpx_sweep_cache_size(const gpx_handle *h);

/* ---- Thompson sampling = model.sample_f(n, rng).get(X)   [pybo/policies/simple.py:44-48] -- */
/* S random-Fourier-feature posterior draws f_s(x) = bias + sum_j theta[s][j] cos(W[s][j].x + b[s][j])
 * (the sqrt(2 rho/n) factor is folded into theta by the caller).  W (S,n,d), b (S,n), theta (S,n)
 * on the host.  For each draw returns the k best candidates: top_val (S,k), top_idx (S,k).
 * vals_all (S,M) optional. */
int gpx_rff_sweep(gpx_handle *h, const double *W, const double *b, const double *theta, int64_t S,
                  int64_t n, int64_t d, double bias, const double *Xc, int64_t M, int64_t k,
                  double *top_val, int64_t *top_idx, double *vals_all);
int gpx_rff_sweep_dev(gpx_handle *h, const double *W, const double *b, const double *theta,
                      int64_t S, int64_t n, int64_t d, double bias, const double *dXc, int64_t M,
                      int64_t k, double *top_val, int64_t *top_idx, double *d_vals_all);
/* value f (M,) and gradient g (M,d) of ONE draw at M points (host buffers): the
 * `f(x[None], grad=True)` calls of the L-BFGS refinement   [pybo/solvers/lbfgs.py:56-58] */
int gpx_rff_grad(gpx_handle *h, const double *W, const double *b, const double *theta, int64_t n,
                 int64_t d, double bias, const double *Xc, int64_t M, double *f, double *g);
/* feature Gram for the weight posterior: Phi = cos(X_obs W^T + b) (N,n) on the device's X_obs;
 * returns A = Phi^T Phi (n,n) and v = Phi^T (y - bias) (n,) in host buffers. */
int gpx_rff_gram(gpx_handle *h, const double *W, const double *b, int64_t n, double *A, double *v);
/* the same for S draws in one call: W (S,n,d), b (S,n) -> A (S,n,n), v (S,n) */
int gpx_rff_gram_batch(gpx_handle *h, const double *W, const double *b, int64_t S, int64_t n, double *A,
                       double *v);
/* the weight posterior of S draws WITHOUT leaving the device (the n x n solve inside sample_f,
 * pybo/policies/simple.py:48): feature Grams as above, then per draw
 *     B = sc^2 A + sn2 I = L L^T,    theta = sc ( B^-1 (sc v) + sqrt(sn2) L^-T z )
 * with z (S,n) the caller's standard-normal draws, sc = sqrt(2 rho / n) and sn2 the fitted noise variance (> 0);
 * theta (S,n) comes back ready for gpx_rff_sweep* / gpx_rff_grad.  n <= 127: all S draws in one launch chain (the n x n
 * posterior of a draw lives in LDS); 128 <= n <= 4096: per draw the blocked Cholesky kernels of the fit + two vector
 * substitutions (`n` is a free keyword of the reference's sample_f, pybo/policies/simple.py:44).  GPX_ENOTPD if a B is not PD. */
int gpx_rff_posterior(gpx_handle *h, const double *W, const double *b, const double *z, int64_t S, int64_t n,
                      double sc, double *theta);

/* ---- hyper-parameter ensemble = pybo's DEFAULT model, reggie.MCMC(gp, n=10)
 *      [pybo/bayesopt.py:115]: every index is the average over the n member GPs.  `members` are fitted
 *      handles on ONE device with the same input dimension (members[0] owns the scratch and the error text).
 *      EI / PI: value = mean_m acq_m(x).  UCB / MEAN: mixture moments mu = mean_m mu_m,
 *      s2 = mean_m(s2_m + mu_m^2) - mu^2, value = mu + sqrt(params[0] * s2) (UCB) or mu (MEAN); only these
 *      two can return mu / s2.  The member sweeps never leave the device; sums run in member order and are
 *      divided once by n.  Outputs as in gpx_sweep / gpx_sweep_dev. */
int gpx_ensemble_sweep(gpx_handle *const *members, int n_members, int acq_id, const double *params,
                       int nparams, const double *Xc, int64_t M, int64_t k, double *top_val,
                       int64_t *top_idx, double *acq_all, double *mu, double *s2);
int gpx_ensemble_sweep_dev(gpx_handle *const *members, int n_members, int acq_id, const double *params,
                           int nparams, const double *dXc, int64_t M, int64_t k, double *top_val,
                           int64_t *top_idx, double *d_acq_all, double *d_mu, double *d_s2);
/* per-member posterior moments AND gradients at M points (host buffers), member-major: mu, s2 (n_members, M);
 * dmu, ds2 (n_members, M, d) -- the `f(x, grad=True)` calls of the L-BFGS refinement on the default model
 * [pybo/solvers/lbfgs.py:56-58 over pybo/bayesopt.py:115]; the members' latency-bound kernels run concurrently on
 * their own streams (one call instead of n_members gpx_predict calls) and the caller forms the average it needs
 * (mixture moments for UCB / mean, the mean of the members' EI / PI and their gradients). */
int gpx_ensemble_predict(gpx_handle *const *members, int n_members, const double *Xc, int64_t M, double *mu,
                         double *s2, double *dmu, double *ds2);

/* ---- candidate grid generated and kept in HBM = the solver's grid
 *      xgrid = init_uniform(bounds, ngrid, rng)   [pybo/solvers/lbfgs.py:45; pybo/inits/methods.py:24-38]
 *      or a Sobol' grid                            [pybo/inits/methods.py:62-77]
 *      without the host array and its PCIe upload; pass gpx_grid_data() to the *_dev sweeps.
 * GPX_GRID_UNIFORM: counter-based Philox4x32-10, key = seed, counter = element-pair index; element 2c, 2c+1
 *      of the row-major (first+M,d) array = the two 53-bit uniforms of output c; x = lo + u*(hi-lo); rows
 *      first .. first+M-1 of that array are generated (a shard of a grid holds the numbers the whole grid holds).
 * GPX_GRID_SOBOL: unscrambled Sobol' points first .. first+M-1 in Gray-code order (the order of
 *      scipy.stats.qmc.Sobol), from caller-supplied direction numbers sv (d, bits) uint32, 1 <= bits <= 32.
 * bounds (d,2) row-major [lo, hi].  Errors of the grid calls are read with gpx_last_error(NULL). */
typedef struct gpx_grid gpx_grid;
enum { GPX_GRID_UNIFORM = 0, GPX_GRID_SOBOL = 1 };
int gpx_grid_create(int device, int kind, const double *bounds, int64_t M, int64_t d, uint64_t seed,
                    int64_t first, const uint32_t *sv, int bits, gpx_grid **out);
const double *gpx_grid_data(const gpx_grid *g);     /* device pointer, (M,d) row-major */
/* rows idx[0..k) -> out (k,d) host; idx == NULL: the whole grid (M,d) */
int gpx_grid_rows(gpx_grid *g, const int64_t *idx, int64_t k, double *out);
int gpx_grid_destroy(gpx_grid *g);

/* ---- multi-GPU exchange: the one collective of the sharded sweep.  The reference is single-process (its only hint at
 *      parallelism is the comment pybo/solvers/lbfgs.py:60); layout per SURVEY.md 8(e): one process per GPU, candidates sharded
 *      contiguously, every rank fits redundantly (bitwise-identical factor), each rank's best (value, GLOBAL index) pairs
 *      all-gathered over RCCL/xGMI and merged "value descending, then index ascending".  RCCL is bound at run time (dlopen of
 *      librccl, override with $GPX_RCCL_LIB): single-GPU use never loads it.  Errors: gpx_comm_last_error() (thread-local). */
typedef struct gpx_comm gpx_comm;
#define GPX_COMM_ID_BYTES 128
/* rank 0 generates the id (ncclGetUniqueId) and hands its 128 bytes to every rank over any side channel */
int gpx_comm_unique_id(unsigned char *id);
/* collective over all ranks: binds the communicator to the handle's device and stream */
int gpx_comm_init(gpx_handle *h, int rank, int nranks, const unsigned char *id, gpx_comm **out);
int gpx_comm_destroy(gpx_comm *c);
int gpx_comm_size(const gpx_comm *c, int *rank, int *nranks);
const char *gpx_comm_last_error(void);
/* All-gather of the n (value, index) pairs the handle's LAST sweep left in HBM (n = its k for gpx_sweep* /
 * gpx_ensemble_sweep*, S*k for gpx_rff_sweep*), taken straight from the device, `index_offset` (this rank's
 * shard origin) added to every valid index.
 *   k > 0: every rank receives the same merged k best in out_val[k], out_idx[k]
 *          = the global argsort(finit)[::-1][:k] of pybo/solvers/lbfgs.py:51 over the sharded grid.
 *   k = 0: no merge; out_val / out_idx receive all nranks*n pairs in rank order (batch-BO: one recommendation
 *          per Thompson draw, draws sharded over ranks). */
int gpx_topk_allgather(gpx_comm *c, int64_t n, int64_t index_offset, int64_t k, double *out_val,
                       int64_t *out_idx);

/* ---- measurement ------------------------------------------------------------------------ */
/* Stage timers in milliseconds accumulated since the last reset (HIP events on the handle's stream): [0] gram [1] cholesky
 * [2] trtri [3] alpha [4] cross_gram [5] sweep_trmm [6] acq_topk [7] rff [8] sweep_trmm launches [9] sweep_trmm algorithmic flop
 * [10] h2d/d2h copies [11] append [12] correction passes over the sweep cache [13] the Thompson sweep kernel alone (part of [7])
 * [14] its algorithmic fp64 lane operations, S n (d + 20) M per launch [15] fits whose task-graph factorisation gave up and re-ran on
 * the stream schedule [16] the shader clock in MHz the sweep_trmm launches sustained (their workgroups' s_memtime over s_memrealtime
 * ticks) [17] the same for the Thompson sweep kernel [18] inversions whose leading part ran behind the factorisation ("trtri_ahead":
 * for those [2] holds only what was left after the factor was done).  Synchronises the stream.  Returns the slots written (<= n). */
int gpx_timers(gpx_handle *h, double *out, int n, int reset);
/* 1 when the library was built with -DGPX_DIAGNOSTICS (the diagnostic options above are accepted), else 0 */
int gpx_diagnostics(void);
int gpx_sync(gpx_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
