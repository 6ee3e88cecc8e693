/* A non-Python consumer of include/gpx.h doing DEVICE work: strict C99, linked against libgpx.so.
 * The model-protocol call sites it stands for: pybo/policies/simple.py:20-25 (target = model.predict(X)[0].max();
 * index = model.get_improvement(target, .)) and pybo/solvers/lbfgs.py:50-51 (finit = f(xgrid); argsort(finit)[::-1]).
 *
 *   abi_sweep problem.bin            gpx_create -> gpx_fit -> gpx_mean_at_obs -> gpx_sweep -> gpx_get_vectors
 *   abi_sweep problem.bin comm       the sharded listing of INTEGRATION.md section 3 with a ONE-rank communicator:
 *                                    ... -> gpx_comm_unique_id -> gpx_comm_init -> gpx_sweep (my shard) -> gpx_topk_allgather
 *
 * problem.bin (little-endian): int64 N, d, M, k; then doubles: X[N*d], y[N], ell[d], rho, sn2, bias, Xc[M*d].
 * Output (text, %.17g): "target t", "top i idx val" x k, "acq j val" for every 97th candidate, "alpha j val" for every 31st
 * observation -- compared by tests/test_gpu_c_consumer.py with oracle/gp_ref.py.  Exit code != 0 on any error. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gpx.h"

static int fail(const char *what, gpx_handle *h) {
    fprintf(stderr, "abi_sweep: %s: %s\n", what, gpx_last_error(h));
    return 1;
}

static int read_all(FILE *f, void *dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : 1; }

int main(int argc, char **argv) {
    int64_t hdr[4], N, d, M, k, i;
    double *X, *y, *ell, *Xc, *val, *acq, *a, *alpha, scal[3], target = 0.0;
    int64_t *idx;
    gpx_handle *h = 0;
    FILE *f;
    int with_comm = argc > 2 && strcmp(argv[2], "comm") == 0;
    if (argc < 2) { fprintf(stderr, "usage: abi_sweep problem.bin [comm]\n"); return 2; }
    f = fopen(argv[1], "rb");
    if (!f || read_all(f, hdr, sizeof hdr)) { fprintf(stderr, "abi_sweep: cannot read %s\n", argv[1]); return 2; }
    N = hdr[0]; d = hdr[1]; M = hdr[2]; k = hdr[3];
    X = (double *)malloc((size_t)(N * d) * 8); y = (double *)malloc((size_t)N * 8); ell = (double *)malloc((size_t)d * 8);
    Xc = (double *)malloc((size_t)(M * d) * 8); val = (double *)malloc((size_t)k * 8); idx = (int64_t *)malloc((size_t)k * 8);
    acq = (double *)malloc((size_t)M * 8); a = (double *)malloc((size_t)N * 8); alpha = (double *)malloc((size_t)N * 8);
    if (!X || !y || !ell || !Xc || !val || !idx || !acq || !a || !alpha) return 2;
    if (read_all(f, X, (size_t)(N * d) * 8) || read_all(f, y, (size_t)N * 8) || read_all(f, ell, (size_t)d * 8) ||
        read_all(f, scal, sizeof scal) || read_all(f, Xc, (size_t)(M * d) * 8)) { fprintf(stderr, "abi_sweep: short file\n"); return 2; }
    fclose(f);

    if (gpx_create(0, NULL, &h) != GPX_OK) return fail("gpx_create", NULL);
    if (gpx_fit(h, X, N, d, y, GPX_KERN_SE_ARD, ell, scal[0], scal[1], scal[2]) != GPX_OK) return fail("gpx_fit", h);
    if (gpx_mean_at_obs(h, NULL, &target) != GPX_OK) return fail("gpx_mean_at_obs", h);      /* the EI target */
    printf("target %.17g\n", target);
    if (!with_comm) {
        if (gpx_sweep(h, GPX_ACQ_EI, &target, 1, Xc, M, k, val, idx, acq, NULL, NULL) != GPX_OK) return fail("gpx_sweep", h);
    } else {
        /* one process per GPU; here the world has one rank, so "my shard" is the whole grid */
        unsigned char id[128];
        gpx_comm *comm = 0;
        int rank = -1, nranks = -1;
        const int64_t lo = 0, hi = M;
        if (gpx_comm_unique_id(id) != GPX_OK) { fprintf(stderr, "abi_sweep: gpx_comm_unique_id: %s\n", gpx_comm_last_error()); return 1; }
        if (gpx_comm_init(h, 0, 1, id, &comm) != GPX_OK) { fprintf(stderr, "abi_sweep: gpx_comm_init: %s\n", gpx_comm_last_error()); return 1; }
        if (gpx_comm_size(comm, &rank, &nranks) != GPX_OK || rank != 0 || nranks != 1) return fail("gpx_comm_size", h);
        if (gpx_sweep(h, GPX_ACQ_EI, &target, 1, Xc + lo * d, hi - lo, k, val, idx, acq, NULL, NULL) != GPX_OK) return fail("gpx_sweep", h);
        memset(val, 0, (size_t)k * 8);
        memset(idx, 0, (size_t)k * 8);
        if (gpx_topk_allgather(comm, k, lo, k, val, idx) != GPX_OK) { fprintf(stderr, "abi_sweep: gpx_topk_allgather: %s\n", gpx_comm_last_error()); return 1; }
        if (gpx_comm_destroy(comm) != GPX_OK) return fail("gpx_comm_destroy", h);
    }
    if (gpx_get_vectors(h, a, alpha) != GPX_OK) return fail("gpx_get_vectors", h);
    for (i = 0; i < k; ++i) printf("top %lld %lld %.17g\n", (long long)i, (long long)idx[i], val[i]);
    for (i = 0; i < M; i += 97) printf("acq %lld %.17g\n", (long long)i, acq[i]);
    for (i = 0; i < N; i += 31) printf("alpha %lld %.17g\n", (long long)i, alpha[i]);
    if (gpx_destroy(h) != GPX_OK) return 1;
    free(X); free(y); free(ell); free(Xc); free(val); free(idx); free(acq); free(a); free(alpha);
    return 0;
}
