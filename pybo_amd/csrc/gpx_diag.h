/*
 * gpx_diag.h -- diagnostic entry points and options of libgpx (NOT part of the shipping C-ABI, include/gpx.h).
 *
 * The entry points below are exported by every build (they only read).  The OPTIONS below are accepted by gpx_set_option
 * only in a library built with -DGPX_DIAGNOSTICS (build.sh builds it next to the shipping one: pybo_amd/csrc/libgpx_diag.so;
 * Python: GPX_DIAGNOSTICS=1 in the environment selects it, tests/conftest.py does); the shipping libgpx.so answers them
 * with GPX_EARG -- a consumer of include/gpx.h cannot switch parts of a factorisation off by a typo.
 *
 *   "chol_tg_chunks"  k-chunk sizes of the task-graph factorisation counted back from the pivot as decimal digits (9 = 16 blocks)
 *                     [0 = by size: 112489 = 1, 1, 2, 4, 8, 16, 16, ..; up to 36 blocks 11112489]
 *   "chol_tg_nap"     longest pause of a waiting workgroup between two looks at its dependencies, x 64 clocks: 8, 16, 32, 64, 127 [16]
 *   "chol_tg_grid"    workgroups launched [0 = by size];  "chol_tg_isolate" 1: the critical workgroups keep their CUs to themselves [-1: up to 112 blocks]
 *   "chol_tg_trace"   1: stamp the critical path (gpx_chol_trace); 2: also a per-workgroup task log
 *   "grad_rb_cs"      columns per segment of the register-blocked triangular matvec, a multiple of 128 [0 = default]
 *   "x_rff"           1: the round-3 Thompson sweep kernel instead of the default (process-wide; A/B and witness of the tests)
 *   "x_bg", "x_bg_lds", "x_bg_iters"   a synthetic register-only fp64-MFMA kernel of x_bg workgroups (x_bg_lds KB of LDS each,
 *                     x_bg_iters rounds) runs beside the factorisation (scripts/chol_bg.py)
 *   "x_skip"          leave out the far updates (bit 0), the chain kernels (bit 1) or the near updates (bit 2) of the stream-scheduled
 *                     factorisation to time its parts alone -- the result is then NOT a factorisation
 */
#ifndef GPX_DIAG_H
#define GPX_DIAG_H
#include "../../include/gpx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ("chol_tg_trace" = 1) wall-clock stamps (100 MHz ticks) the task-graph factorisation of the last fit took with its own clock:
 * out[4 p + {0, 1, 2}] = the diagonal workgroup started waiting for / started / finished block p (nP = N/128 rounded up blocks),
 * then out[4 nP + 2 (8 p + i) + {0, 1}] = stamps of the shadows of block row p (i = 0: S1 started waiting for its right-hand sides /
 * has them loaded, 1: .. / its last rows are stored; 2, 3: the same for S2; 4: U started waiting / the tile's earlier chunks are in,
 * 5: the tile is loaded / stored for the diagonal workgroup).  Returns the number of words written (<= n; 20 nP when complete,
 * followed by up to 1024 x 8 per-workgroup counters: tasks, ticks spent taking / updating / solving / publishing, block updates
 * applied, role, exit stamp), 0 without a trace. */
int64_t gpx_chol_trace(gpx_handle *h, int64_t *out, int64_t n);

/* The task lists the task-graph factorisation of an nblocks x nblocks block matrix walks (host only, no device needed: what the CPU
 * tests replay to prove that every tile receives every block row once, in order, and that the lists never dead-lock):
 * counts[2] = entries of list 0 / tasks of the workers' list, out (total, 8) int16 = {type, I, J, k0, k1, ordinal, aux, reserved},
 * the lists back to back.  Types: 1 = panel solve of the 64-column half aux of tile (I, J); 2 = update of tile (I, J) with block rows
 * [k0, k1), its chunk number `ordinal`; 5 = a fused link: the solve of half aux of tile (I, J) and the final chunk [k0, k1 = I+1)
 * (chunk number `reserved`) of the same half of tile (I+1, J); 4 = list 0's descriptor of block row I for the workgroups that follow
 * the diagonal factorisation (they stand for the solves of tiles (I, I+1 .. I+3), the final chunks of tiles (I+1, I+1 .. I+3) =
 * [k0, k1) with chunk number aux, and the diagonal tile's chunk before it, which starts at block row `reserved` if that is >= 0;
 * `ordinal` = chunks of every tile of row I).  The lists are those of a launch with two k-step images of LDS per workgroup (fused
 * links).  chunks as the option "chol_tg_chunks" (<= 0: default).  Returns the total number of entries (written only when
 * cap >= total), -1 on bad arguments.  (Round 5's gpx_chol_tasks had a `split` argument and three counts: renamed, not re-used.) */
int64_t gpx_chol_tasks2(int nblocks, int chunks, int16_t *out, int64_t cap, int64_t *counts);

#ifdef __cplusplus
}
#endif
#endif /* GPX_DIAG_H */
