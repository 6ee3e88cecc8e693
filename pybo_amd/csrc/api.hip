// api.hip -- the C-ABI of libgpx.so (see include/gpx.h for the contract and the pybo call sites each
// entry point stands in for).  Host-side orchestration only: allocation, chunking, launch order,
// HIP-event timers.  No exception leaves this file.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "gpx_internal.h"
#include "gpx_diag.h"

using namespace gpx;

static thread_local std::string g_create_err;

#define HIPCHK(h, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            char buf_[512];                                                                 \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                   \
            (h)->err = buf_;                                                                \
            return (e_ == hipErrorOutOfMemory) ? GPX_EOOM : GPX_EHIP;                       \
        }                                                                                   \
    } while (0)

static int fail(gpx_handle* h, int code, const char* msg) {
    h->err = msg;
    return code;
}

// Every extern "C" body runs inside this guard: nothing the C++ runtime throws (bad_alloc from the
// std::vector / std::string used for staging and messages) may cross the C boundary.
template <typename F>
static int guarded(gpx_handle* h, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        if (h) {
            try { h->err = "out of host memory"; } catch (...) {}
        }
        return GPX_EOOM;
    } catch (...) {
        if (h) {
            try { h->err = "unexpected C++ exception inside libgpx"; } catch (...) {}
        }
        return GPX_EHIP;
    }
}

template <typename T>
static int ensure(gpx_handle* h, T*& p, int64_t& cap, int64_t need) {
    if (need <= cap && p) return GPX_OK;
    if (p) HIPCHK(h, hipFree(p));
    p = nullptr;
    cap = 0;
    HIPCHK(h, hipMalloc((void**)&p, (size_t)need * sizeof(T)));
    cap = need;
    return GPX_OK;
}

// ---- timers -----------------------------------------------------------------------------------
static hipEvent_t ev_get(gpx_handle* h) {
    if (!h->pool.empty()) {
        hipEvent_t e = h->pool.back();
        h->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
// Recycle finished spans without blocking (callers that never read the timers must not leak events).
static void harvest_finished(gpx_handle* h) {
    size_t keep = 0;
    for (size_t i = 0; i < h->pending.size(); ++i) {
        EventPair p = h->pending[i];
        if (hipEventQuery(p.b) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) h->tacc[p.slot] += ms;
            h->pool.push_back(p.a);
            h->pool.push_back(p.b);
        } else {
            h->pending[keep++] = p;
        }
    }
    h->pending.resize(keep);
}

struct Span {
    gpx_handle* h;
    EventPair p;
    Span(gpx_handle* h_, int slot) : h(h_) {
        if (h->pending.size() >= 256) harvest_finished(h);
        p.a = ev_get(h);
        p.b = ev_get(h);
        p.slot = slot;
        hipEventRecord(p.a, h->stream);
    }
    ~Span() {
        hipEventRecord(p.b, h->stream);
        h->pending.push_back(p);
    }
};
static void harvest(gpx_handle* h) {
    hipStreamSynchronize(h->stream);
    if (h->spec_launched && h->stream3) hipStreamSynchronize(h->stream3);   // its timing events live on that stream
    for (auto& p : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) h->tacc[p.slot] += ms;
        h->pool.push_back(p.a);
        h->pool.push_back(p.b);
    }
    h->pending.clear();
}

// ---- lifetime ---------------------------------------------------------------------------------
extern "C" int gpx_version(void) { return 600; }   // round * 100: 300 added gpx_predict_mean, gpx_var_at_obs, gpx_capacity, gpx_append_begin; 400 gpx_chol_trace, gpx_chol_tasks, GPX_OPTIONS; 500: timers slot 16; 510: timers slots 17, 18, options trtri_ahead*, chol_tg_fuse; 600: gpx_diagnostics, gpx_chol_tasks -> gpx_chol_tasks2 (pybo_amd/csrc/gpx_diag.h), the diagnostic options only in a -DGPX_DIAGNOSTICS build, tile_order default by size (7 below 32 block rows, 19 from there on)
extern "C" int gpx_diagnostics(void) {
#ifdef GPX_DIAGNOSTICS
    return 1;
#else
    return 0;
#endif
}

extern "C" const char* gpx_last_error(const gpx_handle* h) {
    return h ? h->err.c_str() : g_create_err.c_str();
}

static int create_impl(int device, void* stream, gpx_handle** out);

extern "C" int gpx_create(int device, void* stream, gpx_handle** out) {
    try {
        return create_impl(device, stream, out);
    } catch (...) {
        if (out) *out = nullptr;
        return GPX_EOOM;
    }
}

static int create_impl(int device, void* stream, gpx_handle** out) {
    if (!out) { g_create_err = "gpx_create: out is NULL"; return GPX_EARG; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("gpx_create: no HIP device (") + hipGetErrorString(e) + ")";
        return GPX_EHIP;
    }
    if (device < 0 || device >= ndev) { g_create_err = "gpx_create: bad device index"; return GPX_EARG; }
    gpx_handle* h = new (std::nothrow) gpx_handle();
    if (!h) { g_create_err = "gpx_create: out of host memory"; return GPX_EOOM; }
    h->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
        delete h;
        return GPX_EHIP;
    }
    if (stream) {
        h->stream = (hipStream_t)stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
            g_create_err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
            delete h;
            return GPX_EHIP;
        }
        h->own_stream = true;
    }
    // (the factorisation's side streams and events are created on first use: gpx::ensure_side_streams)
    // one allocation for the small per-handle device scalars
    if (hipMalloc((void**)&h->dsmall, 64 + 16 * 8 + DMAX * 8 + 32) != hipSuccess) {
        g_create_err = "gpx_create: device allocation failed";
        delete h;
        return GPX_EOOM;
    }
    if (hipHostMalloc((void**)&h->hinv, (size_t)DMAX * 8, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_inv, hipEventDisableTiming) != hipSuccess) {
        g_create_err = "gpx_create: pinned host allocation failed";
        hipFree(h->dsmall);
        delete h;
        return GPX_EOOM;
    }
    h->dflag = reinterpret_cast<int*>(h->dsmall);
    h->dscal = reinterpret_cast<double*>(h->dsmall + 64);
    h->dinvell = h->dscal + 16;
    h->dclk = reinterpret_cast<unsigned long long*>(h->dinvell + DMAX);
    // (on the handle's own stream: the first use of the NULL stream in a process halves the speed of the stream-scheduled
    //  factorisation -- 6.8 -> 12.6 ms at N = 8192, measured in round 5 after a hipMemset here)
    (void)hipMemsetAsync(h->dclk, 0, 32, h->stream);
    (void)hipStreamSynchronize(h->stream);
    *out = h;
    // GPX_OPTIONS="name=value,name=value": options every new handle starts with (A/B runs THROUGH the plug-in layer, whose
    // handles the caller never sees); an unknown name or a bad value fails the creation loudly
    if (const char* env = getenv("GPX_OPTIONS")) {
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string kv = all.substr(pos, end - pos);
            pos = end + 1;
            if (kv.empty()) continue;
            const size_t eq = kv.find('=');
            char* tail = nullptr;
            const long long v = (eq == std::string::npos) ? 0 : strtoll(kv.c_str() + eq + 1, &tail, 10);
            if (eq == std::string::npos || eq == 0 || !tail || *tail != 0 || tail == kv.c_str() + eq + 1 ||
                gpx_set_option(h, kv.substr(0, eq).c_str(), (int64_t)v) != GPX_OK) {
                g_create_err = "gpx_create: GPX_OPTIONS: bad entry '" + kv + "'";
                *out = nullptr;
                gpx_destroy(h);
                return GPX_EARG;
            }
        }
    }
    return GPX_OK;
}

// The blocked factorisation's lookahead streams and events, created when a fit first needs them (more than one
// outer panel): a handle that only ever holds small models -- the members and proposals of the hyper-parameter
// sampler -- never pays for two HSA queues (stream creation dominated gpx_create: ~8 ms per handle).
int gpx::ensure_side_streams(gpx_handle* h) {
    if (h->stream2) return GPX_OK;
    int prio_lo = 0, prio_hi = 0;   // numerically lowest value = highest priority
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // the side stream carries only off-critical-path work (far trailing updates): lowest priority, so the
    // serial chain on the caller's stream wins the dispatcher whenever both have workgroups pending
    bool ok_ev = true;
    for (auto& e : h->ev_row) ok_ev = ok_ev && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok_ev || hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&h->stream4, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_far, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_rest, hipEventDisableTiming) != hipSuccess)
        return fail(h, GPX_EHIP, "side stream / event creation failed");
    return GPX_OK;
}

extern "C" int gpx_destroy(gpx_handle* h) {
    if (!h) return GPX_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (h->ahead_top != 0 && h->stream2) { hipStreamSynchronize(h->stream2); h->ahead_top = 0; }
    tg_free(h);
    for (auto& p : h->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : h->pool) hipEventDestroy(e);
    void* ptrs[] = {h->dXs, h->dXraw, h->dy, h->dS, h->dR, h->dT, h->dU, h->da, h->dalpha, h->dsmall, h->dKs, h->dQp, h->dXc, h->dout, h->dblkv, h->dblki,
                    h->dtopv, h->drff, h->drffs, h->dgrad, h->dens, h->dcZ, h->dcq, h->dbatch, h->dpend, h->drefine, h->dspec};  // dPp, dtopi, dcp alias dQp, dtopv, dcq
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (h->hpin) hipHostFree(h->hpin);
    if (h->hinv) hipHostFree(h->hinv);
    if (h->chol_exec) hipGraphExecDestroy(h->chol_exec);
    if (h->ev_inv) hipEventDestroy(h->ev_inv);
    if (h->ev_spec_go) hipEventDestroy(h->ev_spec_go);
    if (h->ev_spec_done) hipEventDestroy(h->ev_spec_done);
    if (h->stream_bg) { hipStreamSynchronize(h->stream_bg); hipStreamDestroy(h->stream_bg); }
    if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
    if (h->stream3) { hipStreamSynchronize(h->stream3); hipStreamDestroy(h->stream3); }
    if (h->stream4) { hipStreamSynchronize(h->stream4); hipStreamDestroy(h->stream4); }
    if (h->ev_rest) hipEventDestroy(h->ev_rest);
    for (auto e : h->ev_row)
        if (e) hipEventDestroy(e);
    if (h->ev_chain) hipEventDestroy(h->ev_chain);
    if (h->ev_far) hipEventDestroy(h->ev_far);
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
    return GPX_OK;
}

extern "C" int gpx_set_option(gpx_handle* h, const char* name, int64_t value) {
    return guarded(h, [&]() -> int {
        if (!h || !name) return GPX_EARG;
        if (!strcmp(name, "chunk")) {
            if (value < 128 || value % 128) return fail(h, GPX_EARG, "chunk must be a positive multiple of 128");
            h->chunk = value;
            return GPX_OK;
        }
        if (!strcmp(name, "super_m")) {
            if (value != 1 && value != 2 && value != 4 && value != 8 && value != 16)
                return fail(h, GPX_EARG, "super_m must be 1, 2, 4, 8 or 16");
            h->super_m = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "tile_order")) {
            if (value != -1 && (value < 4 || value > 31))
                return fail(h, GPX_EARG, "tile_order: -1 (by size) or bits 0-1 tile map (0..3), bits 2-4 k-loop schedule (1 .. 7)");
            h->tile_order = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "sweep_cache")) {
            if (value < -1 || value > 1) return fail(h, GPX_EARG, "sweep_cache must be 1, 0 or -1");
            h->cache_on = (value == 1);
            if (value < 0) { h->cache_valid = false; h->npend = 0; }
            return GPX_OK;
        }
        if (!strcmp(name, "chol_w")) {
            if (value != 0 && (value < 2 || value > 8)) return fail(h, GPX_EARG, "chol_w must be 0 (by size) or in [2, 8]");
            h->chol_w = (int)value;
            return GPX_OK;
        }
#ifndef GPX_DIAGNOSTICS
        // the shipping library knows the diagnostic knobs by name only: a consumer of include/gpx.h cannot reach a switch that
        // leaves parts of a factorisation out (pybo_amd/csrc/gpx_diag.h; libgpx_diag.so accepts them)
        for (const char* dn : {"chol_tg_chunks", "chol_tg_nap", "chol_tg_grid", "chol_tg_isolate", "chol_tg_trace", "grad_rb_cs", "x_rff",
                               "x_skip", "x_bg", "x_bg_lds", "x_bg_iters"})
            if (!strcmp(name, dn)) return fail(h, GPX_EARG, "diagnostic option: only in a library built with -DGPX_DIAGNOSTICS (libgpx_diag.so)");
#endif
        if (!strcmp(name, "chol_tg") || !strcmp(name, "chol_tg_chunks") ||
            !strcmp(name, "chol_tg_grid") || !strcmp(name, "chol_tg_trace") || !strcmp(name, "chol_tg_tmo_ms") ||
            !strcmp(name, "chol_tg_min") || !strcmp(name, "chol_tg_max") || !strcmp(name, "chol_tg_isolate") ||
            !strcmp(name, "chol_tg_nap") || !strcmp(name, "chol_tg_db") || !strcmp(name, "chol_tg_db_max") ||
            !strcmp(name, "chol_tg_fuse")) {
            if (value < -1 || value > 1000000000) return fail(h, GPX_EARG, "chol_tg*: out of range");
            const char* sub = name + 7;
            if (*sub == 0) { if (value != 0 && value != 1) return fail(h, GPX_EARG, "chol_tg must be 0 or 1"); h->chol_tg = (int)value; }
            else if (!strcmp(sub, "_db")) h->tg_db = (value < 0) ? -1 : (value != 0 ? 1 : 0);
            else if (!strcmp(sub, "_db_max")) h->tg_db_max = (int)std::max<int64_t>(0, value);
            else if (!strcmp(sub, "_nap")) h->tg_nap = (int)std::max<int64_t>(0, std::min<int64_t>(127, value));
            else if (!strcmp(sub, "_chunks")) h->tg_chunks = (int)value;
            else if (!strcmp(sub, "_grid")) h->tg_grid = (int)value;
            else if (!strcmp(sub, "_trace")) h->tg_trace = (int)value;
            else if (!strcmp(sub, "_tmo_ms")) h->tg_tmo_ms = (int)value;
            else if (!strcmp(sub, "_isolate")) h->tg_isolate = (int)value;
            else if (!strcmp(sub, "_fuse")) h->tg_fuse = (value != 0) ? 1 : 0;
            else if (!strcmp(sub, "_max")) h->tg_max = (int)std::max<int64_t>(1, value);
            else h->tg_min = (int)std::max<int64_t>(1, value);
            return GPX_OK;
        }
        if (!strcmp(name, "trtri_ahead") || !strcmp(name, "trtri_ahead_min")) {
            if (value < 0 || value > 1000000) return fail(h, GPX_EARG, "trtri_ahead*: out of range");
            if (name[11] == 0) h->trtri_ahead = (value != 0) ? 1 : 0;
            else h->trtri_ahead_min = (int)std::max<int64_t>(4, value);
            return GPX_OK;
        }
        if (!strcmp(name, "chol_fuse")) {
            if (value != 0 && value != 1) return fail(h, GPX_EARG, "chol_fuse must be 0 or 1");
            h->chol_fuse = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "chol_graph")) {
            if (value != 0 && value != 1) return fail(h, GPX_EARG, "chol_graph must be 0 or 1");
            h->chol_graph = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "chol_merge")) {
            if (value < 0 || value > 100000) return fail(h, GPX_EARG, "chol_merge: 0 (off) or a minimum number of block rows");
            h->chol_merge = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "chol_rl")) {
            if (value != 0 && value != 1) return fail(h, GPX_EARG, "chol_rl must be 0 or 1");
            h->chol_rl = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "x_skip")) {       // diagnostic only: see launch_cholesky
            if (value < 0 || value > 7) return fail(h, GPX_EARG, "x_skip: bits 0..2");
            h->x_skip = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "x_bg") || !strcmp(name, "x_bg_lds") || !strcmp(name, "x_bg_iters")) {   // diagnostic only
            if (value < 0 || value > 10000000) return fail(h, GPX_EARG, "x_bg*: out of range");
            (name[4] == 0 ? h->x_bg : (name[5] == 'l' ? h->x_bg_lds : h->x_bg_iters)) = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "x_rff")) {       // diagnostic: A/B of the Thompson sweep kernels (process-wide)
            if (value < 0 || value > 1) return fail(h, GPX_EARG, "x_rff must be 0 (by size) or 1 (the round-3 kernel)");
            gpx::g_rff_variant = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "grad_form")) {
            if (value < 0 || value > 2) return fail(h, GPX_EARG, "grad_form must be 0 (auto), 1 (two passes) or 2 (one pass)");
            h->grad_form = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "grad_rb_cs")) {
            if (value < 0 || value % 128 || value > 65536) return fail(h, GPX_EARG, "grad_rb_cs must be a multiple of 128 (0 = default)");
            h->grad_rb_cs = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "grad_kernel")) {
            if (value < -1 || value > 1) return fail(h, GPX_EARG, "grad_kernel must be -1 (auto), 0 or 1");
            h->grad_kernel = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "trtri_left")) {
            if (value != 0 && value != 1) return fail(h, GPX_EARG, "trtri_left must be 0 or 1");
            h->trtri_left = (int)value;
            return GPX_OK;
        }
        if (!strcmp(name, "refine_inverse")) {
            if (value != 0 && value != 1) return fail(h, GPX_EARG, "refine_inverse must be 0 or 1");
            h->refine_inverse = (value != 0);
            return GPX_OK;
        }
        if (!strcmp(name, "eager_inverse")) {
            if (value != 0 && value != 1) return fail(h, GPX_EARG, "eager_inverse must be 0 or 1");
            h->eager_inverse = (value != 0);
            return GPX_OK;
        }
        return fail(h, GPX_EARG, "unknown option");
    });
}

extern "C" int gpx_sync(gpx_handle* h) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return GPX_OK;
    });
}

extern "C" int64_t gpx_chol_tasks2(int nblocks, int chunks, int16_t* out, int64_t cap, int64_t* counts) {
    if (!counts) return -1;
    try {
        return gpx::tg_tasks_copy(nblocks, chunks, out, cap, counts);
    } catch (...) {
        return -1;
    }
}

extern "C" int64_t gpx_chol_trace(gpx_handle* h, int64_t* out, int64_t n) {
    if (!h || !out || n <= 0) return 0;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return 0;
    static_assert(sizeof(long long) == sizeof(int64_t), "trace words");
    return tg_trace_copy(h, reinterpret_cast<long long*>(out), n);
}

extern "C" int gpx_timers(gpx_handle* h, double* out, int n, int reset) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!out || n < 0) return fail(h, GPX_EARG, "timers: NULL output");
        harvest(h);
        unsigned long long clk[4] = {0, 0, 0, 0};
        (void)hipMemcpyAsync(clk, h->dclk, 32, hipMemcpyDeviceToHost, h->stream);
        (void)hipStreamSynchronize(h->stream);
        h->tacc[T_SCLK] = clk[1] ? 100.0 * (double)clk[0] / (double)clk[1] : 0.0;    // MHz
        h->tacc[T_RFFCLK] = clk[3] ? 100.0 * (double)clk[2] / (double)clk[3] : 0.0;
        const int m = std::min(n, (int)T_COUNT);
        for (int i = 0; i < m; ++i) out[i] = h->tacc[i];
        if (reset) {
            for (int i = 0; i < T_COUNT; ++i) h->tacc[i] = 0;
            (void)hipMemsetAsync(h->dclk, 0, 32, h->stream);
        }
        return m;
    });
}

// An announcement's correction pass may still be reading Xs / the cached candidates on the third stream: whoever is
// about to rewrite those (a refit, a new full sweep with the cache on, the next announcement) orders itself after it.
static void spec_cancel(gpx_handle* h) {
    if (h->spec_launched && h->ev_spec_done) hipStreamWaitEvent(h->stream, h->ev_spec_done, 0);
    h->spec_active = false;
    h->spec_launched = false;
    h->apply_pending = false;        // (callers either flushed first or are discarding the cache)
}

// ---- fit --------------------------------------------------------------------------------------
static int check_fit_args(gpx_handle* h, const void* X, int64_t N, int64_t d, const void* y, int kid,
                          const double* ell, double rho, double sn2) {
    if (!h) return GPX_EARG;
    if (!X || !y || !ell) return fail(h, GPX_EARG, "fit: NULL pointer");
    if (N < 1) return fail(h, GPX_EARG, "fit: N must be >= 1");
    if (d < 1 || d > DMAX) return fail(h, GPX_EARG, "fit: d must be in [1, 1024]");
    if (kid < GPX_KERN_SE_ARD || kid > GPX_KERN_MATERN12) return fail(h, GPX_EARG, "fit: unknown kernel id");
    if (!(rho > 0) || !(sn2 >= 0) || !std::isfinite(rho) || !std::isfinite(sn2))
        return fail(h, GPX_EARG, "fit: need finite rho > 0 and sn2 >= 0");
    for (int64_t k = 0; k < d; ++k)
        if (!(ell[k] > 0) || !std::isfinite(ell[k]))
            return fail(h, GPX_EARG, "fit: length-scales must be positive and finite");
    return GPX_OK;
}

static int alloc_model(gpx_handle* h, int64_t Np, int64_t d) {
    if (Np > h->cap_np || d > h->cap_d) {
        // head-room of at least one block (1/16 of the size beyond that): gpx_append can then grow the factor across
        // a 128-block boundary by re-striding inside the existing buffers, without an allocation
        Np = std::max(Np + std::max<int64_t>(NB, Np / 16 / NB * NB), h->cap_np);
        d = std::max(d, h->cap_d);
        double** mats[] = {&h->dS, &h->dR, &h->dT, &h->dU};
        for (auto m : mats) {
            if (*m) HIPCHK(h, hipFree(*m));
            *m = nullptr;
        }
        double** vecs[] = {&h->dy, &h->da, &h->dalpha, &h->dXs, &h->dXraw};
        for (auto v : vecs) {
            if (*v) HIPCHK(h, hipFree(*v));
            *v = nullptr;
        }
        h->cap_np = 0;
        h->cap_d = 0;
        for (auto m : mats) {
            HIPCHK(h, hipMalloc((void**)m, (size_t)Np * Np * 8));
            // never-written regions (e.g. the upper off-diagonal blocks of T) are never read by a
            // kernel either, but keep them defined for introspection
            HIPCHK(h, hipMemsetAsync(*m, 0, (size_t)Np * Np * 8, h->stream));
        }
        HIPCHK(h, hipMalloc((void**)&h->dy, (size_t)Np * 8));
        HIPCHK(h, hipMalloc((void**)&h->da, (size_t)Np * 8));
        HIPCHK(h, hipMalloc((void**)&h->dalpha, (size_t)Np * 8));
        HIPCHK(h, hipMalloc((void**)&h->dXs, (size_t)Np * d * 8));
        HIPCHK(h, hipMalloc((void**)&h->dXraw, (size_t)Np * d * 8));
        h->cap_np = Np;
        h->cap_d = d;
    }
    return GPX_OK;
}

// The triangular inverse T = R^-T (and U, a, alpha) is formed on FIRST USE, not by the fit: the Thompson path
// (gpx_rff_gram / gpx_rff_sweep, pybo/policies/simple.py:44-48) never reads it, and it is a quarter of a fit
// at N = 16384.  Every entry point that reads T, U, a or alpha calls this first.
// the side stream's share of an inversion that nobody asked for in the end (a new fit, a teardown): wait for it and forget it
static void settle_ahead(gpx_handle* h) {
    if (h->ahead_top == 0) return;
    (void)hipStreamSynchronize(h->stream2);
    h->ahead_top = 0;
}

int gpx::ensure_inverse(gpx_handle* h) {
    if (h->stage >= 3) return GPX_OK;
    if (h->stage < 2) return fail(h, GPX_ESTATE, "model is not fitted");
    HIPCHK(h, hipSetDevice(h->device));
    {
        Span sp(h, T_TRTRI);
        launch_trtri(h);
        if (h->refine_inverse) {
            int rc = ensure(h, h->drefine, h->cap_refine, h->cap_np * h->cap_np);
            if (rc) return rc;
            launch_refine_inverse(h, h->drefine);
        }
    }
    {
        Span sp(h, T_ALPHA);
        launch_alpha(h);
    }
    HIPCHK(h, hipGetLastError());
    h->stage = 3;
    return GPX_OK;
}

// core: dX (N,d), dy (N,) device pointers; stage 1 = stop after Gram, 2 = after Cholesky (what gpx_fit does: the
// model counts as fitted, the inverse follows lazily), 3 = Cholesky + inverse eagerly
static int fit_core(gpx_handle* h, const double* dX, int64_t N, int64_t d, const double* dy, int kid,
                    const double* ell, double rho, double sn2, double bias, int stage) {
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t Np = (N + NB - 1) / NB * NB;
    spec_cancel(h);              // an announced observation belongs to the model that is being replaced
    settle_ahead(h);             // ... and so does an inversion still running ahead on the side stream
    const bool prev_inverse_used = h->stage >= 3;
    ++h->gen;
    h->fitted = false;
    h->stage = 0;
    h->fail_pivot = -1;
    h->last_topn = 0;
    h->cache_valid = false;      // new data / hyper-parameters: the cached sums describe another posterior
    h->npend = 0;
    int rc = alloc_model(h, Np, d);
    if (rc) return rc;
    h->N = N; h->Np = Np; h->d = d; h->kernel_id = kid;
    h->rho = rho; h->sn2 = sn2; h->bias = bias;
    h->ell.assign(ell, ell + d);
    // 1/ell goes through a pinned buffer of the handle (round 2 synchronised the stream here for a stack buffer);
    // the previous fit's copy has long completed -- the event wait returns at once
    if (h->inv_inflight) { HIPCHK(h, hipEventSynchronize(h->ev_inv)); h->inv_inflight = false; }
    for (int64_t k = 0; k < d; ++k) h->hinv[k] = 1.0 / ell[k];
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->dinvell, h->hinv, (size_t)d * 8, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipEventRecord(h->ev_inv, s));
    h->inv_inflight = true;
    HIPCHK(h, hipMemcpyAsync(h->dXraw, dX, (size_t)N * d * 8, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipMemsetAsync(h->dy, 0, (size_t)Np * 8, s));
    HIPCHK(h, hipMemcpyAsync(h->dy, dy, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
    {
        Span sp(h, T_GRAM);
        launch_scale_x(s, h->dXraw, N, Np, (int)d, h->dinvell, h->dXs);
        launch_gram_sym(s, h->dXs, N, Np, (int)d, kid, rho, sn2, h->dS);
    }
    h->stage = 1;
    if (stage >= 2) {
        {
            Span sp(h, T_CHOL);
            // the persistent task-graph kernel (kernels_chol_tg.hip) unless switched off, diagnosed in parts, or too small
            h->tg_launched = false;
            // the side streams of a model this size (the stream schedule's lookahead; the announced-observation pass of the
            // warm loop runs on the third one) are created HERE, in the cold fit, whichever factorisation runs: created
            // on first use by gpx_append_begin they cost the first warm iteration ~18 ms (three HSA queues)
            if (h->Np / NB > (h->chol_w ? h->chol_w : 4) && (rc = ensure_side_streams(h))) return rc;
            // the inverse's leading part may ride behind the factorisation when the inverse is certain to follow (stage 3,
            // eager_inverse) or likely to (the previous model's was formed: an acquisition loop refits and sweeps)
            h->want_ahead = (stage >= 3 || h->eager_inverse || prev_inverse_used) && !h->refine_inverse;
            const bool tg = h->chol_tg && h->x_skip == 0 && h->x_bg <= 0 && h->Np / NB >= h->tg_min && h->Np / NB <= h->tg_max && launch_cholesky_tg(h);
            if (!tg) launch_cholesky(h);
        }
        int flag2[2] = {0, 0};       // [0] failing pivot + 1, [1] the task-graph kernel's "a spin gave up" (2): one copy for both
        HIPCHK(h, hipMemcpyAsync(flag2, h->dflag, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        int flag = flag2[0];
        if (h->ahead_top != 0 && flag != 0) {      // not positive definite: what the side stream is doing is of no use
            (void)hipStreamSynchronize(h->stream2);
            h->ahead_top = 0;
        }
        if (h->tg_launched && flag == 0 && flag2[1] == 2) {
            // a spin of the persistent kernel gave up (the device is shared with something that kept its workgroups from
            // becoming resident): S is half-consumed -- rebuild it and run the stream schedule.  Loud, and counted.
            ++h->tg_fallbacks;
            h->tacc[T_TGFALL] += 1.0;
            fprintf(stderr, "libgpx: the task-graph factorisation gave up waiting (N = %lld); re-running the stream schedule\n",
                    (long long)N);
            h->tg_launched = false;
            if (h->ahead_top != 0) { (void)hipStreamSynchronize(h->stream2); h->ahead_top = 0; }
            launch_gram_sym(s, h->dXs, N, Np, (int)d, kid, rho, sn2, h->dS);
            if (h->Np / NB > (h->chol_w ? h->chol_w : 4) && (rc = ensure_side_streams(h))) return rc;
            launch_cholesky(h);
            HIPCHK(h, hipMemcpyAsync(&flag, h->dflag, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
        }
        if (flag != 0) {
            h->fail_pivot = (int64_t)flag - 1;
            char buf[160];
            snprintf(buf, sizeof buf, "fit: K + sn2*I is not positive definite (pivot %lld)",
                     (long long)h->fail_pivot);
            return fail(h, GPX_ENOTPD, buf);
        }
        h->stage = 2;
        h->fitted = true;
    }
    HIPCHK(h, hipGetLastError());
    if (stage >= 3 || h->eager_inverse) return (h->stage >= 2) ? ensure_inverse(h) : GPX_OK;
    return GPX_OK;
}

extern "C" int gpx_fit_dev(gpx_handle* h, const double* dX, int64_t N, int64_t d, const double* dy,
                           int kernel_id, const double* ell, double rho, double sn2, double bias) {
    return guarded(h, [&]() -> int {
        int rc = check_fit_args(h, dX, N, d, dy, kernel_id, ell, rho, sn2);
        if (rc) return rc;
        return fit_core(h, dX, N, d, dy, kernel_id, ell, rho, sn2, bias, 2);
    });
}

static int fit_host(gpx_handle* h, const double* X, int64_t N, int64_t d, const double* y, int kid,
                    const double* ell, double rho, double sn2, double bias, int stage) {
    int rc = check_fit_args(h, X, N, d, y, kid, ell, rho, sn2);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    rc = ensure(h, h->dXc, h->cap_xc, N * d + N);
    if (rc) return rc;
    {
        Span sp(h, T_COPY);
        HIPCHK(h, hipMemcpyAsync(h->dXc, X, (size_t)N * d * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->dXc + N * d, y, (size_t)N * 8, hipMemcpyHostToDevice, h->stream));
    }
    return fit_core(h, h->dXc, N, d, h->dXc + N * d, kid, ell, rho, sn2, bias, stage);
}

extern "C" int gpx_fit(gpx_handle* h, const double* X, int64_t N, int64_t d, const double* y, int kernel_id,
                       const double* ell, double rho, double sn2, double bias) {
    return guarded(h, [&]() -> int {
        return fit_host(h, X, N, d, y, kernel_id, ell, rho, sn2, bias, 2);
    });
}

extern "C" int gpx_fit_stage(gpx_handle* h, const double* X, int64_t N, int64_t d, const double* y,
                             int kernel_id, const double* ell, double rho, double sn2, double bias,
                             int stage) {
    return guarded(h, [&]() -> int {
        if (stage < 1 || stage > 3) return h ? fail(h, GPX_EARG, "fit_stage: stage must be 1..3") : GPX_EARG;
        return fit_host(h, X, N, d, y, kernel_id, ell, rho, sn2, bias, stage);
    });
}

extern "C" int gpx_loglik(gpx_handle* h, double* out) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        return gpx::loglik_host(h, out);
    });
}

// Apply the cache corrections of the observations appended since the last flush: ONE pass over the candidates
// (kernels_sweep.hip: k_sweep_rankq).
static int flush_pending(gpx_handle* h) {
    if (h->apply_pending) {
        // an announced observation was appended: its row of V over the cached candidates was computed on the third
        // stream; fold it into the sums now that somebody needs them (not earlier: the caller's stream stays free for
        // the recommender's gradient calls while the pass finishes)
        h->apply_pending = false;
        if (h->cache_valid && h->spec_M == h->cache_M) {
            const SpecBuf sb = spec_layout(h);
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_spec_done, 0));
            h->spec_launched = false;
            Span sp(h, T_RANK1);
            launch_cache_apply(h->stream, sb.v, sb.scal, h->cache_M, h->dcq, h->dcp);
        }
    }
    if (h->npend == 0) return GPX_OK;
    if (h->cache_valid) {
        Span sp(h, T_RANK1);
        launch_sweep_rankq(h->stream, h->dXs, h->N, (int)h->d, h->dpend, h->pend_ld, h->npend,
                           h->dpend + (int64_t)PEND_MAX * h->pend_ld, h->dcZ, h->cache_M, h->dinvell, h->kernel_id,
                           h->rho, h->dcq, h->dcp);
        HIPCHK(h, hipGetLastError());
    }
    h->npend = 0;
    return GPX_OK;
}

extern "C" int gpx_loglik_batch(gpx_handle* h, int64_t B, const double* hypers, double* out) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        return gpx::loglik_batch_host(h, B, hypers, out);
    });
}

extern "C" int gpx_append_begin(gpx_handle* h, const double* x) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!x) return fail(h, GPX_EARG, "append_begin: NULL point");
        if (!h->fitted) return fail(h, GPX_ESTATE, "append_begin: model is not fitted");
        if (!h->cache_valid) return fail(h, GPX_ESTATE, "append_begin: no live sweep cache (nothing to run ahead)");
        HIPCHK(h, hipSetDevice(h->device));
        int rc;
        if ((rc = ensure_inverse(h))) return rc;
        if ((rc = flush_pending(h))) return rc;                 // earlier appends' corrections: apply them now
        // padding of the last 128-block used up (N = 8192 exactly: the benchmark's first warm iteration): add the block now,
        // as gpx_append would have -- round 3 declined here and that iteration ran its correction pass unhidden (+4.5 ms)
        if ((rc = ensure_side_streams(h))) return rc;
        if (!h->ev_spec_go &&
            (hipEventCreateWithFlags(&h->ev_spec_go, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&h->ev_spec_done, hipEventDisableTiming) != hipSuccess))
            return fail(h, GPX_EHIP, "append_begin: event creation failed");
        spec_cancel(h);                                          // a previous announcement's pass has ended
        if (h->ev_spec_go) HIPCHK(h, hipEventSynchronize(h->ev_spec_go));   // ... and its copy of spec_x too
        // (only now: growing re-strides or re-allocates S / R / T / U and Xs on the main stream, which the previous
        //  announcement's pass on the third stream may still have been reading -- ADVICE round 4)
        if ((rc = grow_factor_if_full(h))) return rc;
        const int64_t d = h->d, M = h->cache_M, ld = h->cap_np;
        const int64_t xpad = (h->cap_d + 63) / 64 * 64;
        const int64_t need = 2 * xpad + 5 * ld + 64 + M;
        if ((rc = ensure(h, h->dspec, h->cap_spec, need))) return rc;
        h->spec_ld = ld;
        h->spec_M = M;
        const SpecBuf sb = spec_layout(h);
        h->spec_x.assign(x, x + d);
        hipStream_t s = h->stream, s3 = h->stream3;
        int* flag = h->dflag + 12;                               // its own word; a bad pivot surfaces at gpx_append
        HIPCHK(h, hipMemsetAsync(flag, 0, sizeof(int), s));
        HIPCHK(h, hipMemcpyAsync(sb.x, h->spec_x.data(), (size_t)d * 8, hipMemcpyHostToDevice, s));
        launch_append_prepare(h, s, sb.x, sb.ks, sb.g, sb.r, sb.tu, 0.0, sb.scal, flag);
        launch_scale_point(s, sb.x, h->dinvell, (int)d, sb.xs);
        launch_pend_store(s, sb.tu, h->N, ld, sb.scal, sb.row, sb.pscal);
        HIPCHK(h, hipEventRecord(h->ev_spec_go, s));
        HIPCHK(h, hipStreamWaitEvent(s3, h->ev_spec_go, 0));
        {
            EventPair p;                                          // T_RANK1, timed on the stream the pass runs on
            if (h->pending.size() >= 256) harvest_finished(h);
            p.a = ev_get(h);
            p.b = ev_get(h);
            p.slot = T_RANK1;
            hipEventRecord(p.a, s3);
            launch_sweep_rank1_v(s3, h->dXs, h->N + 1, (int)d, sb.row, ld, sb.pscal, h->dcZ, M, h->dinvell, h->kernel_id,
                                 h->rho, sb.xs, sb.v);
            hipEventRecord(p.b, s3);
            h->pending.push_back(p);
        }
        HIPCHK(h, hipEventRecord(h->ev_spec_done, s3));
        HIPCHK(h, hipGetLastError());
        h->spec_active = true;
        h->spec_launched = true;
        h->spec_gen = h->gen;
        return GPX_OK;
    });
}

extern "C" int gpx_append(gpx_handle* h, const double* x, double y) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        const int64_t Nold = h->N;
        int rc;
        {
            Span sp(h, T_APPEND);
            rc = gpx::append_host(h, x, y);
        }
        if (rc != GPX_OK) {
            if (rc != GPX_EARG && rc != GPX_ESTATE) { h->cache_valid = false; h->npend = 0; }
            return rc;
        }
        ++h->gen;
        if (h->cache_valid && h->spec_used && h->spec_M == h->cache_M) {
            // the point was announced: its row of V over the cached candidates is ready (or about to be) on the third
            // stream; what is left is q += v^2, p += v a_new -- O(M)
            const SpecBuf sb = spec_layout(h);
            // keep {d, 1/d, a_new, d^2} beside v (the handle's scalar scratch is rewritten by the next call); the sums
            // are updated by flush_pending when they are next read
            HIPCHK(h, hipMemcpyAsync(sb.scal, h->dscal, 4 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            h->apply_pending = true;
            return GPX_OK;
        }
        if (h->cache_valid) {
            // keep the cached per-candidate sums current: the correction for this observation (one N*M pass
            // instead of the next N^2*M sweep) is queued and applied together with up to PEND_MAX - 1 others --
            // at the next gpx_sweep_update, or here when the queue is full.  (scal / w are the device-side
            // results of the append just done.)
            if (h->npend > 0 && h->N > h->pend_ld) {             // the factor outgrew the rows: apply what is queued
                h->N = Nold;                                     // (the queued rows know nothing of the new point)
                const int rc2 = flush_pending(h);
                h->N = Nold + 1;
                if (rc2) return rc2;
            }
            if (h->npend == 0 && h->cap_np > h->pend_ld) {
                if (h->dpend) HIPCHK(h, hipFree(h->dpend));
                h->dpend = nullptr;
                h->pend_ld = 0;
                HIPCHK(h, hipMalloc((void**)&h->dpend, (size_t)PEND_MAX * (h->cap_np + 2) * 8));
                h->pend_ld = h->cap_np;
            }
            launch_pend_store(h->stream, h->app_w, Nold, h->pend_ld, h->dscal, h->dpend + (int64_t)h->npend * h->pend_ld,
                              h->dpend + (int64_t)PEND_MAX * h->pend_ld + 2 * h->npend);
            HIPCHK(h, hipGetLastError());
            if (++h->npend == PEND_MAX) return flush_pending(h);
        }
        return GPX_OK;
    });
}

extern "C" int64_t gpx_fail_pivot(const gpx_handle* h) { return h ? h->fail_pivot : -1; }

extern "C" int gpx_get_matrix(gpx_handle* h, int which, double* out) {
    return guarded(h, [&]() -> int {
        if (!h || !out) return GPX_EARG;
        if (which < 0 || which > 2) return fail(h, GPX_EARG, "get_matrix: which must be 0..2");
        if ((which == 2 && h->stage != 1) || (which != 2 && h->stage < 2))
            return fail(h, GPX_ESTATE, "get_matrix: the requested matrix is not available at this stage");
        HIPCHK(h, hipSetDevice(h->device));
        if (which == 1) {
            int rc0 = ensure_inverse(h);
            if (rc0) return rc0;
        }
        const int64_t N = h->N, Np = h->Np;
        int rc = ensure(h, h->dout, h->cap_out, N * N);
        if (rc) return rc;
        if (which == 0) {
            launch_transpose_lower(h->stream, h->dR, Np, h->dout, N);  // L = R^T
            HIPCHK(h, hipMemcpyAsync(out, h->dout, (size_t)N * N * 8, hipMemcpyDeviceToHost, h->stream));
        } else {
            const double* src = (which == 1) ? h->dT : h->dS;
            HIPCHK(h, hipMemcpy2DAsync(out, (size_t)N * 8, src, (size_t)Np * 8, (size_t)N * 8, (size_t)N,
                                       hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (which == 2) {  // only the upper 128-block triangle was built: blank the rest
            for (int64_t i = 0; i < N; ++i)
                for (int64_t j = 0; j < i; ++j) out[i * N + j] = 0.0;
        }
        if (which == 1) {  // T is lower triangular; blocks above the diagonal are never written
            for (int64_t i = 0; i < N; ++i)
                for (int64_t j = i + 1; j < N; ++j) out[i * N + j] = 0.0;
        }
        return GPX_OK;
    });
}

extern "C" int gpx_get_vectors(gpx_handle* h, double* a, double* alpha) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!h->fitted) return fail(h, GPX_ESTATE, "get_vectors: model is not fitted");
        HIPCHK(h, hipSetDevice(h->device));
        {
            int rc0 = ensure_inverse(h);
            if (rc0) return rc0;
        }
        if (a) HIPCHK(h, hipMemcpyAsync(a, h->da, (size_t)h->N * 8, hipMemcpyDeviceToHost, h->stream));
        if (alpha) HIPCHK(h, hipMemcpyAsync(alpha, h->dalpha, (size_t)h->N * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return GPX_OK;
    });
}

extern "C" int gpx_mean_at_obs(gpx_handle* h, double* mu_host, double* mu_max) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!h->fitted) return fail(h, GPX_ESTATE, "mean_at_obs: model is not fitted");
        HIPCHK(h, hipSetDevice(h->device));
        {
            int rc0 = ensure_inverse(h);
            if (rc0) return rc0;
        }
        const int64_t N = h->N;
        std::vector<double> al((size_t)N), yy((size_t)N);
        HIPCHK(h, hipMemcpyAsync(al.data(), h->dalpha, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(yy.data(), h->dy, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // latent posterior mean at the observed inputs: K alpha + bias = y - sn2 * alpha
        double mx = -HUGE_VAL;
        for (int64_t i = 0; i < N; ++i) {
            const double m = yy[i] - h->sn2 * al[i];
            if (mu_host) mu_host[i] = m;
            if (m > mx) mx = m;
        }
        if (mu_max) *mu_max = mx;
        return GPX_OK;
    });
}

extern "C" int gpx_var_at_obs(gpx_handle* h, double* s2_host) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!s2_host) return fail(h, GPX_EARG, "var_at_obs: NULL output");
        if (!h->fitted) return fail(h, GPX_ESTATE, "var_at_obs: model is not fitted");
        HIPCHK(h, hipSetDevice(h->device));
        int rc;
        if ((rc = ensure_inverse(h))) return rc;
        const int64_t N = h->N;
        if ((rc = ensure(h, h->dout, h->cap_out, N))) return rc;
        launch_kinv_diag(h, h->dout);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(s2_host, h->dout, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // s2_i = sn2 - sn2^2 [K^-1]_ii  (latent variance at observed input i), clipped like the sweep's
        for (int64_t i = 0; i < N; ++i) {
            const double v = h->sn2 * (1.0 - h->sn2 * s2_host[i]);
            s2_host[i] = v > 1e-100 ? v : 1e-100;
        }
        return GPX_OK;
    });
}

extern "C" int64_t gpx_capacity(const gpx_handle* h) { return h ? h->cap_np : 0; }

// ---- sweep ------------------------------------------------------------------------------------
// top-k of a device vector (value descending, index ascending, NaN last) -> host buffers; synchronises.
static int topk_core(gpx_handle* h, const double* d_vals, int64_t M, int64_t k, double* top_val,
                     int64_t* top_idx) {
    int rc;
    hipStream_t s = h->stream;
    const int64_t nblk = topk_blocks(M);
    if ((rc = ensure(h, h->dblkv, h->cap_blk, nblk * k))) return rc;
    if ((rc = ensure(h, h->dblki, h->cap_blki, nblk * k))) return rc;
    if ((rc = ensure(h, h->dtopv, h->cap_top, (int64_t)TOPK_MAX * 2))) return rc;
    h->dtopi = reinterpret_cast<int64_t*>(h->dtopv + TOPK_MAX);
    {
        Span sp(h, T_ACQ);
        launch_topk(s, d_vals, M, (int)k, h->dblkv, h->dblki, nblk, h->dtopv, h->dtopi);
    }
    h->last_topv = h->dtopv; h->last_topi = h->dtopi; h->last_topn = k;
    HIPCHK(h, hipMemcpyAsync(top_val, h->dtopv, (size_t)k * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(top_idx, h->dtopi, (size_t)k * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return GPX_OK;
}

static int sweep_core(gpx_handle* h, int acq_id, const double* params, int nparams, const double* dXc,
                      int64_t M, int64_t k, double* top_val, int64_t* top_idx, double* d_acq,
                      double* d_mu, double* d_s2) {
    if (!h->fitted) return fail(h, GPX_ESTATE, "sweep: model is not fitted");
    if (!dXc || M < 1) return fail(h, GPX_EARG, "sweep: need M >= 1 candidates");
    if (acq_id < GPX_ACQ_EI || acq_id > GPX_ACQ_MEAN) return fail(h, GPX_EARG, "sweep: unknown acquisition id");
    if (acq_id != GPX_ACQ_MEAN && (nparams < 1 || !params)) return fail(h, GPX_EARG, "sweep: missing acquisition parameter");
    if (k < 0 || k > TOPK_MAX) return fail(h, GPX_EARG, "sweep: k must be in [0, 4096]");
    if (k > 0 && (!top_val || !top_idx)) return fail(h, GPX_EARG, "sweep: NULL top-k output");
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_inverse(h))) return rc;
    hipStream_t s = h->stream;
    const int64_t Np = h->Np;
    const int nP = (int)(Np / NB);
    const int64_t chunk_opt = h->chunk > 0 ? h->chunk : (Np <= 4096 ? 131072 : 65536);
    const int64_t chunk = std::min<int64_t>(chunk_opt, (M + TBH - 1) / TBH * TBH);
    if ((rc = ensure(h, h->dKs, h->cap_ks, Np * chunk))) return rc;
    if ((rc = ensure(h, h->dQp, h->cap_part, (int64_t)nP * chunk * 2))) return rc;
    h->dPp = h->dQp + (int64_t)nP * chunk;
    if (!d_acq) {
        if ((rc = ensure(h, h->dout, h->cap_out, M))) return rc;
        d_acq = h->dout;
    }
    const double p0 = (acq_id == GPX_ACQ_MEAN) ? 0.0 : params[0];
    double* cq = nullptr;
    double* cp = nullptr;
    if (h->cache_on) {
        spec_cancel(h);          // (the candidates it reads are about to be replaced)
        h->cache_valid = false;
        h->npend = 0;
        if ((rc = ensure(h, h->dcZ, h->cap_cz, M * h->d))) return rc;
        if ((rc = ensure(h, h->dcq, h->cap_cq, 2 * M))) return rc;
        h->dcp = h->dcq + h->cap_cq / 2;
        cq = h->dcq;
        cp = h->dcp;
        if (dXc != h->dcZ)
            HIPCHK(h, hipMemcpyAsync(h->dcZ, dXc, (size_t)M * h->d * 8, hipMemcpyDeviceToDevice, s));
    }

    for (int64_t m0 = 0; m0 < M; m0 += chunk) {
        const int64_t valid = std::min(chunk, M - m0);
        const int64_t cols = (valid + TBH - 1) / TBH * TBH;
        {
            Span sp(h, T_XGRAM);
            launch_cross_gram(s, h->dXs, Np, h->N, (int)h->d, dXc, m0, M, cols, h->dinvell, h->kernel_id,
                              h->rho, h->dKs, Np);
        }
        {
            Span sp(h, T_TRMM);
            launch_sweep_trmm(s, h->dU, Np, h->dKs, chunk, cols, h->da, h->dQp, h->dPp, chunk,
                              // by size: short tiles (fewer than 32 block rows) on the barrier-free loop, long ones on the shared-image loop
                              // (crossover measured: profiles/r06_sweep_power_probes.txt, section 6)
                              h->tile_order >= 0 ? h->tile_order : (Np / NB < 32 ? 7 : 19), h->super_m, h->dclk);
        }
        h->tacc[T_NLAUNCH] += 1.0;
        // ALGORITHMIC work of this launch: N^2 flop per candidate (N^2/2 multiply-adds of the
        // triangular product).  The kernel executes Np*(Np+128) per padded column (identity padding and full
        // diagonal blocks): 1.6 % more at N = 8192 -- not counted.
        h->tacc[T_FLOP] += (double)h->N * (double)h->N * (double)valid;
        {
            Span sp(h, T_ACQ);
            launch_acq(s, h->dQp, h->dPp, chunk, nP, m0, valid, h->rho, h->bias, acq_id, p0, d_acq, d_mu,
                       d_s2, cq, cp);
        }
    }
    if (h->cache_on) {
        h->cache_valid = true;
        h->cache_M = M;
    }
    if (k > 0 && (rc = topk_core(h, d_acq, M, k, top_val, top_idx))) return rc;
    HIPCHK(h, hipGetLastError());
    return GPX_OK;
}

extern "C" int gpx_sweep_dev(gpx_handle* h, int acq_id, const double* params, int nparams, const double* dXc,
                             int64_t M, int64_t k, double* top_val, int64_t* top_idx, double* d_acq_all,
                             double* d_mu, double* d_s2) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        return sweep_core(h, acq_id, params, nparams, dXc, M, k, top_val, top_idx, d_acq_all, d_mu, d_s2);
    });
}

extern "C" int gpx_sweep(gpx_handle* h, int acq_id, const double* params, int nparams, const double* Xc,
                         int64_t M, int64_t k, double* top_val, int64_t* top_idx, double* acq_all, double* mu,
                         double* s2) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!h->fitted) return fail(h, GPX_ESTATE, "sweep: model is not fitted");
        if (!Xc || M < 1) return fail(h, GPX_EARG, "sweep: need M >= 1 candidates");
        HIPCHK(h, hipSetDevice(h->device));
        int rc;
        const int64_t d = h->d;
        // staging: [Xc (M*d)] [acq M] [mu M] [s2 M]
        if ((rc = ensure(h, h->dXc, h->cap_xc, M * d + 3 * M))) return rc;
        double* dX = h->dXc;
        double* dacq = dX + M * d;
        double* dmu = dacq + M;
        double* ds2 = dmu + M;
        {
            Span sp(h, T_COPY);
            HIPCHK(h, hipMemcpyAsync(dX, Xc, (size_t)M * d * 8, hipMemcpyHostToDevice, h->stream));
        }
        rc = sweep_core(h, acq_id, params, nparams, dX, M, k, top_val, top_idx, dacq, mu ? dmu : nullptr,
                        s2 ? ds2 : nullptr);
        if (rc) return rc;
        {
            Span sp(h, T_COPY);
            if (acq_all) HIPCHK(h, hipMemcpyAsync(acq_all, dacq, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
            if (mu) HIPCHK(h, hipMemcpyAsync(mu, dmu, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
            if (s2) HIPCHK(h, hipMemcpyAsync(s2, ds2, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return GPX_OK;
    });
}

// Re-score the cached candidate set: O(M).  The sums were produced by the last full sweep with option
// "sweep_cache" = 1 and have been kept current by every gpx_append since.
static int sweep_update_core(gpx_handle* h, int acq_id, const double* params, int nparams, int64_t k,
                             double* top_val, int64_t* top_idx, double* d_acq, double* d_mu, double* d_s2) {
    if (!h->fitted) return fail(h, GPX_ESTATE, "sweep_update: model is not fitted");
    if (!h->cache_valid)
        return fail(h, GPX_ESTATE, "sweep_update: no valid sweep cache (set option sweep_cache = 1 and run a full "
                                   "sweep; a refit invalidates it)");
    if (acq_id < GPX_ACQ_EI || acq_id > GPX_ACQ_MEAN) return fail(h, GPX_EARG, "sweep_update: unknown acquisition id");
    if (acq_id != GPX_ACQ_MEAN && (nparams < 1 || !params)) return fail(h, GPX_EARG, "sweep_update: missing acquisition parameter");
    if (k < 0 || k > TOPK_MAX) return fail(h, GPX_EARG, "sweep_update: k must be in [0, 4096]");
    if (k > 0 && (!top_val || !top_idx)) return fail(h, GPX_EARG, "sweep_update: NULL top-k output");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t M = h->cache_M;
    int rc;
    if ((rc = flush_pending(h))) return rc;
    if (!d_acq) {
        if ((rc = ensure(h, h->dout, h->cap_out, M))) return rc;
        d_acq = h->dout;
    }
    const double p0 = (acq_id == GPX_ACQ_MEAN) ? 0.0 : params[0];
    {
        Span sp(h, T_ACQ);
        launch_acq(h->stream, h->dcq, h->dcp, 0, 0, 0, M, h->rho, h->bias, acq_id, p0, d_acq, d_mu, d_s2, nullptr,
                   nullptr);
    }
    if (k > 0 && (rc = topk_core(h, d_acq, M, k, top_val, top_idx))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return GPX_OK;
}

extern "C" int gpx_sweep_update_dev(gpx_handle* h, int acq_id, const double* params, int nparams, int64_t k,
                                    double* top_val, int64_t* top_idx, double* d_acq_all, double* d_mu,
                                    double* d_s2) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        return sweep_update_core(h, acq_id, params, nparams, k, top_val, top_idx, d_acq_all, d_mu, d_s2);
    });
}

extern "C" int gpx_sweep_update(gpx_handle* h, int acq_id, const double* params, int nparams, int64_t k,
                                double* top_val, int64_t* top_idx, double* acq_all, double* mu, double* s2) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!h->cache_valid) return fail(h, GPX_ESTATE, "sweep_update: no valid sweep cache");
        HIPCHK(h, hipSetDevice(h->device));
        const int64_t M = h->cache_M;
        int rc;
        if ((rc = ensure(h, h->dXc, h->cap_xc, 3 * M))) return rc;       // staging [acq M][mu M][s2 M]
        double* dacq = h->dXc;
        double* dmu = dacq + M;
        double* ds2 = dmu + M;
        rc = sweep_update_core(h, acq_id, params, nparams, k, top_val, top_idx, dacq, mu ? dmu : nullptr,
                               s2 ? ds2 : nullptr);
        if (rc) return rc;
        {
            Span sp(h, T_COPY);
            if (acq_all) HIPCHK(h, hipMemcpyAsync(acq_all, dacq, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
            if (mu) HIPCHK(h, hipMemcpyAsync(mu, dmu, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
            if (s2) HIPCHK(h, hipMemcpyAsync(s2, ds2, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return GPX_OK;
    });
}

extern "C" int64_t gpx_sweep_cache_size(const gpx_handle* h) { return (h && h->cache_valid) ? h->cache_M : 0; }

extern "C" int gpx_predict(gpx_handle* h, const double* Xc, int64_t M, double* mu, double* s2, double* dmu,
                           double* ds2) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!mu || !s2) return fail(h, GPX_EARG, "predict: mu and s2 outputs are required");
        if (dmu || ds2) {
            if (!dmu || !ds2) return fail(h, GPX_EARG, "predict: pass both gradient outputs or neither");
            return gpx::predict_grad_host(h, Xc, M, mu, s2, dmu, ds2);
        }
        return gpx_sweep(h, GPX_ACQ_MEAN, nullptr, 0, Xc, M, 0, nullptr, nullptr, nullptr, mu, s2);
    });
}

extern "C" int gpx_predict_mean(gpx_handle* h, const double* Xc, int64_t M, double* mu, double* dmu) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        return gpx::predict_mean_host(h, Xc, M, mu, dmu);
    });
}

// ---- Thompson / RFF ---------------------------------------------------------------------------
static int rff_core(gpx_handle* h, const double* W, const double* b, const double* theta, int64_t S,
                    int64_t n, int64_t d, double bias, const double* dXc, int64_t M, int64_t k,
                    double* top_val, int64_t* top_idx, double* d_vals) {
    if (!W || !b || !theta || !dXc) return fail(h, GPX_EARG, "rff_sweep: NULL pointer");
    if (S < 1 || n < 1 || d < 1 || M < 1) return fail(h, GPX_EARG, "rff_sweep: bad sizes");
    if (d > DMAX_RFF) return fail(h, GPX_EARG, "rff_sweep: d must be <= 1024");
    if (k < 0 || k > TOPK_MAX) return fail(h, GPX_EARG, "rff_sweep: k must be in [0, 4096]");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int rc;
    // re-tile the draw parameters for the MFMA kernel: features padded to 128 per tile, k-major tiles
    const int64_t dp = (d + 3) / 4 * 4;
    const int64_t nfb = (n + TBH - 1) / TBH;
    const int64_t nW = S * nfb * dp * TBH, nV = S * nfb * TBH;
    std::vector<double> stage((size_t)(nW + 2 * nV), 0.0);
    {
        double* Wt = stage.data();
        double* bt = Wt + nW;
        double* tt = bt + nV;
        for (int64_t q = 0; q < S; ++q)
            for (int64_t j = 0; j < n; ++j) {
                const int64_t fb = j / TBH, c = j % TBH;
                for (int64_t kk = 0; kk < d; ++kk)
                    Wt[((q * nfb + fb) * dp + kk) * TBH + c] = W[(q * n + j) * d + kk];
                bt[(q * nfb + fb) * TBH + c] = b[q * n + j];
                tt[(q * nfb + fb) * TBH + c] = theta[q * n + j];
            }
    }
    if ((rc = ensure(h, h->drff, h->cap_rff, nW + 2 * nV))) return rc;
    double* dWt = h->drff;
    double* dbt = dWt + nW;
    double* dtt = dbt + nV;
    HIPCHK(h, hipMemcpyAsync(dWt, stage.data(), stage.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));   // `stage` is a local buffer
    if (!d_vals) {
        if ((rc = ensure(h, h->dout, h->cap_out, S * M))) return rc;
        d_vals = h->dout;
    }
    {
        Span sp(h, T_RFF);
        Span sp2(h, T_RFFSWEEP);          // the Thompson sweep kernel alone (bench.py: roofline_rff)
        launch_rff_mfma(s, dWt, dbt, dtt, (int)S, (int)nfb, (int)n, (int)d, (int)dp, bias, dXc, M, d_vals, h->dclk + 2);
    }
    // algorithmic double-precision lane operations of that launch: per (draw, feature, candidate) d multiply-adds of the
    // projection + 20 instructions of the cosine epilogue (kernels_rff.hip: cos_cw and the weighted sum)
    h->tacc[T_RFFOPS] += (double)S * (double)n * ((double)d + 20.0) * (double)M;
    if (k > 0) {
        const int64_t nblk = topk_blocks(M);
        // all draws ranked by one pair of launches while their block lists fit a modest buffer (k <= 64 per pass, S rows)
        const bool rows = k <= TOPK_PASS && S * nblk * k <= ((int64_t)1 << 24) && S <= 65535;
        const int64_t nlist = rows ? S * nblk * k : nblk * k;
        if ((rc = ensure(h, h->dblkv, h->cap_blk, nlist))) return rc;
        if (!h->dblki || h->cap_blki < nlist) {
            if (h->dblki) HIPCHK(h, hipFree(h->dblki));
            h->dblki = nullptr;
            HIPCHK(h, hipMalloc((void**)&h->dblki, (size_t)nlist * 8));
            h->cap_blki = nlist;
        }
        if ((rc = ensure(h, h->dtopv, h->cap_top, std::max<int64_t>((int64_t)TOPK_MAX * 2, S * k * 2)))) return rc;
        double* tv = h->dtopv;
        int64_t* ti = reinterpret_cast<int64_t*>(h->dtopv + S * k);
        {
            Span sp(h, T_RFF);
            if (rows)
                launch_topk_rows(s, d_vals, M, S, (int)k, h->dblkv, h->dblki, nblk, tv, ti);
            else
                for (int64_t q = 0; q < S; ++q)
                    launch_topk(s, d_vals + q * M, M, (int)k, h->dblkv, h->dblki, nblk, tv + q * k, ti + q * k);
        }
        h->last_topv = tv; h->last_topi = ti; h->last_topn = S * k;
        HIPCHK(h, hipMemcpyAsync(top_val, tv, (size_t)S * k * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(top_idx, ti, (size_t)S * k * 8, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    return GPX_OK;
}

extern "C" int gpx_rff_sweep_dev(gpx_handle* h, const double* W, const double* b, const double* theta,
                                 int64_t S, int64_t n, int64_t d, double bias, const double* dXc, int64_t M,
                                 int64_t k, double* top_val, int64_t* top_idx, double* d_vals_all) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        return rff_core(h, W, b, theta, S, n, d, bias, dXc, M, k, top_val, top_idx, d_vals_all);
    });
}

extern "C" int gpx_rff_sweep(gpx_handle* h, const double* W, const double* b, const double* theta, int64_t S,
                             int64_t n, int64_t d, double bias, const double* Xc, int64_t M, int64_t k,
                             double* top_val, int64_t* top_idx, double* vals_all) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (!Xc || M < 1 || d < 1) return fail(h, GPX_EARG, "rff_sweep: bad candidates");
        HIPCHK(h, hipSetDevice(h->device));
        int rc;
        if ((rc = ensure(h, h->dXc, h->cap_xc, M * d))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->dXc, Xc, (size_t)M * d * 8, hipMemcpyHostToDevice, h->stream));
        if ((rc = ensure(h, h->dout, h->cap_out, S * M))) return rc;
        rc = rff_core(h, W, b, theta, S, n, d, bias, h->dXc, M, k, top_val, top_idx, h->dout);
        if (rc) return rc;
        if (vals_all) {
            HIPCHK(h, hipMemcpyAsync(vals_all, h->dout, (size_t)S * M * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        return GPX_OK;
    });
}

extern "C" int gpx_rff_grad(gpx_handle* h, const double* W, const double* b, const double* theta, int64_t n,
                            int64_t d, double bias, const double* Xc, int64_t M, double* f, double* g) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        return gpx::rff_grad_host(h, W, b, theta, n, d, bias, Xc, M, f, g);
    });
}

// feature Grams of S draws (n < 128 features each) on the device: dA (S,n,n), dv (S,n) inside h->drff, followed by
// `extra` spare doubles (*dextra) for the caller
static int rff_gram_batch_dev(gpx_handle* h, const double* W, const double* b, int64_t S, int64_t n, int64_t extra,
                              double** dA_out, double** dv_out, double** dextra) {
    hipStream_t s = h->stream;
    const int64_t d = h->d, Np = h->Np, N = h->N;
    int rc;
    // batched MFMA path: feature tiles [S][dp][128] k-major, phases [S][128]
    const int64_t dp = (d + 3) / 4 * 4;
    const int64_t nW = S * dp * TBH, nV = S * TBH;
    std::vector<double> stage((size_t)(nW + nV), 0.0);
    for (int64_t q = 0; q < S; ++q)
        for (int64_t j = 0; j < n; ++j) {
            for (int64_t kk = 0; kk < d; ++kk) stage[(size_t)((q * dp + kk) * TBH + j)] = W[(q * n + j) * d + kk];
            stage[(size_t)(nW + q * TBH + j)] = b[q * n + j];
        }
    // device: [Wt nW][bt nV][A S*n*n][v S*n][extra]
    const int64_t need = nW + nV + S * n * n + S * n + extra;
    if ((rc = ensure(h, h->drff, h->cap_rff, need))) return rc;
    if ((rc = ensure(h, h->drffs, h->cap_rffs, rff_gram_batch_scratch(S, Np)))) return rc;
    double* dWt = h->drff;
    double* dbt = dWt + nW;
    double* dA = dbt + nV;
    double* dv = dA + S * n * n;
    HIPCHK(h, hipMemcpyAsync(dWt, stage.data(), stage.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));   // `stage` is a local buffer
    {
        Span sp(h, T_RFF);
        launch_rff_gram_batch(s, h->dXraw, N, Np, (int)d, (int)dp, dWt, dbt, (int)S, (int)n, h->dy, h->bias,
                              h->drffs, dA, dv);
    }
    HIPCHK(h, hipGetLastError());
    *dA_out = dA;
    *dv_out = dv;
    if (dextra) *dextra = dv + S * n;
    return GPX_OK;
}

extern "C" int gpx_rff_gram_batch(gpx_handle* h, const double* W, const double* b, int64_t S, int64_t n,
                                  double* A, double* v) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (h->stage < 1) return fail(h, GPX_ESTATE, "rff_gram: no data on the device (fit first)");
        if (!W || !b || !A || !v || n < 1 || S < 1) return fail(h, GPX_EARG, "rff_gram: bad arguments");
        HIPCHK(h, hipSetDevice(h->device));
        hipStream_t s = h->stream;
        const int64_t d = h->d, Np = h->Np, N = h->N;
        int rc;
        if (n >= TBH) {
            // wide feature maps: one draw at a time on the generic path
            // layout: [W n*d][b n][A n*n][v n][Ft n*Np]
            const int64_t need = n * d + n + n * n + n + n * Np;
            if ((rc = ensure(h, h->drff, h->cap_rff, need))) return rc;
            double* dW = h->drff;
            double* db = dW + n * d;
            double* dA = db + n;
            double* dv = dA + n * n;
            double* dFt = dv + n;
            for (int64_t q = 0; q < S; ++q) {
                HIPCHK(h, hipMemcpyAsync(dW, W + q * n * d, (size_t)n * d * 8, hipMemcpyHostToDevice, s));
                HIPCHK(h, hipMemcpyAsync(db, b + q * n, (size_t)n * 8, hipMemcpyHostToDevice, s));
                {
                    Span sp(h, T_RFF);
                    launch_rff_gram(s, h->dXraw, dFt, N, (int)d, dW, db, (int)n, h->dy, h->bias, dA, dv);
                }
                HIPCHK(h, hipMemcpyAsync(A + q * n * n, dA, (size_t)n * n * 8, hipMemcpyDeviceToHost, s));
                HIPCHK(h, hipMemcpyAsync(v + q * n, dv, (size_t)n * 8, hipMemcpyDeviceToHost, s));
                HIPCHK(h, hipStreamSynchronize(s));
            }
            HIPCHK(h, hipGetLastError());
            return GPX_OK;
        }
        double *dA, *dv;
        if ((rc = rff_gram_batch_dev(h, W, b, S, n, 0, &dA, &dv, nullptr))) return rc;
        HIPCHK(h, hipMemcpyAsync(A, dA, (size_t)S * n * n * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(v, dv, (size_t)S * n * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        HIPCHK(h, hipGetLastError());
        return GPX_OK;
    });
}

extern "C" int gpx_rff_posterior(gpx_handle* h, const double* W, const double* b, const double* z, int64_t S,
                                 int64_t n, double sc, double* theta) {
    return guarded(h, [&]() -> int {
        if (!h) return GPX_EARG;
        if (h->stage < 1) return fail(h, GPX_ESTATE, "rff_posterior: no data on the device (fit first)");
        if (!W || !b || !z || !theta || n < 1 || S < 1) return fail(h, GPX_EARG, "rff_posterior: bad arguments");
        if (n > 4096) return fail(h, GPX_EARG, "rff_posterior: n <= 4096 features");
        if (!(sc > 0.0) || !(h->sn2 > 0.0)) return fail(h, GPX_EARG, "rff_posterior: needs sc > 0 and a noise variance > 0");
        if (n >= TBH) {
            // wide feature maps (n >= 128: the weight posterior no longer fits the LDS-resident kernel): per draw the
            // generic feature Gram, then B = sc^2 A + sn2 I factorised by the blocked Cholesky kernels and two vector
            // substitutions, all on the device -- nothing but theta crosses PCIe
            HIPCHK(h, hipSetDevice(h->device));
            hipStream_t s = h->stream;
            const int64_t d = h->d, Np = h->Np, N = h->N, np = (n + NB - 1) / NB * NB;
            int rc;
            // layout: [W n*d][b n][A n*n][v n][z n][work n][theta n][B np*np][R np*np][Ft n*Np]
            const int64_t need = n * d + 6 * n + n * n + 2 * np * np + n * Np;
            if ((rc = ensure(h, h->drff, h->cap_rff, need))) return rc;
            double* dW = h->drff;
            double* db = dW + n * d;
            double* dA = db + n;
            double* dv = dA + n * n;
            double* dz = dv + n;
            double* dwork = dz + n;
            double* dth = dwork + n;
            double* dB = dth + n;
            double* dR = dB + np * np;
            double* dFt = dR + np * np;
            int* pflag = h->dflag + 8;
            for (int64_t q = 0; q < S; ++q) {
                HIPCHK(h, hipMemcpyAsync(dW, W + q * n * d, (size_t)n * d * 8, hipMemcpyHostToDevice, s));
                HIPCHK(h, hipMemcpyAsync(db, b + q * n, (size_t)n * 8, hipMemcpyHostToDevice, s));
                HIPCHK(h, hipMemcpyAsync(dz, z + q * n, (size_t)n * 8, hipMemcpyHostToDevice, s));
                int flag = 0;
                {
                    Span sp(h, T_RFF);
                    launch_rff_gram(s, h->dXraw, dFt, N, (int)d, dW, db, (int)n, h->dy, h->bias, dA, dv);
                    launch_posterior_wide(s, dA, dv, dz, (int)n, np, sc, h->sn2, dB, dR, dwork, pflag, dth);
                }
                HIPCHK(h, hipMemcpyAsync(theta + q * n, dth, (size_t)n * 8, hipMemcpyDeviceToHost, s));
                HIPCHK(h, hipMemcpyAsync(&flag, pflag, sizeof(int), hipMemcpyDeviceToHost, s));
                HIPCHK(h, hipStreamSynchronize(s));
                HIPCHK(h, hipGetLastError());
                if (flag != 0) return fail(h, GPX_ENOTPD, "rff_posterior: the feature Gram of a draw is not positive definite");
            }
            return GPX_OK;
        }
        HIPCHK(h, hipSetDevice(h->device));
        hipStream_t s = h->stream;
        int rc;
        double *dA, *dv, *dz;
        if ((rc = rff_gram_batch_dev(h, W, b, S, n, 2 * S * n, &dA, &dv, &dz))) return rc;
        double* dth = dz + S * n;
        int* pflag = h->dflag + 8;                    // its own word: the factorisation's flag stays untouched
        HIPCHK(h, hipMemsetAsync(pflag, 0, sizeof(int), s));
        HIPCHK(h, hipMemcpyAsync(dz, z, (size_t)S * n * 8, hipMemcpyHostToDevice, s));
        {
            Span sp(h, T_RFF);
            launch_rff_posterior(s, dA, dv, dz, (int)S, (int)n, sc, h->sn2, dth, pflag);
        }
        int flag = 0;
        HIPCHK(h, hipMemcpyAsync(theta, dth, (size_t)S * n * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(&flag, pflag, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        HIPCHK(h, hipGetLastError());
        if (flag != 0) return fail(h, GPX_ENOTPD, "rff_posterior: the feature Gram of a draw is not positive definite");
        return GPX_OK;
    });
}

extern "C" int gpx_rff_gram(gpx_handle* h, const double* W, const double* b, int64_t n, double* A, double* v) {
    return guarded(h, [&]() -> int {
        return gpx_rff_gram_batch(h, W, b, 1, n, A, v);
    });
}

// ---- ensemble sweep (hyper-parameter marginalisation; pybo/bayesopt.py:115) ------------------------------
static int ensemble_core(gpx_handle* const* mem, int n, int acq_id, const double* params, int nparams,
                         const double* dXc, int64_t M, int64_t k, double* top_val, int64_t* top_idx,
                         double* d_out, double* d_mu, double* d_s2) {
    gpx_handle* L = mem[0];
    if (acq_id < GPX_ACQ_EI || acq_id > GPX_ACQ_MEAN) return fail(L, GPX_EARG, "ensemble_sweep: unknown acquisition id");
    if (acq_id != GPX_ACQ_MEAN && (nparams < 1 || !params)) return fail(L, GPX_EARG, "ensemble_sweep: missing acquisition parameter");
    if (!dXc || M < 1) return fail(L, GPX_EARG, "ensemble_sweep: need M >= 1 candidates");
    if (k < 0 || k > TOPK_MAX) return fail(L, GPX_EARG, "ensemble_sweep: k must be in [0, 4096]");
    if (k > 0 && (!top_val || !top_idx)) return fail(L, GPX_EARG, "ensemble_sweep: NULL top-k output");
    for (int m = 0; m < n; ++m) {
        if (!mem[m]->fitted) return fail(L, GPX_ESTATE, "ensemble_sweep: a member model is not fitted");
        if (mem[m]->device != L->device || mem[m]->d != L->d)
            return fail(L, GPX_EARG, "ensemble_sweep: members must share the device and the input dimension");
    }
    HIPCHK(L, hipSetDevice(L->device));
    // mixture moments for UCB / mean (mu = mean mu_m, s2 = mean(s2_m + mu_m^2) - mu^2), plain mean for EI / PI
    const int mode = (acq_id == GPX_ACQ_UCB || acq_id == GPX_ACQ_MEAN) ? 1 : 0;
    int rc;
    if ((rc = ensure(L, L->dens, L->cap_ens, 5 * M))) return rc;
    double* acc0 = L->dens;
    double* acc1 = acc0 + M;
    double* t0 = acc1 + M;
    double* t1 = t0 + M;
    double* out = d_out ? d_out : t1 + M;
    for (int m = 0; m < n; ++m) {
        gpx_handle* h = mem[m];
        rc = (mode == 0) ? sweep_core(h, acq_id, params, nparams, dXc, M, 0, nullptr, nullptr, t0, nullptr, nullptr)
                         : sweep_core(h, GPX_ACQ_MEAN, nullptr, 0, dXc, M, 0, nullptr, nullptr, nullptr, t0, t1);
        if (rc) {
            if (h != L) L->err = "ensemble member " + std::to_string(m) + ": " + h->err;
            return rc;
        }
        HIPCHK(L, hipStreamSynchronize(h->stream));
        launch_ens_accum(L->stream, acc0, acc1, t0, t1, M, mode, m == 0);
        HIPCHK(L, hipStreamSynchronize(L->stream));    // t0/t1 are reused by the next member
    }
    const double beta = (acq_id == GPX_ACQ_UCB) ? params[0] : 0.0;
    if (acq_id == GPX_ACQ_MEAN)     // value = mixture mean
        launch_ens_finish(L->stream, acc0, acc1, M, 1, (double)n, 0.0, out, d_mu, d_s2);
    else
        launch_ens_finish(L->stream, acc0, acc1, M, mode, (double)n, beta, out, d_mu, d_s2);
    if (k > 0) {
        if ((rc = topk_core(L, out, M, k, top_val, top_idx))) return rc;
    } else {
        HIPCHK(L, hipStreamSynchronize(L->stream));
    }
    HIPCHK(L, hipGetLastError());
    return GPX_OK;
}

static int ensemble_check(gpx_handle* const* members, int n) {
    if (!members || n < 1 || !members[0]) return GPX_EARG;
    for (int m = 1; m < n; ++m)
        if (!members[m]) {
            members[0]->err = "ensemble_sweep: NULL member handle";
            return GPX_EARG;
        }
    return GPX_OK;
}

extern "C" int gpx_ensemble_predict(gpx_handle* const* members, int n_members, const double* Xc, int64_t M,
                                    double* mu, double* s2, double* dmu, double* ds2) {
    if (ensemble_check(members, n_members)) return GPX_EARG;
    gpx_handle* h = members[0];
    return guarded(h, [&]() -> int {
        if (!mu || !s2 || !dmu || !ds2) return fail(h, GPX_EARG, "ensemble_predict: all four outputs are required");
        return gpx::ensemble_predict_grad_host(members, n_members, Xc, M, mu, s2, dmu, ds2);
    });
}

extern "C" int gpx_ensemble_sweep_dev(gpx_handle* const* members, int n_members, int acq_id, const double* params,
                                      int nparams, const double* dXc, int64_t M, int64_t k, double* top_val,
                                      int64_t* top_idx, double* d_acq_all, double* d_mu, double* d_s2) {
    if (ensemble_check(members, n_members)) return GPX_EARG;
    return guarded(members[0], [&]() -> int {
        return ensemble_core(members, n_members, acq_id, params, nparams, dXc, M, k, top_val, top_idx, d_acq_all,
                             d_mu, d_s2);
    });
}

extern "C" int gpx_ensemble_sweep(gpx_handle* const* members, int n_members, int acq_id, const double* params,
                                  int nparams, const double* Xc, int64_t M, int64_t k, double* top_val,
                                  int64_t* top_idx, double* acq_all, double* mu, double* s2) {
    if (ensemble_check(members, n_members)) return GPX_EARG;
    gpx_handle* h = members[0];
    return guarded(h, [&]() -> int {
        if (!Xc || M < 1) return fail(h, GPX_EARG, "ensemble_sweep: need M >= 1 candidates");
        if ((mu || s2) && acq_id != GPX_ACQ_UCB && acq_id != GPX_ACQ_MEAN)
            return fail(h, GPX_EARG, "ensemble_sweep: mixture moments are only formed for UCB / mean");
        HIPCHK(h, hipSetDevice(h->device));
        int rc;
        const int64_t d = h->d;
        if ((rc = ensure(h, h->dXc, h->cap_xc, M * d + 3 * M))) return rc;
        double* dX = h->dXc;
        double* dacq = dX + M * d;
        double* dmu = dacq + M;
        double* ds2 = dmu + M;
        {
            Span sp(h, T_COPY);
            HIPCHK(h, hipMemcpyAsync(dX, Xc, (size_t)M * d * 8, hipMemcpyHostToDevice, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));     // the members read dX from their own streams
        rc = ensemble_core(members, n_members, acq_id, params, nparams, dX, M, k, top_val, top_idx, dacq,
                           mu ? dmu : nullptr, s2 ? ds2 : nullptr);
        if (rc) return rc;
        {
            Span sp(h, T_COPY);
            if (acq_all) HIPCHK(h, hipMemcpyAsync(acq_all, dacq, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
            if (mu) HIPCHK(h, hipMemcpyAsync(mu, dmu, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
            if (s2) HIPCHK(h, hipMemcpyAsync(s2, ds2, (size_t)M * 8, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return GPX_OK;
    });
}

// ---- candidate grids resident in HBM (pybo/solvers/lbfgs.py:45; pybo/inits/methods.py) -------------------
struct gpx_grid {
    int device = 0;
    int64_t M = 0, d = 0;
    double* dX = nullptr;
    hipStream_t s = nullptr;        // the grid's own (non-blocking) stream: nothing in this library touches the NULL stream -- its
                                    // first use in a process halves the speed of multi-stream schedules (see gpx_create)
};

#define GRIDCHK(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            g_create_err = std::string(#call) + " failed: " + hipGetErrorString(e_);               \
            return (e_ == hipErrorOutOfMemory) ? GPX_EOOM : GPX_EHIP;                              \
        }                                                                                          \
    } while (0)

static int grid_create_impl(int device, int kind, const double* bounds, int64_t M, int64_t d, uint64_t seed,
                            int64_t first, const uint32_t* sv, int bits, gpx_grid** out) {
    if (!out) return GPX_EARG;
    *out = nullptr;
    if (!bounds || M < 1 || d < 1 || d > DMAX) { g_create_err = "grid_create: bad sizes or NULL bounds"; return GPX_EARG; }
    if (kind != GPX_GRID_UNIFORM && kind != GPX_GRID_SOBOL) { g_create_err = "grid_create: unknown grid kind"; return GPX_EARG; }
    if (first < 0) { g_create_err = "grid_create: first must be >= 0"; return GPX_EARG; }
    if (kind == GPX_GRID_SOBOL && (!sv || bits < 1 || bits > 32 || first < 0 ||
                                   (bits < 63 && (uint64_t)(first + M) > (1ull << bits)))) {
        g_create_err = "grid_create: Sobol needs direction numbers (d, bits), 1 <= bits <= 32, first + M <= 2^bits";
        return GPX_EARG;
    }
    GRIDCHK(hipSetDevice(device));
    gpx_grid* g = new gpx_grid();
    g->device = device; g->M = M; g->d = d;
    double* dB = nullptr;
    uint32_t* dsv = nullptr;
    auto cleanup = [&]() { if (dB) hipFree(dB); if (dsv) hipFree(dsv); };
    auto bail = [&](int rc) { cleanup(); if (g->dX) hipFree(g->dX); if (g->s) hipStreamDestroy(g->s); delete g; return rc; };
#define GRIDTRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            g_create_err = std::string(#call) + " failed: " + hipGetErrorString(e_);               \
            return bail((e_ == hipErrorOutOfMemory) ? GPX_EOOM : GPX_EHIP);                        \
        }                                                                                          \
    } while (0)
    GRIDTRY(hipStreamCreateWithFlags(&g->s, hipStreamNonBlocking));
    GRIDTRY(hipMalloc((void**)&g->dX, (size_t)M * d * 8));
    GRIDTRY(hipMalloc((void**)&dB, (size_t)d * 2 * 8));
    GRIDTRY(hipMemcpyAsync(dB, bounds, (size_t)d * 2 * 8, hipMemcpyHostToDevice, g->s));
    if (kind == GPX_GRID_SOBOL) {
        GRIDTRY(hipMalloc((void**)&dsv, (size_t)d * bits * 4));
        GRIDTRY(hipMemcpyAsync(dsv, sv, (size_t)d * bits * 4, hipMemcpyHostToDevice, g->s));
        launch_grid_sobol(g->s, dsv, bits, first, M, (int)d, dB, g->dX);
    } else {
        launch_grid_uniform(g->s, seed, first, M, (int)d, dB, g->dX);
    }
    GRIDTRY(hipGetLastError());
    GRIDTRY(hipStreamSynchronize(g->s));             // the grid is complete when this call returns: any stream may read it
    cleanup();
    *out = g;
    return GPX_OK;
}

extern "C" int gpx_grid_create(int device, int kind, const double* bounds, int64_t M, int64_t d, uint64_t seed,
                               int64_t first, const uint32_t* sv, int bits, gpx_grid** out) {
    try {
        return grid_create_impl(device, kind, bounds, M, d, seed, first, sv, bits, out);
    } catch (...) {
        if (out) *out = nullptr;
        return GPX_EOOM;
    }
}

extern "C" const double* gpx_grid_data(const gpx_grid* g) { return g ? g->dX : nullptr; }

extern "C" int gpx_grid_rows(gpx_grid* g, const int64_t* idx, int64_t k, double* out) {
    try {
        if (!g || !out) { g_create_err = "grid_rows: NULL pointer"; return GPX_EARG; }
        GRIDCHK(hipSetDevice(g->device));
        if (!idx) {                                   // the whole grid
            GRIDCHK(hipMemcpyAsync(out, g->dX, (size_t)g->M * g->d * 8, hipMemcpyDeviceToHost, g->s));
            GRIDCHK(hipStreamSynchronize(g->s));
            return GPX_OK;
        }
        if (k < 1) return GPX_OK;
        for (int64_t i = 0; i < k; ++i)
            if (idx[i] < 0 || idx[i] >= g->M) { g_create_err = "grid_rows: index out of range"; return GPX_EARG; }
        for (int64_t i = 0; i < k;) {                 // one copy per run of consecutive rows
            int64_t j = i + 1;
            while (j < k && idx[j] == idx[j - 1] + 1) ++j;
            GRIDCHK(hipMemcpyAsync(out + i * g->d, g->dX + idx[i] * g->d, (size_t)(j - i) * g->d * 8, hipMemcpyDeviceToHost, g->s));
            i = j;
        }
        GRIDCHK(hipStreamSynchronize(g->s));
        return GPX_OK;
    } catch (...) {
        return GPX_EOOM;
    }
}

extern "C" int gpx_grid_destroy(gpx_grid* g) {
    if (!g) return GPX_OK;
    hipSetDevice(g->device);
    if (g->dX) hipFree(g->dX);
    if (g->s) hipStreamDestroy(g->s);
    delete g;
    return GPX_OK;
}
