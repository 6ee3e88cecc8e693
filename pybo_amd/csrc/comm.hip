// comm.hip -- the ONE exchange step of the sharded sweep, behind the C-ABI (gpx_comm_init,
// gpx_topk_allgather).  The reference has no distributed code (its only hint is the comment at
// pybo/solvers/lbfgs.py:60); the layout is SURVEY.md 8(e): candidates sharded contiguously over ranks, fit
// replicated, and each rank's k best (value, GLOBAL index) pairs all-gathered and merged with the deterministic
// rule (value descending, then index ascending) -- RCCL has no MAXLOC.
//
// The pairs never visit the host on their way out: the last sweep left them in HBM (h->last_topv / last_topi),
// a pack kernel adds the shard offset, ncclAllGather runs on the handle's stream over xGMI, and the merge is
// the same single-workgroup kernel that merges the per-block top-k lists of a sweep.  Only the k winners
// (16 k bytes) are copied back.
//
// RCCL is bound at run time (dlopen), so libgpx.so itself has no link-time dependency on it: a single-GPU
// consumer of include/gpx.h never loads the collective library.  The header <rccl/rccl.h> is used for its
// types and enumerators only.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "gpx_internal.h"

using namespace gpx;

static thread_local std::string g_comm_err;

extern "C" const char* gpx_comm_last_error(void) { return g_comm_err.c_str(); }

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// One process-wide binding.  Search order: $GPX_RCCL_LIB, then the soname (which resolves to a copy already
// mapped into the process -- e.g. the one torch.distributed brought -- before touching the library path).
static std::string g_load_err;     // why the binding failed (written once, under the once-flag; read-only afterwards)

static void rccl_bind(Rccl& r);

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_bind(r); });          // thread-safe: handles are driven from several host threads
    if (!r.lib) {
        g_comm_err = g_load_err;                         // EVERY failing call reports why, not only the first
        return nullptr;
    }
    return &r;
}

static void rccl_bind(Rccl& r) {
    const char* names[] = {getenv("GPX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        const char* e = dlerror();
        g_load_err = std::string("cannot load librccl (set GPX_RCCL_LIB): ") + (e ? e : "unknown error");
        return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        g_load_err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
        dlclose(r.lib);
        r.lib = nullptr;
    }
}

int nccl_fail(Rccl* r, const char* what, ncclResult_t e) {
    g_comm_err = std::string(what) + " failed: " + r->GetErrorString(e);
    return GPX_ERCCL;
}

// A rank's message: n (value, index + offset) pairs, then ONE status pair (status, rank).  status != 0: this rank
// could not contribute (its last sweep left no n pairs on the device) -- it still takes part in the collective, with
// padding entries, so that no rank is left blocked in the all-gather and EVERY rank returns the same error.
// send[2i], send[2i+1] = pair i (padding: index -1); v == NULL: padding only
__global__ void k_pack_pairs(const double* __restrict__ v, const int64_t* __restrict__ idx, int64_t n, int64_t off,
                             int64_t status, int64_t rank, double* __restrict__ send) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        send[2 * n] = __longlong_as_double(status);
        send[2 * n + 1] = __longlong_as_double(rank);
        return;
    }
    if (!v) {
        send[2 * i] = -HUGE_VAL;
        send[2 * i + 1] = __longlong_as_double((int64_t)-1);
        return;
    }
    const int64_t g = idx[i] < 0 ? (int64_t)-1 : idx[i] + off;
    send[2 * i] = v[i];
    send[2 * i + 1] = __longlong_as_double(g);
}

// pair j of rank r (message stride n + 1 pairs) -> vals[r n + j], idx[r n + j]; index < 0 becomes the "no entry" marker
// of the merge kernel; stat[r] = rank r's status word
__global__ void k_unpack_pairs(const double* __restrict__ all, int64_t n, int64_t nranks, double* __restrict__ vals,
                               int64_t* __restrict__ idx, int for_merge, int64_t* __restrict__ stat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nranks) stat[i] = __double_as_longlong(all[2 * (i * (n + 1) + n)]);
    if (i >= n * nranks) return;
    const int64_t r = i / n, j = i - r * n;
    const double* src = all + 2 * (r * (n + 1) + j);
    const int64_t g = __double_as_longlong(src[1]);
    vals[i] = src[0];
    idx[i] = (g < 0 && for_merge) ? (int64_t)0x7fffffffffffffffLL : g;
}

}  // namespace

struct gpx_comm {
    gpx_handle* h = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    double* dbuf = nullptr;     // [send 2n][all 2 n nranks][vals n nranks][idx n nranks][topv k][topi k]
    int64_t cap = 0;
};

extern "C" int gpx_comm_unique_id(unsigned char* id) {
    try {
        if (!id) { g_comm_err = "comm_unique_id: NULL buffer"; return GPX_EARG; }
        Rccl* r = rccl();
        if (!r) return GPX_ERCCL;
        ncclUniqueId u;
        ncclResult_t e = r->GetUniqueId(&u);
        if (e != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", e);
        static_assert(NCCL_UNIQUE_ID_BYTES == GPX_COMM_ID_BYTES, "id size");
        memcpy(id, u.internal, GPX_COMM_ID_BYTES);
        return GPX_OK;
    } catch (...) {
        return GPX_EOOM;
    }
}

extern "C" int gpx_comm_init(gpx_handle* h, int rank, int nranks, const unsigned char* id, gpx_comm** out) {
    try {
        if (out) *out = nullptr;
        if (!h || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) {
            g_comm_err = "comm_init: bad arguments";
            return GPX_EARG;
        }
        Rccl* r = rccl();
        if (!r) return GPX_ERCCL;
        if (hipSetDevice(h->device) != hipSuccess) { g_comm_err = "comm_init: hipSetDevice failed"; return GPX_EHIP; }
        gpx_comm* c = new (std::nothrow) gpx_comm();
        if (!c) { g_comm_err = "comm_init: out of host memory"; return GPX_EOOM; }
        c->h = h; c->rank = rank; c->nranks = nranks;
        ncclUniqueId u;
        memcpy(u.internal, id, GPX_COMM_ID_BYTES);
        ncclResult_t e = r->CommInitRank(&c->comm, nranks, u, rank);      // collective: every rank calls it
        if (e != ncclSuccess) { delete c; return nccl_fail(r, "ncclCommInitRank", e); }
        *out = c;
        return GPX_OK;
    } catch (...) {
        return GPX_EOOM;
    }
}

extern "C" int gpx_comm_destroy(gpx_comm* c) {
    if (!c) return GPX_OK;
    Rccl* r = rccl();
    hipSetDevice(c->h->device);
    hipStreamSynchronize(c->h->stream);
    if (r && c->comm) r->CommDestroy(c->comm);
    if (c->dbuf) hipFree(c->dbuf);
    delete c;
    return GPX_OK;
}

extern "C" int gpx_comm_size(const gpx_comm* c, int* rank, int* nranks) {
    if (!c) return GPX_EARG;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return GPX_OK;
}

extern "C" int gpx_topk_allgather(gpx_comm* c, int64_t n, int64_t index_offset, int64_t k, double* out_val,
                                  int64_t* out_idx) {
    try {
        if (!c || !out_val || !out_idx || n < 1 || k < 0 || k > TOPK_MAX) {
            g_comm_err = "topk_allgather: bad arguments (need n >= 1, 0 <= k <= 4096, output buffers)";
            return GPX_EARG;
        }
        gpx_handle* h = c->h;
        // a rank that cannot contribute still joins the collective (see k_pack_pairs): the verdict is collective too
        const bool mine_ok = h->last_topv && h->last_topi && h->last_topn == n;
        Rccl* r = rccl();
        if (!r) return GPX_ERCCL;
        if (hipSetDevice(h->device) != hipSuccess) { g_comm_err = "topk_allgather: hipSetDevice failed"; return GPX_EHIP; }
        hipStream_t s = h->stream;
        const int64_t W = c->nranks, tot = n * W;
        // [send 2(n+1)][all 2(n+1)W][vals tot][idx tot][stat W][topv k][topi k]
        const int64_t need = 2 * (n + 1) + 2 * (n + 1) * W + 2 * tot + W + 2 * (k > 0 ? k : 1);
        if (need > c->cap) {
            if (c->dbuf) hipFree(c->dbuf);
            c->dbuf = nullptr;
            c->cap = 0;
            if (hipMalloc((void**)&c->dbuf, (size_t)need * 8) != hipSuccess) {
                g_comm_err = "topk_allgather: device allocation failed";
                return GPX_EOOM;
            }
            c->cap = need;
        }
        double* send = c->dbuf;
        double* all = send + 2 * (n + 1);
        double* vals = all + 2 * (n + 1) * W;
        int64_t* idx = reinterpret_cast<int64_t*>(vals + tot);
        int64_t* stat = idx + tot;
        double* topv = reinterpret_cast<double*>(stat + W);
        int64_t* topi = reinterpret_cast<int64_t*>(topv + (k > 0 ? k : 1));
        hipLaunchKernelGGL(k_pack_pairs, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s,
                           mine_ok ? h->last_topv : (const double*)nullptr, h->last_topi, n, index_offset,
                           (int64_t)(mine_ok ? 0 : 1), (int64_t)c->rank, send);
        // (value, index) pairs travel as raw 64-bit words
        ncclResult_t e = r->AllGather(send, all, (size_t)(2 * (n + 1)), ncclUint64, c->comm, s);
        if (e != ncclSuccess) return nccl_fail(r, "ncclAllGather", e);
        const int64_t nthreads = tot > W ? tot : W;
        hipLaunchKernelGGL(k_unpack_pairs, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, all, n, W, vals, idx,
                           k > 0 ? 1 : 0, stat);
        std::vector<int64_t> hstat((size_t)W, 0);
        bool ok = hipMemcpyAsync(hstat.data(), stat, (size_t)W * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
        if (k > 0) {
            launch_topk_merge(s, vals, idx, tot, (int)k, topv, topi);
            ok = ok && hipMemcpyAsync(out_val, topv, (size_t)k * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipMemcpyAsync(out_idx, topi, (size_t)k * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
        } else {
            ok = ok && hipMemcpyAsync(out_val, vals, (size_t)tot * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipMemcpyAsync(out_idx, idx, (size_t)tot * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
        }
        if (!ok || hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
            g_comm_err = "topk_allgather: kernel, collective or D2H copy failed";
            return GPX_EHIP;
        }
        for (int64_t q = 0; q < W; ++q)
            if (hstat[(size_t)q] != 0) {           // the same verdict on every rank
                g_comm_err = "topk_allgather: rank " + std::to_string(q) +
                             "'s last sweep did not leave n (value, index) pairs on the device";
                return GPX_ESTATE;
            }
        return GPX_OK;
    } catch (...) {
        return GPX_EOOM;
    }
}
