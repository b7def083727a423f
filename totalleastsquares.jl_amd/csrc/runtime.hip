// Runtime of libtlsqhip.so: error reporting, the grow-only device workspace, the pinned upload ring, the RCCL
// row-shard exchange (librccl is loaded with dlopen), and the handle / communicator entry points of include/tlsq.h.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <numeric>
#include <thread>
#include <condition_variable>
#include <memory>
#include <mutex>

#include "internal.hpp"

namespace tlsq {


// ------------------------------------------------------------------------------------------------
// development switches (declared in common.hpp)
// ------------------------------------------------------------------------------------------------
static const char* const kDevNames[DEV_COUNT] = {
#define TLSQ_DEV_NAME(n) #n,
    TLSQ_DEV_LIST(TLSQ_DEV_NAME)
#undef TLSQ_DEV_NAME
};
static std::string g_dev_val[DEV_COUNT];
static bool g_dev_set[DEV_COUNT] = {};

const char* dev_get(DevKey k) { return g_dev_set[k] ? g_dev_val[k].c_str() : nullptr; }

int dev_set(const char* name, const char* value) {
    if (!name) return TLSQ_ERR_ARG;
    if (strncmp(name, "TLSQ_", 5) == 0) name += 5;
#ifdef TLSQ_DEV_SWITCHES
    const int nkeys = DEV_COUNT;
#else
    const int nkeys = DEV_LIVE_COUNT;   // (the ablation switches exist in development builds only: common.hpp)
#endif
    for (int k = 0; k < nkeys; ++k)
        if (strcmp(name, kDevNames[k]) == 0) {
            g_dev_set[k] = value != nullptr;
            g_dev_val[k] = value ? value : "";
            return TLSQ_OK;
        }
    return TLSQ_ERR_ARG;
}

double now_ms();
// HOST_TRACE=1: wall-clock marks of the host side of a call (where the time between two kernels of the trace goes when the GPU
// is waiting for the host), printed to stderr when the call returns.  One table for the process: a tool for single-handle runs
// (the worker threads of a multi-GPU group would interleave their marks)
static struct { const char* tag; double t; } g_host_marks[64];
static int g_host_nmarks = 0;
void host_mark(const char* tag) {
    if (!g_dev_set[DEV_HOST_TRACE]) return;
    if (g_host_nmarks < 64) g_host_marks[g_host_nmarks++] = {tag, now_ms()};
}
void host_trace_dump() {
    if (!g_dev_set[DEV_HOST_TRACE] || g_host_nmarks == 0) return;
    fprintf(stderr, "[host]");
    for (int i = 0; i < g_host_nmarks; ++i)
        fprintf(stderr, " %s %+.1f us |", g_host_marks[i].tag, i ? (g_host_marks[i].t - g_host_marks[i - 1].t) * 1e3 : 0.0);
    fprintf(stderr, " total %.1f us\n", (g_host_marks[g_host_nmarks - 1].t - g_host_marks[0].t) * 1e3);
    g_host_nmarks = 0;
}

void dev_load_env() {
#ifdef TLSQ_DEV_SWITCHES
    static bool done = false;
    if (done) return;
    done = true;
    for (int k = 0; k < DEV_COUNT; ++k) {
        const std::string var = std::string("TLSQ_") + kDevNames[k];
        if (const char* e = getenv(var.c_str())) (void)dev_set(kDevNames[k], e);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// errors / workspace
// ------------------------------------------------------------------------------------------------
int set_err(Handle* h, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) {   // (a host-pointer call's download thread and the solver thread may both fail: one string, one writer at a time)
        static std::mutex err_mu;
        std::lock_guard<std::mutex> lk(err_mu);
        h->err = buf;
    }
    return code;
}

int ws_get(Handle* h, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    DevBuf& b = h->ws[slot];
    if (b.bytes < bytes) {
        if (b.p) {
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            TLSQ_HIP(h, hipFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
        }
        size_t want = (bytes + 255) & ~size_t(255);
        TLSQ_HIP(h, hipMalloc(&b.p, want));
        b.bytes = want;
        // WS_POISON: fresh workspace memory reads as NaN (0xFF bytes) in fp64 and fp32 and as -1 in integers, so that a
        // kernel that reads what this call has not written shows up in the results instead of depending on whatever
        // the allocator handed back
        if (dev_is(DEV_WS_POISON, '1')) TLSQ_HIP(h, hipMemsetAsync(b.p, 0xFF, want, h->stream));
    }
    *out = b.p;
    return TLSQ_OK;
}

void dbg_hash(Handle* h, const char* tag, const void* dev_ptr, size_t bytes, long long k) {
    if (!dev_is(DEV_DEBUG_HASH, '1') || !dev_ptr || bytes == 0) return;
    std::vector<unsigned char> buf(bytes);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return;
    if (h->stream_b) (void)hipStreamSynchronize(h->stream_b);
    if (hipMemcpy(buf.data(), dev_ptr, bytes, hipMemcpyDeviceToHost) != hipSuccess) return;
    uint64_t x = 1469598103934665603ull;
    for (size_t i = 0; i < bytes; ++i) {
        x ^= buf[i];
        x *= 1099511628211ull;
    }
    fprintf(stderr, "[hash] r%d k=%lld %s %016llx\n", h->rank, k, tag, (unsigned long long)x);
}

// WS_POISON=1: every slot the handle holds is refilled with 0xFF bytes at the start of a solve, and the flags that say
// "this piece of device state has been initialised" are dropped with it - a solve then cannot see anything an earlier
// call on the same handle left behind (tests/test_gpu_determinism.py).
int ws_poison_all(Handle* h) {
    if (!h || !dev_is(DEV_WS_POISON, '1') || h->in_multi) return TLSQ_OK;
    for (Handle* sub : h->subs) TLSQ_TRY(ws_poison_all(sub));
    TLSQ_HIP(h, hipSetDevice(h->device));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    for (auto& b : h->ws)
        if (b.p) TLSQ_HIP(h, hipMemsetAsync(b.p, 0xFF, b.bytes, h->stream));
    h->mail_counter_ready = false;
    h->cert_ticket_ready = false;
    h->gram_tab_nti = 0;
    h->gram_tab2_nti = 0;
    h->absmax_panel = nullptr;
    h->gram_tab3_ready = false;
    h->warm_n = 0;
    if (h->mailbox)
        for (size_t i = 1; i < h->mailbox_bytes / 8; ++i) h->mailbox[i] = std::numeric_limits<double>::quiet_NaN();
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    return TLSQ_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL (loaded lazily; single-GPU use never touches it)
// ------------------------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static bool rccl_load() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return false;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(lib, "ncclCommInitAll");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(lib, "ncclCommAbort");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(lib, "ncclCommCount");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        dlclose(lib);
        return false;
    }
    g_rccl.lib = lib;
    return true;
}

// Loop-back group: the "ranks" of a tlsq_create_multi handle that names the same device more than once (tests: the
// rank > 1 control flow of the sharded solvers on a one-GPU box; RCCL refuses duplicate devices).  Collectives go
// through host memory: every rank stages its buffer, a barrier, every rank reduces all slots in rank order (identical
// bits everywhere, like RCCL's all-reduce), a second barrier before the slots are reused.  Slow, and only meant to be.
struct LocalGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<std::vector<double>> slot;
    bool failed = false;   // a rank has left the group call with an error: nobody waits for it any more
    // false: a rank has failed (multi_run says so at once), or the others did not arrive within a minute
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (failed) return false;
        const uint64_t g = gen;
        if (++arrived == n) {
            arrived = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        // The product waits on the steady clock (wait_for): a step of the wall clock (NTP, a resumed VM) must neither fail the
        // group nor stretch the timeout.  Only the ThreadSanitizer build waits on the system clock (wait_until =
        // pthread_cond_timedwait): wait_for goes through pthread_cond_clockwait, which the TSan runtime of this toolchain does
        // not intercept - it then believes the mutex is held across the wait.
#if defined(__SANITIZE_THREAD__) || defined(TLSQ_TSAN_BUILD)
        const bool woke = cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::seconds(60), [&] { return gen != g || failed; });
#else
        const bool woke = cv.wait_for(lk, std::chrono::seconds(60), [&] { return gen != g || failed; });
#endif
        if (woke && gen != g) return true;
        --arrived;
        return false;
    }
    void fail() {
#ifdef TLSQ_TSAN_PLANT_RACE   // (tests/test_tsan_cpu.py: the harness must report this)
        failed = true;
        cv.notify_all();
        return;
#endif
        std::lock_guard<std::mutex> lk(m);
        failed = true;
        cv.notify_all();
    }
    void reset() {
        std::lock_guard<std::mutex> lk(m);
        failed = false;
        arrived = 0;
    }
};

struct Comm {
    // `comm` is used by its rank's thread (collectives) and may be aborted from ANOTHER rank's thread (multi_run's
    // fail_group, when that rank leaves the group call with an error): every use of the communicator - the enqueue of a
    // collective, the abort - happens under `m`, and an aborted communicator is never handed to RCCL again (ncclCommAbort
    // frees it).  ADVICE r3: the abort used to race with a collective that had just read the pointer.
    std::mutex m;
    ncclComm_t comm = nullptr;
    std::shared_ptr<LocalGroup> local;   // set instead of `comm` for a loop-back group
    int local_rank = 0;
    std::atomic<bool> aborted{false};    // ncclCommAbort has been called on `comm` (a rank of the group failed)
    std::shared_ptr<std::atomic<bool>> group_failed;   // shared by the ranks of a tlsq_create_multi group
};

#define TLSQ_NCCL(h, expr)                                                                     \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess)                                                                 \
            return set_err((h), TLSQ_ERR_COMM, "%s failed: %s", #expr,                         \
                           g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error"); \
    } while (0)

// in-place sum of an N x N Gram over the row shards (the one real exchange of the path)
static int group_alive(Handle* h) {
    if (h->comm->group_failed && h->comm->group_failed->load())
        return set_err(h, TLSQ_ERR_COMM, "multi-GPU group: another rank left the call with an error");
    return TLSQ_OK;
}

int comm_allreduce(Handle* h, double* dev, size_t count, ncclRedOp_t op) {
    if (!h->comm) return TLSQ_OK;
    TLSQ_TRY(group_alive(h));
    if (h->comm->local) {
        LocalGroup& g = *h->comm->local;
        std::vector<double>& mine = g.slot[(size_t)h->comm->local_rank];
        mine.resize(count);
        TLSQ_HIP(h, hipMemcpyAsync(mine.data(), dev, count * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        if (!g.barrier()) return set_err(h, TLSQ_ERR_COMM, "loop-back all-reduce: a rank is missing (diverged control flow?)");
        std::vector<double> acc(g.slot[0]);
        bool ok = acc.size() == count;
        for (int r = 1; r < g.n && ok; ++r) {
            const std::vector<double>& o = g.slot[(size_t)r];
            if (o.size() != count) {
                ok = false;
                break;
            }
            for (size_t i = 0; i < count; ++i) {
                if (op == ncclSum) acc[i] += o[i];
                else if (op == ncclMax) acc[i] = o[i] > acc[i] ? o[i] : acc[i];
                else if (op == ncclMin) acc[i] = o[i] < acc[i] ? o[i] : acc[i];
                else acc[i] *= o[i];
            }
        }
        if (!g.barrier()) return set_err(h, TLSQ_ERR_COMM, "loop-back all-reduce: a rank is missing (diverged control flow?)");
        if (!ok) return set_err(h, TLSQ_ERR_COMM, "loop-back all-reduce: the ranks disagree on the element count");
        TLSQ_HIP(h, hipMemcpyAsync(dev, acc.data(), count * 8, hipMemcpyHostToDevice, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        return TLSQ_OK;
    }
    {
        std::lock_guard<std::mutex> lk(h->comm->m);
        if (h->comm->aborted.load() || !h->comm->comm)
            return set_err(h, TLSQ_ERR_COMM, "multi-GPU group: the communicator was aborted (another rank left the call with an error)");
        TLSQ_NCCL(h, g_rccl.AllReduce(dev, dev, count, ncclDouble, op, h->comm->comm, h->stream));
    }
    return TLSQ_OK;
}

// the TSQR exchange: every rank's N x N triangular factor to every rank
int comm_allgather(Handle* h, const double* send, double* recv, size_t count) {
    if (!h->comm) {
        TLSQ_HIP(h, hipMemcpyAsync(recv, send, count * 8, hipMemcpyDeviceToDevice, h->stream));
        return TLSQ_OK;
    }
    TLSQ_TRY(group_alive(h));
    if (h->comm->local) {
        LocalGroup& g = *h->comm->local;
        std::vector<double>& mine = g.slot[(size_t)h->comm->local_rank];
        mine.resize(count);
        TLSQ_HIP(h, hipMemcpyAsync(mine.data(), send, count * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        if (!g.barrier()) return set_err(h, TLSQ_ERR_COMM, "loop-back all-gather: a rank is missing (diverged control flow?)");
        std::vector<double> all((size_t)g.n * count);
        bool ok = true;
        for (int r = 0; r < g.n; ++r) {
            if (g.slot[(size_t)r].size() != count) {
                ok = false;
                break;
            }
            memcpy(all.data() + (size_t)r * count, g.slot[(size_t)r].data(), count * 8);
        }
        if (!g.barrier()) return set_err(h, TLSQ_ERR_COMM, "loop-back all-gather: a rank is missing (diverged control flow?)");
        if (!ok) return set_err(h, TLSQ_ERR_COMM, "loop-back all-gather: the ranks disagree on the element count");
        TLSQ_HIP(h, hipMemcpyAsync(recv, all.data(), all.size() * 8, hipMemcpyHostToDevice, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        return TLSQ_OK;
    }
    {
        std::lock_guard<std::mutex> lk(h->comm->m);
        if (h->comm->aborted.load() || !h->comm->comm)
            return set_err(h, TLSQ_ERR_COMM, "multi-GPU group: the communicator was aborted (another rank left the call with an error)");
        TLSQ_NCCL(h, g_rccl.AllGather(send, recv, count, ncclDouble, h->comm->comm, h->stream));
    }
    return TLSQ_OK;
}

int comm_allreduce_host_vec(Handle* h, double* v, int count, ncclRedOp_t op) {
    if (!h->comm) return TLSQ_OK;
    if (count < 1 || count > 8) return set_err(h, TLSQ_ERR_ARG, "comm_allreduce_host_vec: count");
    void* slot;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &slot));
    double* d = reinterpret_cast<double*>(reinterpret_cast<char*>(slot) + 256);   // [256, 320): 8 doubles (320.. belongs to others)
    TLSQ_HIP(h, hipMemcpyAsync(d, v, (size_t)count * 8, hipMemcpyHostToDevice, h->stream));
    TLSQ_TRY(comm_allreduce(h, d, count, op));
    TLSQ_HIP(h, hipMemcpyAsync(v, d, (size_t)count * 8, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    return TLSQ_OK;
}

int comm_allreduce_host_scalar(Handle* h, double* v, ncclRedOp_t op) { return comm_allreduce_host_vec(h, v, 1, op); }

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
int copy2d(Handle* h, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t rows,
                  int64_t cols, size_t esz, hipMemcpyKind kind) {
    if (rows <= 0 || cols <= 0) return TLSQ_OK;
    if (ldd == rows && lds == rows) {
        TLSQ_HIP(h, hipMemcpyAsync(dst, src, (size_t)rows * cols * esz, kind, h->stream));
    } else {
        TLSQ_HIP(h, hipMemcpy2DAsync(dst, (size_t)ldd * esz, src, (size_t)lds * esz, (size_t)rows * esz,
                                     (size_t)cols, kind, h->stream));
    }
    return TLSQ_OK;
}

int multi_run(Handle* h, const std::function<int(Handle*, int, int)>& fn) {
    const int n = h->multi_n;
    std::vector<Handle*> hs((size_t)n);
    hs[0] = h;
    for (int r = 1; r < n; ++r) hs[(size_t)r] = h->subs[(size_t)r - 1];
    std::vector<int> st((size_t)n, TLSQ_OK);
    if (h->multi_comm && h->multi_comm->local) {
        h->multi_comm->local->reset();
        if (h->multi_comm->group_failed) h->multi_comm->group_failed->store(false);   // (a loop-back group survives a failed call)
    }
    for (int r = 0; r < n; ++r)
        if (hs[(size_t)r]->multi_comm && hs[(size_t)r]->multi_comm->aborted.load())
            return set_err(h, TLSQ_ERR_COMM, "multi-GPU group: the communicators were aborted by an earlier failed call; "
                                             "destroy the handle and create a new one");
    // A rank that leaves the call with an error before a collective the others have entered would leave them blocked for
    // ever (ncclAllReduce / the stream synchronisation behind it) and join() below would never return: the failing rank
    // therefore takes the whole group down - ncclCommAbort on every communicator of the group (the collectives' kernels
    // poll the abort flag and exit), the fail flag of a loop-back group - and every other rank comes back with an error.
    std::mutex abort_m;
    bool group_failed = false;
    auto fail_group = [&]() {
        std::lock_guard<std::mutex> lk(abort_m);
        if (group_failed) return;
        group_failed = true;
        for (int q = 0; q < n; ++q)
            if (hs[(size_t)q]->multi_comm && hs[(size_t)q]->multi_comm->group_failed) hs[(size_t)q]->multi_comm->group_failed->store(true);
        for (int q = 0; q < n; ++q) {
            Comm* c = hs[(size_t)q]->multi_comm;
            if (!c) continue;
            if (c->local) {
                c->local->fail();
            } else {
                // (under the communicator's own lock: its rank is either before the enqueue of a collective - it will find the
                //  flag - or past it, in which case the abort makes the collective's kernel give up)
                std::lock_guard<std::mutex> ck(c->m);
                if (c->comm && g_rccl.CommAbort && !c->aborted.load()) {
                    c->aborted.store(true);
                    (void)g_rccl.CommAbort(c->comm);
                    c->comm = nullptr;
                }
            }
        }
    };
    auto body = [&](int r) {
        Handle* hr = hs[(size_t)r];
        if (hipSetDevice(hr->device) != hipSuccess) {
            st[(size_t)r] = set_err(hr, TLSQ_ERR_HIP, "hipSetDevice(%d) failed", hr->device);
            fail_group();
            return;
        }
        Comm* prev_comm = hr->comm;     // (a communicator of tlsq_comm_init on this handle is put back afterwards)
        const int prev_n = hr->nranks, prev_r = hr->rank;
        hr->comm = hr->multi_comm;
        hr->nranks = n;
        hr->rank = r;
        hr->in_multi = true;
        try {
            st[(size_t)r] = fn(hr, r, n);
        } catch (const std::bad_alloc&) {
            st[(size_t)r] = set_err(hr, TLSQ_ERR_OOM, "out of host memory on rank %d", r);
        } catch (const std::exception& e) {
            st[(size_t)r] = set_err(hr, TLSQ_ERR_HIP, "exception on rank %d: %s", r, e.what());
        }
        if (st[(size_t)r] < 0 && n > 1) fail_group();
        (void)hipStreamSynchronize(hr->stream);
        hr->in_multi = false;
        hr->comm = prev_comm;
        hr->nranks = prev_n;
        hr->rank = prev_r;
    };
    std::vector<std::thread> workers;
    for (int r = 1; r < n; ++r) workers.emplace_back(body, r);
    body(0);
    for (auto& t : workers) t.join();
    (void)hipSetDevice(h->device);
    int worst = TLSQ_OK;
    int first_bad = -1;
    for (int r = 0; r < n; ++r) {
        if (st[(size_t)r] < 0) {
            // (the ranks the failure took down report TLSQ_ERR_COMM: the root cause is the first status that is not)
            if (first_bad < 0 || (st[(size_t)first_bad] == TLSQ_ERR_COMM && st[(size_t)r] != TLSQ_ERR_COMM)) first_bad = r;
        } else {
            worst = std::max(worst, st[(size_t)r]);
        }
    }
    if (first_bad >= 0) {
        if (first_bad > 0) h->err = "rank " + std::to_string(first_bad) + ": " + hs[(size_t)first_bad]->err;
        return st[(size_t)first_bad];
    }
    return worst;
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}


// Stream-ordered upload of a small host array through the pinned ring: no host synchronisation, and `src` may be
// freed or overwritten as soon as this returns.  A slot is only reused after the ring has wrapped, and wrapping
// synchronises the stream first.
int second_stream(Handle* h) {
    if (h->stream_b) return TLSQ_OK;
    // (highest priority: its workgroups are few and large - they should get a CU as soon as one has room)
    int least = 0, greatest = 0;
    const bool no_prio = dev_is(DEV_OVERLAP_NOPRIO, '1');
    const char* cp = dev_get(DEV_CERT_PRIO);   // (experiment: "low" / "normal" instead of the highest priority)
    const bool want_low = cp && cp[0] == 'l', want_normal = cp && cp[0] == 'n';
    if (!no_prio && !want_normal && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
        hipStreamCreateWithPriority(&h->stream_b, hipStreamNonBlocking, want_low ? least : greatest) != hipSuccess) {
        (void)hipGetLastError();
        h->stream_b = nullptr;
    }
    if (!h->stream_b) TLSQ_HIP(h, hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking));
    for (auto& e : h->ev_b) TLSQ_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return TLSQ_OK;
}

int upload_async(Handle* h, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return TLSQ_OK;
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (need > h->up_bytes) {   // never the case for the index lists this is meant for
        TLSQ_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        return TLSQ_OK;
    }
    if (h->up_off + need > h->up_bytes) {
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        h->up_off = 0;
    }
    char* stage = reinterpret_cast<char*>(h->up_ring) + h->up_off;
    memcpy(stage, src, bytes);
    h->up_off += need;
    TLSQ_HIP(h, hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, h->stream));
    return TLSQ_OK;
}

}  // namespace tlsq

using namespace tlsq;

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* tlsq_version(void) { return "tlsq-hip 0.1.0 (gfx950)"; }

int tlsq_dev_set(const char* name, const char* value) { return tlsq::dev_set(name, value); }

void tlsq_rpca_opts_default(tlsq_rpca_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->lambda = std::numeric_limits<double>::quiet_NaN();
    o->tol = std::numeric_limits<double>::quiet_NaN();
    o->rho = std::numeric_limits<double>::quiet_NaN();
    o->maxrank = 0;
    o->iters = 0;
    o->nukeA = 1;
    o->svd_mode = TLSQ_SVD_FULL;
    o->opnorm_mode = TLSQ_OPNORM_EXACT;
    o->opnorm_mvps = 10;
    o->memory = TLSQ_MEM_HOST;
}

int tlsq_create(int device_id, tlsq_handle* out) {
    if (!out) return TLSQ_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TLSQ_ERR_HIP;  // no GPU: fail loudly
    dev_load_env();
    if (device_id < 0 || device_id >= ndev) return TLSQ_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return TLSQ_ERR_HIP;
    tlsq_handle h = new (std::nothrow) tlsq_handle_s();
    if (!h) return TLSQ_ERR_OOM;
    h->device = device_id;
    h->ws.resize(WS_COUNT);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return TLSQ_ERR_HIP;
    }
    for (auto& e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete h;
            return TLSQ_ERR_HIP;
        }
    h->pinned_bytes = 1 << 16;
    if (hipHostMalloc(&h->pinned, h->pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        delete h;
        return TLSQ_ERR_OOM;
    }
    h->up_bytes = 1 << 18;
    if (hipHostMalloc(&h->up_ring, h->up_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipHostFree(h->pinned);
        delete h;
        return TLSQ_ERR_OOM;
    }
    // optional: without it the read-backs go through copy + synchronise as before
    void* mb = nullptr;
    const size_t mb_bytes = 32768;
    if (hipHostMalloc(&mb, mb_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        void* mbd = nullptr;
        if (hipHostGetDevicePointer(&mbd, mb, 0) == hipSuccess && mbd) {
            memset(mb, 0, mb_bytes);
            h->mailbox = (double*)mb;
            h->mailbox_bytes = mb_bytes;
            h->mailbox_dev = (double*)mbd;
        } else {
            (void)hipHostFree(mb);
        }
    }
    (void)hipGetLastError();
    *out = h;
    return TLSQ_OK;
}

int tlsq_create_multi(int ngpus, const int* device_ids, tlsq_handle* out) {
    if (!out || ngpus < 1) return TLSQ_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TLSQ_ERR_HIP;   // no GPU: fail loudly
    if (ngpus > 64) return TLSQ_ERR_ARG;
    std::vector<int> devs((size_t)ngpus);
    // the same device twice: a loop-back group (see LocalGroup), device ids must then be given.  FORCE_LOCALGROUP=1 gives
    // distinct devices the host-staged collectives as well (a debugging aid: takes RCCL out of the picture)
    bool repeats = dev_is(DEV_FORCE_LOCALGROUP, '1');
    for (int r = 0; r < ngpus; ++r) {
        devs[(size_t)r] = device_ids ? device_ids[r] : r;
        if (devs[(size_t)r] < 0 || devs[(size_t)r] >= ndev) return TLSQ_ERR_ARG;
        for (int q = 0; q < r; ++q)
            if (devs[(size_t)q] == devs[(size_t)r]) repeats = true;
    }
    if (!repeats && (!rccl_load() || !g_rccl.CommInitAll)) return TLSQ_ERR_COMM;
    std::vector<tlsq_handle> hs((size_t)ngpus, nullptr);
    auto undo = [&]() {
        for (auto x : hs)
            if (x) tlsq_destroy(x);
    };
    for (int r = 0; r < ngpus; ++r) {
        const int st = tlsq_create(devs[(size_t)r], &hs[(size_t)r]);
        if (st != TLSQ_OK) {
            undo();
            return st;
        }
    }
    std::vector<ncclComm_t> comms((size_t)ngpus, nullptr);
    std::shared_ptr<LocalGroup> lg;
    if (repeats) {
        lg = std::make_shared<LocalGroup>();
        lg->n = ngpus;
        lg->slot.resize((size_t)ngpus);
    } else if (g_rccl.CommInitAll(comms.data(), ngpus, devs.data()) != ncclSuccess) {
        undo();
        return TLSQ_ERR_COMM;
    }
    auto gf = std::make_shared<std::atomic<bool>>(false);
    for (int r = 0; r < ngpus; ++r) {
        hs[(size_t)r]->multi_comm = new Comm();
        hs[(size_t)r]->multi_comm->group_failed = gf;
        hs[(size_t)r]->multi_comm->comm = comms[(size_t)r];
        hs[(size_t)r]->multi_comm->local = lg;
        hs[(size_t)r]->multi_comm->local_rank = r;
        hs[(size_t)r]->multi_n = ngpus;
        hs[(size_t)r]->multi_rank = r;
    }
    for (int r = 1; r < ngpus; ++r) hs[0]->subs.push_back(hs[(size_t)r]);
    (void)hipSetDevice(devs[0]);
    *out = hs[0];
    return TLSQ_OK;
}

int tlsq_ngpus(tlsq_handle h) { return h ? h->multi_n : 0; }

int tlsq_destroy(tlsq_handle h) {
    if (!h) return TLSQ_OK;
    for (Handle* sub : h->subs) tlsq_destroy(static_cast<tlsq_handle>(sub));
    h->subs.clear();
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->multi_comm) {
        if (h->multi_comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->multi_comm->comm);
        delete h->multi_comm;
        h->multi_comm = nullptr;
    }
    tlsq_comm_destroy(h);
    for (auto& b : h->ws)
        if (b.p) (void)hipFree(b.p);
    stager_destroy(h);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->up_ring) (void)hipHostFree(h->up_ring);
    if (h->mailbox) (void)hipHostFree(h->mailbox);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : h->ev_b)
        if (e) (void)hipEventDestroy(e);
    if (h->stream_b) (void)hipStreamDestroy(h->stream_b);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return TLSQ_OK;
}

const char* tlsq_last_error(tlsq_handle h) { return h ? h->err.c_str() : "null handle"; }
void* tlsq_stream(tlsq_handle h) { return h ? (void*)h->stream : nullptr; }
int tlsq_synchronize(tlsq_handle h) {
    TLSQ_TRY(check_handle(h));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    return TLSQ_OK;
}

int tlsq_comm_unique_id(unsigned char id[TLSQ_UNIQUE_ID_BYTES]) {
    if (!id) return TLSQ_ERR_ARG;
    if (!rccl_load()) return TLSQ_ERR_COMM;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return TLSQ_ERR_COMM;
    static_assert(sizeof(u) == TLSQ_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, TLSQ_UNIQUE_ID_BYTES);
    return TLSQ_OK;
}

int tlsq_comm_init(tlsq_handle h, int nranks, int rank, const unsigned char id[TLSQ_UNIQUE_ID_BYTES]) {
    TLSQ_TRY(check_handle(h));
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) return set_err(h, TLSQ_ERR_ARG, "bad comm args");
    if (h->multi_comm)
        return set_err(h, TLSQ_ERR_ARG, "tlsq_comm_init: this handle is a multi-GPU group (tlsq_create_multi) and shards by "
                                        "itself; use one plain handle per process for a communicator of your own");
    tlsq_comm_destroy(h);
    // a single rank needs no communicator (TLSQ_FORCE_COMM=1 creates one anyway: exercises the RCCL path on one GPU)
    const char* force = dev_get(DEV_FORCE_COMM);
    if (nranks == 1 && !(force && force[0] == '1')) {
        h->nranks = 1;
        h->rank = 0;
        return TLSQ_OK;
    }
    if (!rccl_load()) return set_err(h, TLSQ_ERR_COMM, "cannot load librccl.so: %s", dlerror());
    TLSQ_HIP(h, hipSetDevice(h->device));
    ncclUniqueId u;
    memcpy(&u, id, TLSQ_UNIQUE_ID_BYTES);
    h->comm = new Comm();
    ncclResult_t r = g_rccl.CommInitRank(&h->comm->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete h->comm;
        h->comm = nullptr;
        return set_err(h, TLSQ_ERR_COMM, "ncclCommInitRank failed: %s",
                       g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    }
    h->nranks = nranks;
    h->rank = rank;
    return TLSQ_OK;
}

int tlsq_comm_size(tlsq_handle h, int* nranks) {
    TLSQ_TRY(check_handle(h));
    if (!nranks) return set_err(h, TLSQ_ERR_ARG, "comm_size: null argument");
    *nranks = 1;
    const Comm* c = h->comm ? h->comm : h->multi_comm;
    if (!c) return TLSQ_OK;
    if (c->local) {
        *nranks = c->local->n;
        return TLSQ_OK;
    }
    if (!c->comm || !g_rccl.CommCount) return set_err(h, TLSQ_ERR_COMM, "comm_size: no live RCCL communicator");
    int n = 0;
    TLSQ_NCCL(h, g_rccl.CommCount(c->comm, &n));
    *nranks = n;
    return TLSQ_OK;
}

int tlsq_comm_destroy(tlsq_handle h) {
    if (!h) return TLSQ_OK;
    if (h->comm) {
        if (h->comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm->comm);
        delete h->comm;
        h->comm = nullptr;
    }
    h->nranks = 1;
    h->rank = 0;
    return TLSQ_OK;
}


}  // extern "C"
