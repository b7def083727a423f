// Pipelined pinned staging for HOST-pointer calls (the drop-in boundary: a Julia `Array` / numpy array is pageable memory).
//
// hipMemcpy on pageable memory stages through the runtime's own bounce buffer on one thread: ~12 GB/s measured for the 82 MB
// panels of a 20000 x 512 fp64 call, i.e. 21 ms of copies around a 10.7 ms solve.  Registering the caller's arrays for the
// call (hipHostRegister) pins page by page and costs more than the copy it saves, and a registration must not outlive the
// call (the host language's GC owns the memory).  So the library stages itself: a few worker threads, each with two pinned
// 4 MB slots and a stream of its own, move pieces of the matrices - memcpy between the caller's memory and a slot on the
// worker's core, DMA between the slot and HBM - so that PCIe (~50 GB/s) rather than one core's memcpy is the limit.
// Pieces are whole column groups (one hipMemcpy2DAsync when the device side has a leading dimension of its own) or row
// segments of one column when a column exceeds a slot (lowrankfilter-sized panels).
//
// Replaces the copy2d(..., hipMemcpyHostToDevice / DeviceToHost) calls of the host-mode entry points (solver.hip rpca_entry;
// reference call shape src/robustPCA.jl:156, :238 - D in, A, E and s.U out).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "internal.hpp"

namespace tlsq {

namespace {
constexpr size_t kSlotBytes = (size_t)4 << 20;
constexpr int kMaxWorkers = 8;

struct Piece {
    int job;
    int64_t c0, c1;   // columns [c0, c1) ...
    int64_t r0, r1;   // ... rows [r0, r1) of them (a row segment only when c1 == c0 + 1)
};
}  // namespace

struct Stager {
    int nworkers = 0;
    char* arena = nullptr;   // nworkers x 2 slots
    hipStream_t stream[kMaxWorkers] = {};
    hipEvent_t ev[kMaxWorkers][2] = {};
};

void stager_destroy(Handle* h) {
    Stager* s = h->stager;
    if (!s) return;
    for (int w = 0; w < s->nworkers; ++w) {
        if (s->stream[w]) (void)hipStreamDestroy(s->stream[w]);
        for (int k = 0; k < 2; ++k)
            if (s->ev[w][k]) (void)hipEventDestroy(s->ev[w][k]);
    }
    if (s->arena) (void)hipHostFree(s->arena);
    delete s;
    h->stager = nullptr;
}

static int stager_get(Handle* h, Stager** out) {
    if (h->stager) {
        *out = h->stager;
        return TLSQ_OK;
    }
    Stager* s = new (std::nothrow) Stager();
    if (!s) return set_err(h, TLSQ_ERR_OOM, "staging: out of host memory");
    h->stager = s;
    const unsigned hc = std::thread::hardware_concurrency();
    // (a multi-GPU group stages on every rank at once: fewer workers per rank there)
    int want = h->multi_n > 1 ? 3 : 6;
    if (hc > 0 && (int)hc - 1 < want) want = std::max(1, (int)hc - 1);
    s->nworkers = std::min(want, kMaxWorkers);
    if (hipHostMalloc((void**)&s->arena, (size_t)s->nworkers * 2 * kSlotBytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        s->arena = nullptr;
        stager_destroy(h);
        return set_err(h, TLSQ_ERR_OOM, "staging: cannot allocate %zu MB of pinned host memory",
                       ((size_t)want * 2 * kSlotBytes) >> 20);
    }
    for (int w = 0; w < s->nworkers; ++w) {
        hipError_t e = hipStreamCreateWithFlags(&s->stream[w], hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&s->ev[w][k], hipEventDisableTiming);
        if (e != hipSuccess) {   // (a half-built stager must not survive: later calls would reuse its missing streams)
            (void)hipGetLastError();
            stager_destroy(h);
            return set_err(h, TLSQ_ERR_HIP, "staging: cannot create the worker streams / events: %s", hipGetErrorString(e));
        }
    }
    *out = s;
    return TLSQ_OK;
}

// Copies every job (host <-> device, column-major with leading dimensions, in elements of esz bytes) and returns when all
// of them have landed.  Device-side sources must be complete before the call (synchronise the producing stream first);
// device-side destinations are complete - visible to kernels launched afterwards on any stream - on return.
int staged_copy(Handle* h, const StageJob* jobs, int njobs) {
    size_t total = 0;
    for (int j = 0; j < njobs; ++j) total += (size_t)std::max<int64_t>(jobs[j].rows, 0) * (size_t)std::max<int64_t>(jobs[j].cols, 0) * jobs[j].esz;
    if (njobs == 0) {   // (no job: make sure the staging workers' streams and pinned slots exist - rpca_entry calls this
                        //  before it hands copies to a thread of its own)
        Stager* s0 = nullptr;
        return stager_get(h, &s0);
    }
    if (total == 0) return TLSQ_OK;
    if (total < ((size_t)1 << 20)) {   // small: the plain copies
        for (int j = 0; j < njobs; ++j) {
            const StageJob& q = jobs[j];
            TLSQ_TRY(copy2d(h, q.dst, q.ldd, q.src, q.lds, q.rows, q.cols, q.esz, q.to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
        }
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        return TLSQ_OK;
    }
    Stager* s = nullptr;
    TLSQ_TRY(stager_get(h, &s));
    std::vector<Piece> pieces;
    for (int j = 0; j < njobs; ++j) {
        const StageJob& q = jobs[j];
        if (q.rows <= 0 || q.cols <= 0) continue;
        const size_t colb = (size_t)q.rows * q.esz;
        if (colb <= kSlotBytes) {
            const int64_t per = std::max<int64_t>(1, (int64_t)(kSlotBytes / colb));
            for (int64_t c = 0; c < q.cols; c += per) pieces.push_back(Piece{j, c, std::min(q.cols, c + per), 0, q.rows});
        } else {
            const int64_t rper = (int64_t)(kSlotBytes / q.esz);
            for (int64_t c = 0; c < q.cols; ++c)
                for (int64_t r = 0; r < q.rows; r += rper) pieces.push_back(Piece{j, c, c + 1, r, std::min(q.rows, r + rper)});
        }
    }
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    const int dev = h->device;
    auto work = [&](int w) {
        if (hipSetDevice(dev) != hipSuccess) {
            failed.store(1);
            return;
        }
        char* slot[2] = {s->arena + (size_t)(2 * w) * kSlotBytes, s->arena + (size_t)(2 * w + 1) * kSlotBytes};
        bool busy[2] = {false, false};
        // a device-to-host piece whose DMA is in flight: finished (event, then memcpy to the caller) one piece later
        const Piece* pend = nullptr;
        int pend_slot = 0;
        auto host_side = [&](const Piece& p, char* buf, bool to_slot) {
            const StageJob& q = jobs[p.job];
            const int64_t nr = p.r1 - p.r0;
            char* hp = (char*)(q.to_device ? const_cast<void*>(q.src) : q.dst);
            const int64_t ldh = q.to_device ? q.lds : q.ldd;
            if (ldh == nr && p.r0 == 0) {   // contiguous columns: one memcpy
                char* base = hp + (size_t)p.c0 * ldh * q.esz;
                const size_t nb = (size_t)(p.c1 - p.c0) * nr * q.esz;
                if (to_slot) memcpy(buf, base, nb); else memcpy(base, buf, nb);
                return;
            }
            for (int64_t c = p.c0; c < p.c1; ++c) {
                char* col = hp + ((size_t)c * ldh + (size_t)p.r0) * q.esz;
                char* b = buf + (size_t)(c - p.c0) * nr * q.esz;
                if (to_slot) memcpy(b, col, (size_t)nr * q.esz); else memcpy(col, b, (size_t)nr * q.esz);
            }
        };
        auto dma = [&](const Piece& p, char* buf) -> hipError_t {
            const StageJob& q = jobs[p.job];
            const int64_t nr = p.r1 - p.r0, nc = p.c1 - p.c0;
            char* dp = (char*)(q.to_device ? q.dst : const_cast<void*>(q.src));
            const int64_t ldv = q.to_device ? q.ldd : q.lds;
            char* dbase = dp + ((size_t)p.c0 * ldv + (size_t)p.r0) * q.esz;
            if (ldv == nr || nc == 1) {
                const size_t nb = (size_t)nc * nr * q.esz;
                return q.to_device ? hipMemcpyAsync(dbase, buf, nb, hipMemcpyHostToDevice, s->stream[w])
                                   : hipMemcpyAsync(buf, dbase, nb, hipMemcpyDeviceToHost, s->stream[w]);
            }
            return q.to_device ? hipMemcpy2DAsync(dbase, (size_t)ldv * q.esz, buf, (size_t)nr * q.esz, (size_t)nr * q.esz, (size_t)nc,
                                                  hipMemcpyHostToDevice, s->stream[w])
                               : hipMemcpy2DAsync(buf, (size_t)nr * q.esz, dbase, (size_t)ldv * q.esz, (size_t)nr * q.esz, (size_t)nc,
                                                  hipMemcpyDeviceToHost, s->stream[w]);
        };
        auto finish_pending = [&]() -> bool {
            if (!pend) return true;
            if (hipEventSynchronize(s->ev[w][pend_slot]) != hipSuccess) return false;
            host_side(*pend, slot[pend_slot], false);
            busy[pend_slot] = false;
            pend = nullptr;
            return true;
        };
        int turn = 0;
        for (;;) {
            if (failed.load()) break;
            const size_t i = next.fetch_add(1);
            if (i >= pieces.size()) break;
            const Piece& p = pieces[i];
            const int k = turn & 1;
            ++turn;
            if (pend && pend_slot == k && !finish_pending()) {
                failed.store(1);
                break;
            }
            if (busy[k]) {   // (an upload from this slot still in flight)
                if (hipEventSynchronize(s->ev[w][k]) != hipSuccess) {
                    failed.store(1);
                    break;
                }
                busy[k] = false;
            }
            if (jobs[p.job].to_device) {
                host_side(p, slot[k], true);
                if (dma(p, slot[k]) != hipSuccess || hipEventRecord(s->ev[w][k], s->stream[w]) != hipSuccess) {
                    failed.store(1);
                    break;
                }
                busy[k] = true;
            } else {
                if (dma(p, slot[k]) != hipSuccess || hipEventRecord(s->ev[w][k], s->stream[w]) != hipSuccess) {
                    failed.store(1);
                    break;
                }
                busy[k] = true;
                // the previous download (other slot) has had this piece's issue time to land: hand it to the caller now
                const Piece* mine = &p;
                if (pend && !finish_pending()) {
                    failed.store(1);
                    break;
                }
                pend = mine;
                pend_slot = k;
            }
        }
        if (!failed.load() && !finish_pending()) failed.store(1);
        if (hipStreamSynchronize(s->stream[w]) != hipSuccess) failed.store(1);
    };
    const int nw = (int)std::min<size_t>((size_t)s->nworkers, pieces.size());
    std::vector<std::thread> th;
    th.reserve((size_t)std::max(0, nw - 1));
    for (int w = 1; w < nw; ++w) th.emplace_back(work, w);
    work(0);
    for (auto& t : th) t.join();
    (void)hipSetDevice(h->device);
    if (failed.load()) {
        const hipError_t e = hipGetLastError();
        return set_err(h, TLSQ_ERR_HIP, "staged copy failed: %s", hipGetErrorString(e));
    }
    return TLSQ_OK;
}

}  // namespace tlsq
