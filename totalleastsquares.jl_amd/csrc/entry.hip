// Entry of the rpca calls (both precisions): staging of the caller's memory (host or device pointers, any leading dimension)
// around rpca_core (solver.hip), wide problems (M < N) on the transposed panel, the single-process multi-GPU group
// (tlsq_create_multi: contiguous row blocks, one worker thread per GPU), and the host-side q x q partition solve of tls!
// (src/TotalLeastSquares.jl:65-69).  Split from solver.hip in round 5: nothing here takes part in the ALM loop itself.
#include <algorithm>
#include <cmath>
#include <functional>
#include <thread>
#include <vector>

#include "internal.hpp"

namespace tlsq {

// solve X * V22 = -V21 for X (n x q); V = Vt' where Vt is (ncols x ncols, ldVt) — TotalLeastSquares.jl:65-69
int tls_partition_solve(const double* Vt, int64_t ncols, int64_t ldVt, int64_t n, double* x,
                               int64_t ldx) {
    const int64_t q = ncols - n;
    if (n <= 0 || q <= 0) return TLSQ_ERR_ARG;
    // V[i][j] = Vt[j + i*ldVt].  V21 = V[0:n, n:], V22 = V[n:, n:]
    // X V22 = -V21  <=>  V22' X' = -V21'.  Build M = V22' (q x q): M[a][b] = V22[b][a] = V[n+b][n+a] = Vt[(n+a) + (n+b)*ldVt]
    std::vector<double> Mq((size_t)q * q), rhs((size_t)q * n);
    for (int64_t a = 0; a < q; ++a)
        for (int64_t b = 0; b < q; ++b) Mq[a * q + b] = Vt[(n + a) + (n + b) * ldVt];
    // rhs[a][i] = -V21'[a][i] = -V21[i][a] = -V[i][n+a] = -Vt[(n+a) + i*ldVt]
    for (int64_t a = 0; a < q; ++a)
        for (int64_t i = 0; i < n; ++i) rhs[a * n + i] = -Vt[(n + a) + i * ldVt];
    // LU with partial pivoting on Mq (row-major), applied to rhs
    for (int64_t c = 0; c < q; ++c) {
        int64_t piv = c;
        double best = std::fabs(Mq[c * q + c]);
        for (int64_t r2 = c + 1; r2 < q; ++r2)
            if (std::fabs(Mq[r2 * q + c]) > best) best = std::fabs(Mq[r2 * q + c]), piv = r2;
        if (piv != c) {
            for (int64_t b = 0; b < q; ++b) std::swap(Mq[c * q + b], Mq[piv * q + b]);
            for (int64_t i = 0; i < n; ++i) std::swap(rhs[c * n + i], rhs[piv * n + i]);
        }
        const double pv = Mq[c * q + c];
        for (int64_t r2 = c + 1; r2 < q; ++r2) {
            const double f = Mq[r2 * q + c] / pv;
            if (f == 0.0) continue;
            for (int64_t b = c; b < q; ++b) Mq[r2 * q + b] -= f * Mq[c * q + b];
            for (int64_t i = 0; i < n; ++i) rhs[r2 * n + i] -= f * rhs[c * n + i];
        }
    }
    for (int64_t c = q - 1; c >= 0; --c) {
        for (int64_t i = 0; i < n; ++i) {
            double v = rhs[c * n + i];
            for (int64_t b = c + 1; b < q; ++b) v -= Mq[c * q + b] * rhs[b * n + i];
            rhs[c * n + i] = v / Mq[c * q + c];
        }
    }
    // rhs = X' (q x n)  ->  x[i + a*ldx] = X[i][a]
    for (int64_t a = 0; a < q; ++a)
        for (int64_t i = 0; i < n; ++i) x[i + a * ldx] = rhs[a * n + i];
    return TLSQ_OK;
}

// ------------------------------------------------------------------------------------------------
// rpca entry (both precisions): staging of caller memory, M < N handled on the transposed problem
// ------------------------------------------------------------------------------------------------
// rpca on a single-process multi-GPU group (tlsq_create_multi): host matrices, contiguous row blocks, one worker per
// GPU.  Every rank runs the ordinary row-sharded entry on its block of the caller's arrays (column-major with the
// caller's leading dimensions, so a row block is just an offset pointer: the strided 2-D copies of rpca_entry do the
// scatter and the gather).  All ranks get structurally identical requests - history arrays, an on_iter hook, S / Vt
// buffers - because those requests steer which collectives a rank enters.
static void noop_on_iter(int64_t, double, int64_t, void*) {}

template <typename T>
static int rpca_multi(tlsq_handle h, const T* D, int64_t M, int64_t N, int64_t ldD, const tlsq_rpca_opts* opts, T* A,
                      int64_t ldA, T* E, int64_t ldE, T* U, int64_t ldU, T* S, T* Vt, int64_t ldVt, int64_t* sv,
                      tlsq_rpca_info* info) {
    const int n = h->multi_n;
    const int64_t d = std::min(M, N);
    tlsq_rpca_opts base;
    if (opts) base = *opts; else tlsq_rpca_opts_default(&base);
    base.m_global = M;
    base.memory = TLSQ_MEM_HOST;
    std::vector<tlsq_rpca_opts> ro((size_t)n, base);
    std::vector<tlsq_rpca_info> ri((size_t)n);
    std::vector<std::vector<double>> ch((size_t)n);
    std::vector<std::vector<int64_t>> sh((size_t)n);
    std::vector<std::vector<T>> Sr((size_t)n), Vr((size_t)n);
    std::vector<int64_t> svr((size_t)n, 0);
    for (int r = 0; r < n; ++r) {
        memset(&ri[(size_t)r], 0, sizeof(tlsq_rpca_info));
        if (r == 0) {
            if (info) ri[0] = *info;
        } else {
            if (base.on_iter) ro[(size_t)r].on_iter = noop_on_iter;   // the caller's hook runs on the calling thread only
            if (info && info->cost_hist) {
                ch[(size_t)r].resize((size_t)std::max<int64_t>(info->hist_capacity, 1));
                ri[(size_t)r].cost_hist = ch[(size_t)r].data();
            }
            if (info && info->svp_hist) {
                sh[(size_t)r].resize((size_t)std::max<int64_t>(info->hist_capacity, 1));
                ri[(size_t)r].svp_hist = sh[(size_t)r].data();
            }
            ri[(size_t)r].hist_capacity = info ? info->hist_capacity : 0;
            if (S) Sr[(size_t)r].resize((size_t)d);
            if (Vt) Vr[(size_t)r].resize((size_t)d * N);
        }
    }
    const int st = multi_run(h, [&](Handle* hr, int r, int nr) -> int {
        const int64_t bs = M / nr, rem = M % nr;
        const int64_t lo = r * bs + std::min<int64_t>(r, rem), rows = bs + (r < rem ? 1 : 0);
        return rpca_entry<T>(static_cast<tlsq_handle>(hr), D + lo, rows, N, ldD, &ro[(size_t)r], A + lo, ldA, E + lo, ldE,
                             U ? U + lo : nullptr, ldU, S ? (r == 0 ? S : Sr[(size_t)r].data()) : nullptr,
                             Vt ? (r == 0 ? Vt : Vr[(size_t)r].data()) : nullptr, r == 0 ? ldVt : d, &svr[(size_t)r],
                             &ri[(size_t)r]);
    });
    if (info) *info = ri[0];
    if (sv) *sv = svr[0];
    return st;
}

template <typename T>
int rpca_entry(tlsq_handle h, const T* D, int64_t M, int64_t N, int64_t ldD, const tlsq_rpca_opts* opts,
                      T* A, int64_t ldA, T* E, int64_t ldE, T* U, int64_t ldU, T* S, T* Vt, int64_t ldVt,
                      int64_t* sv, tlsq_rpca_info* info) {
    TLSQ_TRY(check_handle(h));
    if (!D || !A || !E || M <= 0 || N <= 0 || ldD < M || ldA < M || ldE < M)
        return set_err(h, TLSQ_ERR_ARG, "rpca: bad argument (M=%lld N=%lld)", (long long)M, (long long)N);
    if (is_multi_call(h)) {
        const bool dev_mem = opts && opts->memory == TLSQ_MEM_DEVICE;
        if (dev_mem)
            return set_err(h, TLSQ_ERR_UNSUPPORTED, "rpca: a multi-GPU handle takes host matrices (device pointers belong "
                           "to one GPU; use one handle per GPU with tlsq_comm_init for device-resident shards)");
        // tall problems with enough rows per GPU are row-sharded (a caller's svd / opnorm hook included: the shards are
        // gathered for every call of the hook, which runs on the calling thread - rank 0's - solver.hip); anything else runs on
        // the first GPU alone
        if (M >= N && M >= 32 * (int64_t)h->multi_n && (!opts || opts->m_global <= 0 || opts->m_global == M))
            return rpca_multi<T>(h, D, M, N, ldD, opts, A, ldA, E, ldE, U, ldU, S, Vt, ldVt, sv, info);
    }
    host_mark("entry");
    TLSQ_HIP(h, hipSetDevice(h->device));
    const double t0 = now_ms();
    reset_info(info);
    const double eps_t = (double)std::numeric_limits<T>::epsilon();
    const ResolvedOpts ro = resolve(opts, M, N, std::sqrt(eps_t));     // tol = sqrt(eps(real(T)))  (:160)
    const bool dev = opts && opts->memory == TLSQ_MEM_DEVICE;
    const size_t es = sizeof(T);
    const int64_t n = M * N;
    const int64_t d = std::min(ro.m_global, N);
    if (U && ldU < M) return set_err(h, TLSQ_ERR_ARG, "rpca: ldU < M");
    if (Vt && ldVt < d) return set_err(h, TLSQ_ERR_ARG, "rpca: ldVt < min(M,N)");
    // rpca is invariant under transposition (elementwise sweeps, singular-value thresholding, lambda =
    // 1/sqrt(max(M,N))).  A wide unsharded D is solved as its tall transpose: the Gram matrix is then M x M
    // and has no structurally-zero eigenvalues (accuracy of the Gram route).
    const bool transposed = (M < N) && ro.m_global == M && !h->comm;
    // the small-matrix solvers of this release keep their panels in LDS: the Gram dimension is limited
    if ((transposed ? M : N) > kGramMaxN)
        return set_err(h, TLSQ_ERR_UNSUPPORTED, "rpca: min(M,N) = %lld exceeds %lld, the largest Gram dimension of this "
                       "release", (long long)(transposed ? M : N), (long long)kGramMaxN);

    // The panels the MFMA kernels stream (Z, R, ...) inherit the row count of the working problem as their leading
    // dimension.  The Gram kernel reads 16-row (128-byte) segments of every column: when the leading dimension is
    // not a multiple of 16 every segment straddles two cache lines (measured 3x slower at 9,999,745 rows), and an
    // odd one also forces 8-byte loads.  So the row count is padded with zero rows to a multiple of 16 in private
    // panels — zero rows change nothing in the algorithm (lambda and d use the true size through m_global).
    const int64_t Mw = transposed ? N : M;          // rows of the working (tall) problem
    const int64_t Nw = transposed ? M : N;
    const bool has_cb = opts && (opts->svd_mode == TLSQ_SVD_CALLBACK || opts->opnorm_mode == TLSQ_OPNORM_CALLBACK);
    // soft_hankel! would see the extra rows, and so would a caller's svd / opnorm hook: keep the exact shape there
    const bool pad = (Mw % 16 != 0) && !ro.hankel && !has_cb;
    const int64_t Mp = pad ? (Mw + 15) / 16 * 16 : Mw;
    const size_t nw = (size_t)Mp * Nw;

    const T* dD = D;
    T *dA = A, *dE = E, *dU = U;
    void* p;
    double th = now_ms();
    const bool priv = !dev || transposed || pad;    // work on private panels?
    if (!dev || ldD != M) {
        TLSQ_TRY(ws_get(h, WS_D, (size_t)n * es, &p));
        if (dev) {
            TLSQ_TRY(copy2d(h, p, M, D, ldD, M, N, es, hipMemcpyDeviceToDevice));
        } else {
            // the caller's (pageable) matrix: pinned slots on worker threads instead of the runtime's one-thread bounce buffer
            const StageJob up{p, M, D, ldD, M, N, es, true};
            TLSQ_TRY(staged_copy(h, &up, 1));
        }
        dD = (const T*)p;
    }
    if (!dev || ldA != M) {
        TLSQ_TRY(ws_get(h, WS_A, (size_t)n * es, &p));
        dA = (T*)p;
    }
    if (!dev || ldE != M) {
        TLSQ_TRY(ws_get(h, WS_E, (size_t)n * es, &p));
        dE = (T*)p;
    }
    if (U && !transposed && !pad && (!dev || ldU != M)) {
        TLSQ_TRY(ws_get(h, WS_AUX2, (size_t)M * d * es, &p));
        dU = (T*)p;
    }
    (void)priv;
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    if (info) info->ms_h2d = now_ms() - th;

    // S / Vt are small: always produced on the host in fp64, then converted / copied to the caller's memory
    std::vector<double> hS((size_t)(S ? d : 0)), hVt((size_t)(Vt ? d * N : 0));
    int status;
    // Host-pointer call that also returns s: A and E are final when the loop ends, 10+ ms before U, S, Vt are - their 164 MB go
    // back to the caller's memory on the staging workers WHILE the decomposition of the last Z runs (a thread of its own drives
    // the staged copy; the solver thread keeps queueing kernels).
    bool vt_on_device = false;   // rpca_core has written Vt to device memory itself (ResolvedOpts::vt_dev)
    std::thread ae_thread;
    int ae_status = TLSQ_OK;
    bool ae_sent = false;
    if (!transposed && !pad) {
        ResolvedOpts ro2 = ro;
        // (the download thread shares the handle with the solver thread: it only ever takes the pinned-slot path of
        //  staged_copy - below 1 MB that function would use the handle's own stream - the stager exists before the thread
        //  does, set_err is serialised (runtime.hip), and a thread that cannot be created means the copy happens after the
        //  decomposition as in a call without `s`; the guard joins on every way out of this scope)
        struct JoinGuard {
            std::thread& t;
            ~JoinGuard() {
                if (t.joinable()) t.join();
            }
        } ae_guard{ae_thread};
        const std::function<void()> send_ae = [&]() {
            if (dev || dA == A || dE == E) return;
            if ((size_t)n * es < ((size_t)1 << 20)) return;
            if (staged_copy(h, nullptr, 0) < 0) return;   // (creates the staging workers' streams on this thread)
            try {
                ae_thread = std::thread([&]() {
                    StageJob down[2] = {StageJob{A, ldA, dA, M, M, N, es, false}, StageJob{E, ldE, dE, M, M, N, es, false}};
                    ae_status = staged_copy(h, down, 2);
                });
                ae_sent = true;
            } catch (...) {
                ae_sent = false;
            }
        };
        ro2.ae_final = &send_ae;
        if (Vt) {   // Vt straight to device memory: the caller's own with TLSQ_MEM_DEVICE, a workspace buffer (one copy out) otherwise
            if (dev) {
                ro2.vt_dev = Vt;
                ro2.vt_ld = ldVt;
            } else {
                void* vtb;
                TLSQ_TRY(ws_get(h, WS_VTOUT, (size_t)d * N * es, &vtb));
                ro2.vt_dev = vtb;
                ro2.vt_ld = d;
            }
            ro2.vt_written = &vt_on_device;
        }
        status = rpca_core<T>(h, dD, M, N, ro2, opts, dA, dE, U ? dU : nullptr, S ? hS.data() : nullptr,
                              Vt ? hVt.data() : nullptr, d, sv, info);
        if (vt_on_device && !dev) TLSQ_TRY(copy2d(h, Vt, ldVt, ro2.vt_dev, d, d, N, es, hipMemcpyDeviceToHost));
        if (ae_thread.joinable()) ae_thread.join();
        (void)hipSetDevice(h->device);
        if (status < 0) return status;
        if (ae_status < 0) return ae_status;
    } else {
        // working copies: Dw (Mp x Nw) = D or D', zero pad row; Aw, Ew results; Uw (Mp x d) left vectors of Zw
        void *Dw, *Aw, *Ew, *Uw = nullptr;
        TLSQ_TRY(ws_get(h, WS_DT, nw * es, &Dw));
        TLSQ_TRY(ws_get(h, WS_AT, nw * es, &Aw));
        TLSQ_TRY(ws_get(h, WS_ET, nw * es, &Ew));
        const bool need_Uw = transposed ? (Vt != nullptr) : (U != nullptr);
        if (need_Uw) TLSQ_TRY(ws_get(h, WS_UT, (size_t)Mp * d * es, &Uw));
        if (pad) TLSQ_HIP(h, hipMemsetAsync(Dw, 0, nw * es, h->stream));
        if (transposed) TLSQ_TRY(launch_transpose<T>(h, dD, M, M, N, (T*)Dw, Mp));
        else TLSQ_TRY(copy2d(h, Dw, Mp, dD, M, M, N, es, hipMemcpyDeviceToDevice));
        ResolvedOpts rw = ro;
        rw.m_global = transposed ? N : ro.m_global;
        std::vector<double> hVtW((size_t)d * Nw);                      // right vectors of the working problem
        status = rpca_core<T>(h, (const T*)Dw, Mp, Nw, rw, opts, (T*)Aw, (T*)Ew, need_Uw ? (T*)Uw : nullptr,
                              S ? hS.data() : nullptr, (transposed ? (U != nullptr) : (Vt != nullptr)) ? hVtW.data() : nullptr,
                              d, sv, info);
        if (status < 0) return status;
        if (transposed) {
            TLSQ_TRY(launch_transpose<T>(h, (const T*)Aw, Mp, N, M, dA, M));
            TLSQ_TRY(launch_transpose<T>(h, (const T*)Ew, Mp, N, M, dE, M));
            if (Vt) {   // Vt (d x N) = Uw^T  (Uw is N(+1) x d, ld Mp)
                std::vector<T> hu((size_t)Mp * d);
                TLSQ_HIP(h, hipMemcpyAsync(hu.data(), Uw, (size_t)Mp * d * es, hipMemcpyDeviceToHost, h->stream));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                for (int64_t pcol = 0; pcol < d; ++pcol)
                    for (int64_t j = 0; j < N; ++j) hVt[pcol + j * d] = (double)hu[j + pcol * Mp];
            }
            if (U) {    // U (M x d) = VtW^T
                std::vector<T> hu((size_t)M * d);
                for (int64_t pcol = 0; pcol < d; ++pcol)
                    for (int64_t i = 0; i < M; ++i) hu[i + pcol * M] = (T)hVtW[pcol + i * d];
                TLSQ_TRY(copy2d(h, U, ldU, hu.data(), M, M, d, es, dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            }
        } else {
            TLSQ_TRY(copy2d(h, dA, M, Aw, Mp, M, N, es, hipMemcpyDeviceToDevice));
            TLSQ_TRY(copy2d(h, dE, M, Ew, Mp, M, N, es, hipMemcpyDeviceToDevice));
            if (Vt) hVt = hVtW;
            if (U) {
                TLSQ_TRY(copy2d(h, U, ldU, Uw, Mp, M, d, es, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            }
        }
    }

    th = now_ms();
    const hipMemcpyKind back = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (dev) {
        if (dA != A) TLSQ_TRY(copy2d(h, A, ldA, dA, M, M, N, es, back));
        if (dE != E) TLSQ_TRY(copy2d(h, E, ldE, dE, M, M, N, es, back));
        if (U && !transposed && !pad && dU != U) TLSQ_TRY(copy2d(h, U, ldU, dU, M, M, d, es, back));
    } else {
        // A, E (and U) go back to the caller's memory together, pipelined through the pinned slots
        StageJob down[3];
        int nd = 0;
        if (dA != A && !ae_sent) down[nd++] = StageJob{A, ldA, dA, M, M, N, es, false};
        if (dE != E && !ae_sent) down[nd++] = StageJob{E, ldE, dE, M, M, N, es, false};
        if (U && !transposed && !pad && dU != U) down[nd++] = StageJob{U, ldU, dU, M, M, d, es, false};
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        TLSQ_TRY(staged_copy(h, down, nd));
    }
    std::vector<T> tS, tVt;
    if (S) {
        tS.resize((size_t)d);
        for (int64_t i = 0; i < d; ++i) tS[i] = (T)hS[i];
        TLSQ_HIP(h, hipMemcpyAsync(S, tS.data(), (size_t)d * es, dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost,
                                   h->stream));
    }
    if (Vt && !vt_on_device) {
        tVt.resize((size_t)d * N);
        for (size_t i = 0; i < tVt.size(); ++i) tVt[i] = (T)hVt[i];
        TLSQ_TRY(copy2d(h, Vt, ldVt, tVt.data(), d, d, N, es, dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
    }
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    if (info) {
        info->ms_d2h = now_ms() - th;
        info->ms_total = now_ms() - t0;
    }
    host_mark("return");
    host_trace_dump();
    return status;
}


// explicit instantiations used by api.hip
template int rpca_entry<double>(tlsq_handle, const double*, int64_t, int64_t, int64_t, const tlsq_rpca_opts*, double*,
                                int64_t, double*, int64_t, double*, int64_t, double*, double*, int64_t, int64_t*,
                                tlsq_rpca_info*);
template int rpca_entry<float>(tlsq_handle, const float*, int64_t, int64_t, int64_t, const tlsq_rpca_opts*, float*, int64_t,
                               float*, int64_t, float*, int64_t, float*, float*, int64_t, int64_t*, tlsq_rpca_info*);

}  // namespace tlsq
