// Declarations shared by runtime.hip, solver.hip and api.hip (the host side of libtlsqhip.so); the kernel launchers are
// declared in common.hpp.
#pragma once
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

#include "common.hpp"

namespace tlsq {

// ---- runtime.hip -------------------------------------------------------------------------------------------
// in-place reduction over the row shards of a handle with a communicator (no-op otherwise)
int comm_allreduce(Handle* h, double* dev, size_t count, ncclRedOp_t op);
int comm_allreduce_host_scalar(Handle* h, double* v, ncclRedOp_t op);
int comm_allreduce_host_vec(Handle* h, double* v, int count, ncclRedOp_t op);   // count <= 8
// recv (nranks * count doubles, rank-major) <- every rank's send (count doubles)
int comm_allgather(Handle* h, const double* send, double* recv, size_t count);
int copy2d(Handle* h, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t rows, int64_t cols, size_t esz,
           hipMemcpyKind kind);
// ---- staging.hip: large host <-> device transfers of host-pointer calls through pinned slots on worker threads -------
struct StageJob {
    void* dst;
    int64_t ldd;
    const void* src;
    int64_t lds;
    int64_t rows, cols;
    size_t esz;
    bool to_device;   // host -> device (src is the caller's memory) or device -> host (dst is)
};
int staged_copy(Handle* h, const StageJob* jobs, int njobs);
void stager_destroy(Handle* h);
double now_ms();
int ws_poison_all(Handle* h);   // WS_POISON=1: refill every workspace slot with 0xFF bytes (start of a solve)
// Single-process multi-GPU group: runs fn(rank handle, rank, nranks) on every GPU of the group concurrently - rank 0 on
// the calling thread, the others on worker threads that never call back into the host language - with each rank's
// communicator attached for the duration.  Returns the first fatal status (its message is copied to h), else the
// largest non-fatal one.
int multi_run(Handle* h, const std::function<int(Handle*, int, int)>& fn);
inline bool is_multi_call(const Handle* h) { return h->multi_comm != nullptr && !h->in_multi; }
// stream-ordered upload of a small host array through the pinned ring (src may be reused at once)
int upload_async(Handle* h, void* dst, const void* src, size_t bytes);
int second_stream(Handle* h);   // creates Handle::stream_b / ev_b on first use

inline int check_handle(tlsq_handle h) { return h ? TLSQ_OK : TLSQ_ERR_ARG; }

inline void reset_info(tlsq_rpca_info* info) {
    if (!info) return;
    double* ch = info->cost_hist;
    int64_t* sh = info->svp_hist;
    int64_t cap = info->hist_capacity;
    memset(info, 0, sizeof(*info));
    info->cost_hist = ch;
    info->svp_hist = sh;
    info->hist_capacity = cap;
}

// Phase marks of one ALM iteration: HIP events on the handle's stream, two banks so that an iteration's marks can be
// read back while the next iteration is already queued (no stream-wide synchronisation just for the timing).
struct PhaseTimer {
    Handle* h;
    bool on;
    // full: every phase boundary records an event.  Otherwise only the sweep kernels are bracketed (windows 0 and 4:
    // what the HBM roofline needs) - every recorded event is a barrier packet the command processor works through
    // between two kernels (~5.7 us each at C2 size: six of them were 7 % of an iteration).
    bool full;
    // light mode: the essential marks are recorded only while `sample` is set (rpca_core sets it for the first shrink and the
    // sweep of every few iterations: two packets per bracketed sweep are 3 % of a C2 iteration)
    bool sample = true;
    int bank = 0;
    int n[2] = {0, 0};
    int slot[2][16] = {};   // event behind each mark (an empty phase reuses the previous mark's event; -1: none)
    PhaseTimer(Handle* hh, bool enable, bool all_phases) : h(hh), on(enable), full(all_phases) {}
    // empty = nothing was queued since the previous mark: no event is recorded, the phase simply gets zero time.
    // essential = one of the marks around a sweep kernel (recorded in the light mode too)
    void mark(bool empty = false, bool essential = false) {
        if (!on || n[bank] >= 16) return;
        const int i = n[bank]++;
        if (!full && !(essential && sample)) empty = true;
        if (empty) {
            slot[bank][i] = i > 0 ? slot[bank][i - 1] : -1;
            return;
        }
        slot[bank][i] = bank * 16 + i;
        (void)hipEventRecord(h->ev[slot[bank][i]], h->stream);
    }
    // adds elapsed(mark i, mark i+1) of bank b to *acc[i]; the bank's last event must have completed
    void collect_bank(int b, double** acc) {
        if (on && n[b] > 0) {
            int last = -1;
            for (int i = 0; i < n[b]; ++i)
                if (slot[b][i] >= 0) last = slot[b][i];
            if (last >= 0) (void)hipEventSynchronize(h->ev[last]);
            for (int i = 0; i + 1 < n[b]; ++i) {
                if (slot[b][i] < 0 || slot[b][i + 1] < 0 || slot[b][i] == slot[b][i + 1]) continue;
                if (!full && i != 0 && i != 4) continue;   // light mode: only the sweep windows are bracketed properly
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, h->ev[slot[b][i]], h->ev[slot[b][i + 1]]) == hipSuccess && acc[i])
                    *acc[i] += ms;
            }
        }
        n[b] = 0;
    }
    // read the PREVIOUS iteration's marks (long completed).  Best called while the GPU has plenty queued: the
    // event queries cost host time
    void collect_previous(double** acc) { collect_bank(bank ^ 1, acc); }
    // end of an iteration: switch banks (collecting the other one first if nobody did)
    void next_iteration(double** acc) {
        collect_bank(bank ^ 1, acc);
        bank ^= 1;
    }
    // after the loop: whatever is still pending
    void finish(double** acc) {
        collect_bank(bank ^ 1, acc);
        collect_bank(bank, acc);
    }
};


// ---- solver.hip --------------------------------------------------------------------------------------------
// Largest Gram dimension the LDS-resident full Jacobi solvers handle.  Above it ("large mode") rpca relies on the
// certified subspace iteration alone; see rpca_core.
constexpr int64_t kFullEigMaxN = 2048;
constexpr int64_t kGramMaxN = 65536;   // (round 4: beyond 16384 everything runs through the operator form - no N x N matrix exists from N = 8192 on)
// largest N for which the complete returned SVD is computed after a large-mode loop (one-sided Jacobi on R' with at
// least two resident columns per workgroup: 2 * 2 * N * 8 bytes of LDS)
constexpr int64_t kReturnedSvdMaxN = 4608;


template <typename T>
struct Prec {
    static constexpr int f32 = std::is_same<T, float>::value ? 1 : 0;
};


// Decomposition of the Gram of Z: V (device, N x ncols, ld N), sigma (host, per column of V), order (descending)
struct SmallSvd {
    std::vector<double> sigma;   // per column of V
    std::vector<int32_t> order;  // column indices sorted by sigma descending
    int64_t ncols = 0;           // N for a full decomposition, p for a subspace one
};

inline void sort_desc(SmallSvd& s) {
    s.order.resize(s.sigma.size());
    std::iota(s.order.begin(), s.order.end(), 0);
    std::stable_sort(s.order.begin(), s.order.end(),
                     [&](int32_t a, int32_t b) { return s.sigma[a] > s.sigma[b]; });
}


struct ResolvedOpts {
    double lambda, tol, rho;
    int64_t maxrank, iters, m_global;
    bool nonnegA, nonnegE, hankel, nukeA;
    // D is the (row-padded) Hankel matrix of this device vector, D[i, j] = hankel_y[i + j] for i < hankel_K: the big
    // fused sweep reads the vector instead of the panel (set by lowrankfilter; fp64, one channel, lag 1)
    const void* hankel_y = nullptr;
    int64_t hankel_K = 0;
    // ... in general D[k, l Dch + d] = hankel_y[k lag + l + d ldx] (several channels, lag > 1: src/robustPCA.jl:81-90)
    HankelGeom hankel_geom;
    // ... and the caller has not built that panel at all (rpca_core is called with D == nullptr): the set-up works on a
    // transient copy in a buffer the loop only needs later, and the rare kernels without an implicit form build one on
    // demand (seven resident panels instead of eight)
    bool hankel_lazy = false;
    // The caller wants neither the panel A nor E, only what unhankel makes of A (lowrankfilter, one channel, lag 1): when
    // the loop ends with A still in factors (rank <= 32, no nonnegA) it is not materialised - Handle::out_factors says so
    // and out_Tm / out_Vs / out_r describe it - and the returned E is not formed.  rpca_core may then be called with
    // A == nullptr: the panel is only allocated (WS_A) if some iteration needs it in memory.
    bool factors_out = false;
    // called once A and E are final (the loop is over, the stream synchronised), before the returned decomposition is computed:
    // a host-pointer call starts their way back to the caller's memory here, beside the SVD of the last Z (solver.hip, rpca_entry)
    const std::function<void()>* ae_final = nullptr;
    // Vt on the device: when set, rpca_core writes the returned Vt (d x N, leading dimension vt_ld, element type of the call)
    // there with one kernel - no 2 MB round trip through host loops - and says so in *vt_written (the paths that only have
    // part of the decomposition keep the host form: Vt_host must be passed as well)
    void* vt_dev = nullptr;
    int64_t vt_ld = 0;
    bool* vt_written = nullptr;
};

inline ResolvedOpts resolve(const tlsq_rpca_opts* o, int64_t M, int64_t N, double default_tol) {
    ResolvedOpts r;
    r.m_global = (o && o->m_global > 0) ? o->m_global : M;
    const int64_t mx = std::max(r.m_global, N);
    r.lambda = (o && !std::isnan(o->lambda)) ? o->lambda : 1.0 / std::sqrt((double)mx);  // :157
    r.maxrank = (o && o->maxrank > 0) ? o->maxrank : std::numeric_limits<int64_t>::max();  // :158
    r.iters = (o && o->iters > 0) ? o->iters : 1000;                                     // :159
    r.tol = (o && !std::isnan(o->tol)) ? o->tol : default_tol;                           // :160
    r.rho = (o && !std::isnan(o->rho)) ? o->rho : 1.5;                                   // :161
    r.nonnegA = o && o->nonnegA;
    r.nonnegE = o && o->nonnegE;
    r.hankel = o && o->hankel;
    r.nukeA = o ? (o->nukeA != 0) : true;
    return r;
}


// small-matrix helpers of solver.hip that the complex path (solver_complex.hip) shares:
// sqrt(lambda_max) of a Gram matrix already on the device (Lanczos from a power start; dense fall-back); the full
// eigen-decomposition of G (V in workspace slot vslot); X = V[:, sel]
int sigma_max_of_gram(Handle* h, const double* G, int64_t N, double rel_tol, double* out, int64_t* sweeps,
                      double stop_above_sigma = 0.0);
int eig_full(Handle* h, const double* G, int64_t N, double** V_out, SmallSvd& s, int64_t* sweeps, bool allow_warm = false,
             int vslot = WS_V, bool need_all_vectors = false,
             // hints for the spectrum slicer (sliced.hip): an upper bound of the eigenvalues, a known cluster away from the rest
             double lam_hi = 0.0, int n_out = 0, double val_out = 0.0, double bulk_hi = 0.0);
int gather_cols(Handle* h, const double* V, int64_t N, const std::vector<int32_t>& sel, double* X);
// G (workspace slot `slot`) = Z'Z summed over the row shards
template <typename T>
inline int gram_allreduce(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, double** G_out, int slot = WS_G) {
    void* G;
    TLSQ_TRY(ws_get(h, slot, (size_t)N * N * 8, &G));
    TLSQ_TRY(gram_any(h, Z, Prec<T>::f32, M, N, ld, (double*)G, N));
    TLSQ_TRY(comm_allreduce(h, (double*)G, (size_t)N * N, ncclSum));
    *G_out = (double*)G;
    return TLSQ_OK;
}
// sigma_max of Z (device M x N, ld) through the Gram matrix in workspace slot gslot (the default `opnorm`)
template <typename T>
int opnorm_gram(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, double* out, int64_t* sweeps,
                double rel_tol = 1e-13, double stop_above_sigma = 0.0, int gslot = WS_G);
// V (WS_V, all N vectors) / s of the Gram of Z: what tls! and the SSA truncation of lowrankfilter need
template <typename T>
int svd_via_gram(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, double** V_out, SmallSvd& s,
                 int64_t* sweeps, PhaseTimer* pt);
// the same through the TSQR route (tsqr.hip + one-sided Jacobi on R'): full accuracy for every singular value; M >= N
template <typename T>
int svd_via_r(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, double** V_out, SmallSvd& s, int64_t* sweeps);
// Aout (M x N, ldA) = Z * V[:,sel] * diag(g) * V[:,sel]'
template <typename T>
int rebuild_lowrank(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ldZ, const double* V,
                    const std::vector<int32_t>& sel, const std::vector<double>& g, T* Aout, int64_t ldA);
// the ALM loop on device-resident contiguous panels (see solver.hip)
template <typename T>
int rpca_core(Handle* h, const T* D, int64_t M, int64_t N, const ResolvedOpts& ro, const tlsq_rpca_opts* opts, T* A,
              T* E, T* U_dev, double* S_host, double* Vt_host, int64_t ldVt, int64_t* sv_out, tlsq_rpca_info* info);
int rpca_core_complex(Handle* h, const double* D, int64_t M, int64_t N, const ResolvedOpts& ro,
                      const tlsq_rpca_opts* opts, double* A, double* E, double* S_host, int64_t* sv_out,
                      tlsq_rpca_info* info,
                      double* U_dev = nullptr, double* Vt_host = nullptr, int64_t ldVt = 0);
// staging of caller memory (host or device, any leading dimension) around rpca_core; wide problems are transposed
template <typename T>
int rpca_entry(tlsq_handle h, const T* D, int64_t M, int64_t N, int64_t ldD, const tlsq_rpca_opts* opts, T* A,
               int64_t ldA, T* E, int64_t ldE, T* U, int64_t ldU, T* S, T* Vt, int64_t ldVt, int64_t* sv,
               tlsq_rpca_info* info);
// x (n x q, ldx) from the right singular vectors: X V22 = -V21  (src/TotalLeastSquares.jl:65-69)
int tls_partition_solve(const double* Vt, int64_t ncols, int64_t ldVt, int64_t n, double* x, int64_t ldx);

}  // namespace tlsq
