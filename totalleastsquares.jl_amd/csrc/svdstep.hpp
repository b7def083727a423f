// The SVD step's small-matrix solver (svdstep.hip), shared with the ALM loop (solver.hip): the operator the solvers work on,
// the state of the warm-started subspace iteration with its count certificate, and their entry points.  Split from
// solver.hip in round 5.
#pragma once
#include "internal.hpp"

namespace tlsq {

// The N x N operator the small solvers work on: either the explicit Gram matrix G = Z'Z (summed over the row
// shards), or - large mode, where forming G would cost far more than the few products the subspace solver needs
// (2 M N^2 flops against 4 M N p per product) - the panel itself: G X = Z'(Z X), two streaming passes over Z.
struct GramOp {
    const double* G = nullptr;   // explicit (N x N, ld N)
    const void* Z = nullptr;     // implicit: M x N panel (ld ldZ), fp32 when z_f32
    int z_f32 = 0;
    int64_t M = 0, ldZ = 0;
    // products of an fp32 panel may round the block to fp32 on the way (op_gram_f32: 6e-8 per entry): fine for the range finder
    // of the randomized hook, which accepts whatever comes out; NOT for the certified solver, whose acceptance test wants
    // residuals of 2e-13 (16384 x 8192 fp32 in the default mode ended with "could not be served" while this was unconditional)
    bool lowp_ok = false;
    bool implicit() const { return G == nullptr; }
};

// ---- warm-started subspace iteration (subspace.hip) ------------------------------------------------
struct SubspaceState {
    bool valid = false;
    bool allow_cold = true;
    int64_t p = 0;       // columns of X (WS_SX, N x p)
    int64_t ntop = 0;    // the first ntop columns of X were >= 1/mu in the iteration that produced them
    // hook mode (`svd = rsvd`-style user hook, src/robustPCA.jl:195-197): rank-`hook_rank` randomized SVD from a
    // fresh random block, fixed number of passes, no convergence test and no count certificate
    int64_t hook_rank = 0;
    uint64_t hook_seed = 0;
    // (in) leading columns of the block buffer (WS_SX) that still hold the sorted Ritz vectors of the previous iteration's
    // decomposition: the hook starts from them instead of a fresh random block; (out) the columns this call left there
    int64_t hook_carry = 0;
    // (out, sketch on an fp32 panel) the product Z Q of the hook's Rayleigh-Ritz step (fp32, rows x hook_zq_lw, ld = rows, in
    // WS_OPT until the next operator product), the eigenvector matrix S of Q'GQ on the device (p x p, WS_SS) and the order in
    // which its columns were sorted: the caller forms the factor Z X[:, sel] = (Z Q) S[:, order[sel]] from them instead of a pass
    // over the panel.  hook_zq == nullptr: not available.
    const float* hook_zq = nullptr;
    int hook_zq_lw = 0;
    const double* hook_S = nullptr;
    std::vector<int32_t> hook_order;
    int64_t fast = 0, full = 0, steps = 0;
    // why the last call gave up (0 = it did not): the caller may enlarge the block and try again
    enum { FAIL_NONE = 0, FAIL_SMALL = 1, FAIL_NOCONV = 2, FAIL_CERT = 3, FAIL_NUMERIC = 4, FAIL_WINDOW = 5 };
    int fail = FAIL_NONE;
    bool skip_certificate = false;   // the caller certifies the count itself (late iterations, see svd_precise_fast)
    // deferred certificate: svd_subspace returns with *ok = true as soon as the Lanczos steps of the certificate are
    // queued; the caller queues its own work (the rebuild) behind them and then asks svd_subspace_certify
    bool defer_certificate = false;
    bool cert_pending = false;
    // ... and its kernels run on the handle's second stream with a mailbox region of their own (cert_async): the caller may
    // queue anything that does not modify G or the block X on the main stream meanwhile - rpca_core queues the factor
    // product, the next sweep and the next Gram before it asks for the verdict
    bool cert_async = false;
    volatile double* cert_mb = nullptr;   // where cert_finish polls (nullptr: the main mailbox)
    // asynchronous form: the two kernels are not queued by svd_subspace itself but by whoever calls cert_launch - rpca_core does
    // once the HBM-bound sweep is through (beside the sweep they cost it ~8 % of its bandwidth; beside the MFMA-bound Gram of
    // the next iteration they are not noticed)
    std::function<int()> cert_launch;
    // Speculative factor product of the rebuild (rpca_core): queued right behind k_ritz_finish of a warm block's first step,
    // with the selection, the weights and the count taken from the device-side decision block that kernel writes - the host
    // round trip (poll, sort, count, launch: ~10 us) is then hidden behind the product instead of standing in front of it.
    // The host checks afterwards that the device decided what it decides itself (dev_ok, dev_r); anything else - a second
    // step, a re-ordered block, a rank that needs more accumulator tiles - simply launches the product again.
    struct SpecRebuild {
        bool enable = false;             // the caller wants it for this call (buffers below are valid)
        const void* Z = nullptr;
        int z_f32 = 0;
        int64_t M = 0, ldz = 0;
        double *Tout = nullptr, *Vs = nullptr;   // room for 32 columns each
        bool nukeA = true;
        bool launched = false;           // result: a product was queued in the step that converged ...
        int nct = 0;                     // ... with this many 16-column accumulator tiles
        bool dev_ok = false;             // ... and this is what the device decided
        int64_t dev_r = 0;
        std::function<void()> before_launch;   // phase accounting of the caller: the eig window ends where the product starts
    } spec;
    LanczosRun cert;
    int q_warm = 3;        // multiplications by G applied to the top columns of a warm block per step
    int q_floor = 1;       // smallest count that may be tried again (raised when a count needed a second step)
    double chol_piv = 0.0; // smallest CholeskyQR pivot of the previous step on this warm block (one-pass guess, launch_orth)
    int64_t chol_p = 0;    //   ... and the block width it was measured at
    int64_t cold_p = 18;   // block size of a cold start
    int extra_steps = 0;   // added to the step budget (retries in large mode)
    // Uncertainty of an eigenvalue of the computed Gram matrix relative to lambda_max (rounding of G = Z'Z, Ritz
    // residuals).  An eigenvalue within dlam = noise_rel * lambda_max of the threshold (1/mu)^2 cannot be counted
    // reliably on this route (FAIL_WINDOW: the caller decides on the TSQR route), and the tail certificate has to
    // clear the threshold by the same margin.  0: no window (large mode, where no other solver exists).
    double noise_rel = 0.0;
    double dlam = 0.0;     // noise_rel * lambda_max of the last call
    // count certificate in flight (scaled by 1 / tau^2): deflated matrix, pass mark, which bound was queued
    const double* cert_GD = nullptr;
    int64_t cert_N = 0;
    double cert_margin = 0.0, cert_seq = 0.0;
    int cert_ntile = 0;
    bool cert_power = false;
    int64_t n_power = 0, n_power_l2 = 0, n_lanczos_cert = 0;   // statistics: served by S^2 / S^4 / Lanczos
    int64_t n_rr_fast = 0, n_rr_declined = 0;                  // steps served / declined by the fused Rayleigh-Ritz kernel
    int rr_streak = 0, rr_skip = 0;                            // consecutive declines / calls in which it is not tried
    double cert_tail = 0.0;   // Lanczos estimate of lambda_max(GD) / tau^2 of the last failed certificate (0: unknown)
};

// the asynchronous certificate's region of the mailbox (in doubles; the main region - flag at [0], payloads from [8] - ends
// below it for every block size in use)
constexpr size_t kCertMailboxOffset = 2048;

// runs a scope's launches on another stream of the handle (every launcher takes h->stream)
struct StreamScope {
    Handle* h;
    hipStream_t saved;
    StreamScope(Handle* hh, hipStream_t s) : h(hh), saved(hh->stream) {
        if (s) h->stream = s;
    }
    ~StreamScope() { h->stream = saved; }
};


// sigma_max of the panel behind an implicit operator (Lanczos on Z'Z through products)
int sigma_max_of_op(Handle* h, const GramOp& op, int64_t N, double rel_tol, double* out, double stop_above_sigma = 0.0);
// the count certificate of a subspace step: queued (asynchronous form) / its verdict
int power_cert_begin(Handle* h, SubspaceState& st);
int cert_finish(Handle* h, SubspaceState& st, bool* pass);
// the sigma_i >= inv_mu pairs of the operator from the block carried in st; *ok = false: the caller runs a dense tier
int svd_subspace(Handle* h, const GramOp& op, int64_t N, double inv_mu, SubspaceState& st, double** V_out, SmallSvd& s,
                 int64_t* sweeps, bool* ok);
int svd_subspace_certify(Handle* h, SubspaceState& st, double inv_mu, bool* ok);
// Xs = V[:, sel] diag(w) (and the plain gather) through an uploaded selection list
int gather_scale_host(Handle* h, const double* V, int64_t N, const std::vector<int32_t>& sel,
                             const std::vector<double>& w, void* aux, double* Vg, double* Vs);
// WS_SX = the dominant block of this iteration's decomposition, for the next iteration's warm start
int carry_block(Handle* h, const double* V, int64_t N, const SmallSvd& s, int64_t svp, int64_t pmax, SubspaceState& sub);

}  // namespace tlsq
