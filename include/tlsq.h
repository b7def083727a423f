/* tlsq.h — C ABI of libtlsqhip.so: MI355X (gfx950) robust-PCA / low-rank-recovery engine.
 *
 * Drop-in boundary for the `rpca`, `lowrankfilter`, `hankel`, `unhankel`, `tls!` and `rtls`
 * entry points of baggepinnen/TotalLeastSquares.jl (reference paths below are relative to
 * /root/reference).  The reference has no FFI of its own; these are the entry points a thin Julia
 * `ccall` shim (julia/TotalLeastSquaresHIP.jl) binds in place of the Julia bodies.
 *
 * Conventions
 *  - All matrices are COLUMN-MAJOR (Julia native) with an explicit leading dimension, 64-bit sizes.
 *  - The caller owns every input and output buffer.  `memory` in the options says whether the
 *    pointers are host pointers (TLSQ_MEM_HOST: Julia `Array`s through `Ptr{T}`) or device pointers
 *    already resident in this GPU's HBM (TLSQ_MEM_DEVICE).  The library never retains a caller
 *    pointer after a call returns and never frees caller memory.
 *  - Every entry point returns an int status: 0 OK, >0 non-fatal (TLSQ_MAXITER = the reference's
 *    `@warn "Maximum number of iterations reached"`, src/robustPCA.jl:232), <0 fatal.  No C++
 *    exception or exit() crosses the ABI.  tlsq_last_error(h) gives the text of the last failure.
 *  - A handle is used by one host thread at a time.  All calls block until the device work has completed (except the
 *    tlsq_k_* kernel entry points at the end of this file, which are asynchronous on the handle's stream).
 *    Multi-GPU, row-sharded, two ways: tlsq_create_multi (one process, one handle, host matrices - the drop-in for
 *    the Julia package), or one handle per GPU / process joined by tlsq_comm_init for device-resident shards.  The
 *    only exchanged data are N x N matrices (Gram all-reduce, TSQR factor all-gather) over RCCL / xGMI.
 *  - There is NO CPU fallback: without a GPU (or for complex element types) calls fail with
 *    TLSQ_ERR_HIP / TLSQ_ERR_UNSUPPORTED.
 */
#ifndef TLSQ_H
#define TLSQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tlsq_handle_s* tlsq_handle;

enum {
    TLSQ_OK = 0,
    TLSQ_MAXITER = 1,           /* src/robustPCA.jl:232 — a warning in the reference, not an error */
    TLSQ_ERR_ARG = -1,          /* bad size / NULL / the reference's @assert (src/robustPCA.jl:79-80) */
    TLSQ_ERR_HIP = -2,          /* HIP runtime error, no device */
    TLSQ_ERR_OOM = -3,
    TLSQ_ERR_COMM = -4,         /* RCCL error */
    TLSQ_ERR_UNSUPPORTED = -5,  /* e.g. N too large for the LDS-resident Jacobi, complex T */
    TLSQ_ERR_NOCONV = -6,       /* small eigensolver did not converge */
    TLSQ_ERR_NONFINITE = -7     /* the input contains Infs or NaNs: what LAPACK's chkfinite throws behind svd! / opnorm in the
                                   reference (ArgumentError("matrix contains Infs or NaNs"), src/robustPCA.jl:177, :194) */
};

enum { TLSQ_MEM_HOST = 0, TLSQ_MEM_DEVICE = 1 };
enum { TLSQ_SVD_FULL = 0, TLSQ_SVD_RANDOMIZED = 1, TLSQ_SVD_CALLBACK = 2 };       /* the `svd` hook, src/robustPCA.jl:168,193-197 */
enum { TLSQ_OPNORM_EXACT = 0, TLSQ_OPNORM_POWER = 1, TLSQ_OPNORM_CALLBACK = 2 };  /* the `opnorm` hook, :169,177,225 */

/* Arbitrary user hooks (TLSQ_SVD_CALLBACK / TLSQ_OPNORM_CALLBACK): the reference accepts any `svd(Z, sv)` returning
 * .U, .S, .Vt and any `opnorm(X)::real(T)` (src/robustPCA.jl:168-169; test/runtests.jl:384-398).  A closure of the host
 * language cannot run on the GPU, so in these modes the working panel is copied to HOST memory and the callback is
 * invoked there, on the calling thread, once per use (iterations k >= 2 for svd, exactly like :193-197; set-up and every
 * iteration for opnorm).  Element type = that of the entry point (double for _f64, float for _f32), column-major.
 *   svd_cb:    Z (M x N, ldZ) in; fill U (M x k, ldU), S (k, descending), Vt (k x N, ldVt) and *k_out = k <= min(M,N)
 *              (buffers hold min(M,N) triplets); return 0, anything else aborts the call with TLSQ_ERR_ARG.
 *   opnorm_cb: returns the norm estimate of X (M x N, ldX).
 * For a wide input (M < N, unsharded) the library works on the transposed problem and the hooks receive that panel
 * (N x M): singular values and norms are the same, U and V swap roles consistently.
 * Row shards (a tlsq_create_multi group, or one handle per GPU joined by tlsq_comm_init): every use of a hook gathers the
 * shards, RANK 0 calls its hook on the whole M_global x N panel - on the calling thread of a group handle; in rank 0's
 * process otherwise, the other ranks' pointers only have to be non-NULL - and the results are sent back to the ranks
 * (the panel crosses PCIe twice per use: correct, and slow like every host hook). */
typedef int (*tlsq_svd_cb)(const void* Z, int64_t M, int64_t N, int64_t ldZ, int64_t sv, void* U, int64_t ldU, void* S,
                           void* Vt, int64_t ldVt, int64_t* k_out, void* user);
typedef double (*tlsq_opnorm_cb)(const void* X, int64_t M, int64_t N, int64_t ldX, void* user);

/* Keyword arguments of rpca (src/robustPCA.jl:156-170).  Fill with tlsq_rpca_opts_default() first;
 * NaN / <=0 sentinels resolve to the reference's defaults at call time (they depend on T, M, N). */
typedef struct tlsq_rpca_opts {
    double  lambda;      /* NaN -> 1/sqrt(max(M_global,N))   (:157) */
    int64_t maxrank;     /* <=0 -> typemax(Int)              (:158) */
    int64_t iters;       /* <=0 -> 1000                      (:159) */
    double  tol;         /* NaN -> sqrt(eps(real(T)))        (:160) */
    double  rho;         /* NaN -> 1.5                       (:161) */
    int32_t nonnegA;     /* (:163) */
    int32_t nonnegE;     /* (:164) */
    int32_t hankel;      /* (:165) */
    int32_t nukeA;       /* (:167) default 1 */
    int32_t svd_mode;    /* TLSQ_SVD_*    */
    int32_t opnorm_mode; /* TLSQ_OPNORM_* */
    int32_t opnorm_mvps; /* power-iteration mat-vec pairs for TLSQ_OPNORM_POWER (rnorm(x, mvps)) */
    int32_t memory;      /* TLSQ_MEM_* for every matrix/vector pointer of the call */
    int64_t m_global;    /* total rows over all ranks when row-sharded; 0 -> M (single GPU) */
    uint64_t seed;       /* randomized modes */
    /* live `verbose` hook (src/robustPCA.jl:226): called on the calling thread after each iteration */
    void (*on_iter)(int64_t k, double cost, int64_t svp, void* user);
    void* user;
    tlsq_svd_cb svd_cb;         /* svd_mode == TLSQ_SVD_CALLBACK */
    tlsq_opnorm_cb opnorm_cb;   /* opnorm_mode == TLSQ_OPNORM_CALLBACK */
    /* 1: bracket every phase of an iteration with HIP events (ms_gram, ms_eig, ms_rebuild, ms_opnorm of the report).  0:
     * only the sweep kernels are bracketed (ms_shrink, ms_update) - each recorded event is a packet the command processor
     * works through between two kernels, ~6 us, and an iteration has nine phase boundaries. */
    int32_t phase_timing;
    int32_t reserved0;
} tlsq_rpca_opts;

/* Per-call report.  cost_hist / svp_hist are optional caller-provided arrays of hist_capacity entries.
 * When neither cost_hist nor opts->on_iter is given, the per-iteration opnorm is only resolved far enough to
 * settle `cost < tol` (src/robustPCA.jl:228); final_cost is exact whenever the loop ends. */
typedef struct tlsq_rpca_info {
    int64_t iters_done;
    int32_t converged;
    int32_t tsqr_iterations; /* iterations whose SVD step went through the TSQR route (TSQR + one-sided Jacobi) */
    double  final_cost;
    double  final_mu;
    double  d_norm;          /* opnorm(D), src/robustPCA.jl:177 */
    double* cost_hist;
    int64_t* svp_hist;
    int64_t hist_capacity;
    int64_t jacobi_sweeps;   /* total sweeps of the small eigensolver */
    /* wall/device time in ms.  ms_shrink / ms_update: the sweep kernels between HIP events on the handle's stream - every
     * sweep with opts->phase_timing, otherwise a SAMPLE of them (the first shrink and the sweep of every fourth iteration:
     * an event record is a packet of its own between two kernels, ~6 us each; sweeps_timed / hbm_bytes_sweeps_timed
     * below say how many launches and how many bytes the two figures cover).  The other phases only with
     * opts->phase_timing (zero otherwise) */
    double ms_total, ms_loop, ms_h2d, ms_d2h;
    double ms_shrink, ms_update, ms_gram, ms_eig, ms_rebuild, ms_opnorm;
    /* how the SVD step of each iteration was served: full Jacobi decompositions vs warm-started subspace
     * iterations (and the total number of subspace steps) */
    int64_t eig_full, eig_fast, subspace_steps;
    /* sweeps that did not store the residual panel (its cost evaluation was predicted to be skipped): these moved
     * one panel pass less */
    int64_t residual_stores_skipped;
    /* HBM bytes the panel-sized kernels of the ALM loop had to move (algorithmic bytes of what was launched: passes over
     * M x N x sizeof(T) panels): the sweep kernels alone, and everything (sweeps + Gram reads + the
     * rebuild's read of Z and write of A + residual Gram).  Per GPU when row-sharded. */
    double hbm_bytes_sweeps, hbm_bytes;
    /* the sweep launches bracketed by HIP events (ms_shrink + ms_update are their device time) and the algorithmic bytes
     * of exactly those launches */
    int64_t sweeps_timed;
    double hbm_bytes_sweeps_timed;
    /* launches of the kernels that only some shapes take, counted over this call (per GPU): the fp16-split Gram matrix
     * (gram16.hip, k_gram_h3), the fp16-split operator products (opgram16.hip: k_zx_h alone = the rebuild's factor product,
     * k_zx_h + k_zty_h = one Z'(Z X)), the wide-rank sweep (sweeps.hip, k_zsweep_wide) and the fused sweep + Gram kernel
     * (fused.hip, k_fused_zgram).  The parity tests assert on them that a fixture exercised the path it was made for. */
    int64_t kern_gram_h3, kern_zx_h, kern_zty_h, kern_zsweep_wide, kern_fused_zgram;
} tlsq_rpca_info;

const char* tlsq_version(void);

/* Development switches (tests and tools only; process-wide; not part of the reference's interface).  Every alternative HIP
   path of the library - "classic sweeps instead of the E-free loop", "no mailbox read-backs", "poison the workspace" ... -
   is selected by name through this call (the list: csrc/common.hpp, TLSQ_DEV_LIST).  `name` with or
   without the TLSQ_ prefix, `value` a short string ("1", "0", "8.0" ...), NULL clears the switch.  Returns TLSQ_ERR_ARG for
   a name that is not on the list.  The shipped library reads NO environment variable; a build with -DTLSQ_DEV_SWITCHES
   additionally takes TLSQ_<NAME> from the environment at the first tlsq_create.  Not thread-safe: set between calls. */
int tlsq_dev_set(const char* name, const char* value);
void        tlsq_rpca_opts_default(tlsq_rpca_opts* o);

int  tlsq_create(int device_id, tlsq_handle* out);
/* Single-process multi-GPU handle: one host process (e.g. a Julia session) drives `ngpus` GPUs of
 * the node.  device_ids == NULL means 0 .. ngpus-1.  The library creates one stream + workspace per GPU and one RCCL
 * communicator over them (ncclCommInitAll), and serves tlsq_rpca_f64/_f32 and tlsq_lowrankfilter_f64 on HOST matrices
 * by scattering contiguous row blocks (strided 2-D copies straight from the caller's column-major arrays), running
 * one worker thread per GPU, and gathering A, E (U) back; S, Vt, sv, the report and the on_iter hook come from rank
 * 0 on the calling thread - worker threads never call back into the host language.  Every other entry point (and
 * problems that do not shard: wide matrices, tiny row counts) runs on the first GPU alone.  ngpus = 1 is valid
 * (same code path, one-rank communicator).  A device id may be repeated (testing the rank > 1 paths on a box with
 * fewer GPUs): such a group gets a host-staged loop-back communicator instead of RCCL - correct, not fast. */
int  tlsq_create_multi(int ngpus, const int* device_ids, tlsq_handle* out);
int  tlsq_ngpus(tlsq_handle h);   /* GPUs behind the handle (1 for tlsq_create) */
int  tlsq_destroy(tlsq_handle h);
const char* tlsq_last_error(tlsq_handle h);
/* hipStream_t the handle launches on (as void*), so a caller can time it with HIP events */
void* tlsq_stream(tlsq_handle h);
/* block until everything queued on the handle's stream has completed (the tlsq_k_* entry points are
 * asynchronous on that stream; all other entry points synchronise before returning) */
int   tlsq_synchronize(tlsq_handle h);

/* ---- multi-GPU (row sharding, one handle per GPU) -------------------------------------------- */
#define TLSQ_UNIQUE_ID_BYTES 128
int tlsq_comm_unique_id(unsigned char id[TLSQ_UNIQUE_ID_BYTES]);              /* rank 0, then broadcast */
int tlsq_comm_init(tlsq_handle h, int nranks, int rank, const unsigned char id[TLSQ_UNIQUE_ID_BYTES]);
int tlsq_comm_destroy(tlsq_handle h);
/* number of ranks of the communicator attached to the handle (tlsq_comm_init, or the group of tlsq_create_multi) as RCCL
 * itself reports it (ncclCommCount); 1 without a communicator.  bench.py prints it so that a multi-GPU run shows that
 * the exchange really went through an N-rank communicator. */
int tlsq_comm_size(tlsq_handle h, int* nranks);

/* ---- rpca: src/robustPCA.jl:156-239 ----------------------------------------------------------
 * D  M x N (ldD)  in;  A, E  M x N out;  optional (may be NULL): U M x d (ldU), S d, Vt d x N (ldVt),
 * d = min(M_global,N) — the SVD of the last Z = D-E+Y/mu (the reference's returned `s`, :194,:238).
 * *sv = estimated rank (:204,:238).  When row-sharded, M/D/A/E/U are the local shard.
 * Large mode, min(M,N) in (2048, 65536]: every SVD step of the loop is served by the certified subspace solver (there
 * is no dense eigensolver of that size in the loop); A, E, sv, the iteration count and the costs are the same as
 * always.  The returned U/S/Vt are complete up to min(M,N) = 4608 (TSQR + one-sided Jacobi once after the loop, a
 * few seconds, only when asked for); beyond that only the leading triplets (the sigma >= 1/mu ones plus the solver's
 * padding) are returned - the rest of S is NaN and the corresponding vectors are zero.  The subspace block grows to
 * 512 columns; an iteration it cannot serve (a rank beyond ~480, or a cold start on a high-rank problem) goes through
 * the TSQR route up to min(M,N) = 4608 (about a second per decomposition) and is TLSQ_ERR_UNSUPPORTED beyond.
 * From min(M,N) = 8192 on no N x N matrix is formed at all (operator products G X = Z'(Z X)); min(M,N) > 65536:
 * TLSQ_ERR_UNSUPPORTED (untested beyond; 40000 x 20000 fp32 and 20000 x 17000 fp64 are part of the GPU suite's sizes).
 * Device memory held by the handle (grow-only workspace, freed by tlsq_destroy): three M x N panels of the element type
 * (Y, Z, residual; a fourth - the second Z of the speculative loop - for panels up to 2 GB) beside D, A, E - which are the caller's with TLSQ_MEM_DEVICE and three more workspace panels with
 * TLSQ_MEM_HOST; the hankel flag and the svd / opnorm hook modes add two more (a second E and Z).  The A and E panels
 * are scratch while the call runs (E holds the second copy of Y); they are written in full before the call returns. */
int tlsq_rpca_f64(tlsq_handle h, const double* D, int64_t M, int64_t N, int64_t ldD,
                  const tlsq_rpca_opts* opts, double* A, int64_t ldA, double* E, int64_t ldE,
                  double* U, int64_t ldU, double* S, double* Vt, int64_t ldVt,
                  int64_t* sv, tlsq_rpca_info* info);
int tlsq_rpca_f32(tlsq_handle h, const float* D, int64_t M, int64_t N, int64_t ldD,
                  const tlsq_rpca_opts* opts, float* A, int64_t ldA, float* E, int64_t ldE,
                  float* U, int64_t ldU, float* S, float* Vt, int64_t ldVt,
                  int64_t* sv, tlsq_rpca_info* info);

/* ---- Hankel lag embedding / anti-diagonal averaging ------------------------------------------
 * hankel:   src/robustPCA.jl:76-92   x (Nx x Dch, ldx) -> X (K x L*Dch, ldX), K=(Nx-L)/lag+1
 * unhankel: src/robustPCA.jl:28-39,53-68   A (K x L*Dch) -> y (Nx x Dch, ldy)
 * soft_hankel: src/robustPCA.jl:9-21  in place on A (K x L)
 * `memory` = TLSQ_MEM_* for all pointers of the call. */
int tlsq_hankel_f64(tlsq_handle h, const double* x, int64_t Nx, int64_t Dch, int64_t ldx,
                    int64_t L, int64_t lag, double* X, int64_t ldX, int memory);
int tlsq_unhankel_f64(tlsq_handle h, const double* A, int64_t K, int64_t LD, int64_t ldA,
                      int64_t lag, int64_t Nx, int64_t Dch, double* y, int64_t ldy, int memory);
int tlsq_soft_hankel_f64(tlsq_handle h, double* A, int64_t K, int64_t L, int64_t ldA,
                         double eps, int memory);
int tlsq_hankel_f32(tlsq_handle h, const float* x, int64_t Nx, int64_t Dch, int64_t ldx,
                    int64_t L, int64_t lag, float* X, int64_t ldX, int memory);
int tlsq_unhankel_f32(tlsq_handle h, const float* A, int64_t K, int64_t LD, int64_t ldA,
                      int64_t lag, int64_t Nx, int64_t Dch, float* y, int64_t ldy, int memory);
int tlsq_soft_hankel_f32(tlsq_handle h, float* A, int64_t K, int64_t L, int64_t ldA,
                         float eps, int memory);

/* ---- lowrankfilter: src/robustPCA.jl:119-128 --------------------------------------------------
 * y (Nx x Dch, ldy) -> yf (Nx x Dch, ldyf).  n = embedding size (<=0 -> min(Nx/20,2000)),
 * sv>0 -> plain rank-sv truncation (:123-126).  opts->tol NaN -> 1e-3 (the lowrankfilter default).
 * The Hankel matrix is built, factored and averaged on the device; it never visits the host.
 * With a communicator (tlsq_comm_init) every rank passes the WHOLE series and receives the whole filtered series: a
 * rank owns a contiguous block of the rows of H — a time window of y with an (n-1)-sample halo — rpca runs
 * row-sharded, and the anti-diagonal averaging exchanges partial sums and counts with one all-reduce.
 * Device memory: with one channel and lag 1 (robust mode, one GPU) neither the Hankel matrix nor the low-rank panel is
 * ever stored - the sweeps read y[i + j], the filtered series is averaged from the factors of the low-rank part - and
 * the handle holds four K x n panels (measured 83 GB for Nx = 1e7, n = 256); several channels, a lag above 1 and row
 * shards keep H and A as panels (six to eight panels). */
int tlsq_lowrankfilter_f64(tlsq_handle h, const double* y, int64_t Nx, int64_t Dch, int64_t ldy,
                           int64_t n, int64_t lag, int64_t sv, const tlsq_rpca_opts* opts,
                           double* yf, int64_t ldyf, tlsq_rpca_info* info);

int tlsq_lowrankfilter_f32(tlsq_handle h, const float* y, int64_t Nx, int64_t Dch, int64_t ldy,
                           int64_t n, int64_t lag, int64_t sv, const tlsq_rpca_opts* opts,
                           float* yf, int64_t ldyf, tlsq_rpca_info* info);   /* same forms as _f64 (group handles and communicators included) */

/* ---- tls! / rtls: src/TotalLeastSquares.jl:63-69, 152-156 -------------------------------------
 * tls:  Ay (M x ncols, ldAy; NOT destroyed, unlike svd!) , n = columns of A -> x (n x q), q=ncols-n
 * rtls: A (M x n), y (M x q) -> x (n x q) via rpca([A y]; nukeA=false) then tls!(s, n).
 * tls_from_vt: the `tls!(s::SVD, n)` method on a caller-held Vt (ncols x ncols, HOST memory). */
int tlsq_tls_f64(tlsq_handle h, const double* Ay, int64_t M, int64_t ncols, int64_t ldAy,
                 int64_t n, double* x, int64_t ldx, int memory);
int tlsq_rtls_f64(tlsq_handle h, const double* A, int64_t M, int64_t n, int64_t ldA,
                  const double* y, int64_t q, int64_t ldy, const tlsq_rpca_opts* opts,
                  double* x, int64_t ldx, tlsq_rpca_info* info);
/* Float32 methods (the reference functions are generic in the element type, src/TotalLeastSquares.jl:63,152): fp32
 * panels, fp64 small-matrix work, x returned in fp32 */
int tlsq_tls_f32(tlsq_handle h, const float* Ay, int64_t M, int64_t ncols, int64_t ldAy,
                 int64_t n, float* x, int64_t ldx, int memory);
int tlsq_rtls_f32(tlsq_handle h, const float* A, int64_t M, int64_t n, int64_t ldA,
                  const float* y, int64_t q, int64_t ldy, const tlsq_rpca_opts* opts,
                  float* x, int64_t ldx, tlsq_rpca_info* info);

/* ---- ComplexF64 rpca: the complex soft_th method (src/robustPCA.jl:3-7; test/runtests.jl:187-199) -------------
 * D, A, E are interleaved (re, im) complex M x N matrices, column-major, leading dimensions in complex elements;
 * S (optional) receives the min(M,N) singular values of the last Z.  Spectral steps run on the realified
 * 2M x 2N panel with the real path's kernels (full decompositions, N <= 1024).  nonnegA / nonnegE / hankel (no
 * complex method in the reference either), hook modes and row sharding: TLSQ_ERR_UNSUPPORTED. */
int tlsq_rpca_c64(tlsq_handle h, const double* D, int64_t M, int64_t N, int64_t ldD, const tlsq_rpca_opts* opts,
                  double* A, int64_t ldA, double* E, int64_t ldE, double* S, int64_t* sv, tlsq_rpca_info* info);
/* ... with the singular vectors of the returned `s` (src/robustPCA.jl:194,238): U (M x d complex, ldU), Vt = V' (d x N
 * complex, ldVt), d = min(M,N), interleaved like D; each may be NULL.  One more complete decomposition of the realified
 * last Z after the loop (TSQR route when M >= N); vectors are unique up to a phase per singular value (an orthonormal
 * basis per cluster of equal singular values); left vectors of zero singular values are returned as zero columns. */
int tlsq_rpca_c64_svd(tlsq_handle h, const double* D, int64_t M, int64_t N, int64_t ldD, const tlsq_rpca_opts* opts,
                      double* A, int64_t ldA, double* E, int64_t ldE, double* U, int64_t ldU, double* S, double* Vt,
                      int64_t ldVt, int64_t* sv, tlsq_rpca_info* info);

/* ComplexF32 data (`rpca(D::Matrix{ComplexF32})`: the generic method of src/robustPCA.jl:156 with the complex soft_th of
 * :3-7): interleaved (re, im) FLOAT matrices in and out, S in float.  The panels are widened on the device and the solve
 * runs in the ComplexF64 path above (fp64 arithmetic throughout; default tol = sqrt(eps(Float32)) like the reference's
 * `tol = sqrt(eps(real(T)))`, :160); results are rounded to float on the way out.  Same restrictions as tlsq_rpca_c64. */
int tlsq_rpca_c32_svd(tlsq_handle h, const float* D, int64_t M, int64_t N, int64_t ldD, const tlsq_rpca_opts* opts,
                      float* A, int64_t ldA, float* E, int64_t ldE, float* U, int64_t ldU, float* S, float* Vt,
                      int64_t ldVt, int64_t* sv, tlsq_rpca_info* info);

/* ---- batched tiny problems ---------------------------------------------------------
 * The reference's typical use is a loop over thousands of independent small problems
 * (test/runtests.jl:205-235: `rtls(A, y)` on 50x4 ... 500x6 matrices).  These entry points run `batch` such
 * problems in one launch, one workgroup per problem with all panels in LDS; each problem follows
 * src/robustPCA.jl:156-239 exactly like tlsq_rpca_f64 (both SVDs of an iteration are one-sided Jacobi SVDs of the
 * panel itself, so there is no Gram-route accuracy limit).  Problems are stored back to back: D is M x N x batch
 * (column-major M x N blocks, ld = M), likewise A, E; optional per-problem outputs S (N), Vt (N x N, ld N), sv,
 * iters, status (0 converged / 1 iteration limit), cost (final).  N <= 32, M >= N; the hankel flag, hook modes
 * and on_iter are not available (TLSQ_ERR_UNSUPPORTED).  Returns TLSQ_MAXITER when any problem did not converge: per-problem
 * status 0 = converged, 1 = iteration limit, 2 = the problem contains Infs or NaNs (NaN results, no iterations; where a loop of
 * reference calls would stop with LAPACK's ArgumentError - the other problems of the batch are solved). */
int tlsq_rpca_batched_f64(tlsq_handle h, const double* D, int64_t M, int64_t N, int64_t batch,
                          const tlsq_rpca_opts* opts, double* A, double* E, double* S, double* Vt, int64_t* sv,
                          int32_t* iters, int32_t* status, double* cost);
/* Float32 elements (the reference's functions are generic in the element type): the same kernel with fp32 panels and
 * fp32 Jacobi SVDs (twice the problems per CU in LDS); default tol = sqrt(eps(Float32)). */
int tlsq_rpca_batched_f32(tlsq_handle h, const float* D, int64_t M, int64_t N, int64_t batch,
                          const tlsq_rpca_opts* opts, float* A, float* E, float* S, float* Vt, int64_t* sv,
                          int32_t* iters, int32_t* status, float* cost);
/* x_b = rtls(A_b, y_b) for b = 1..batch (src/TotalLeastSquares.jl:152-156): A is M x n x batch, y is M x q x batch,
 * x is n x q x batch. */
int tlsq_rtls_batched_f64(tlsq_handle h, const double* A, const double* y, int64_t M, int64_t n, int64_t q,
                          int64_t batch, const tlsq_rpca_opts* opts, double* x, int32_t* iters, int32_t* status);
int tlsq_rtls_batched_f32(tlsq_handle h, const float* A, const float* y, int64_t M, int64_t n, int64_t q,
                          int64_t batch, const tlsq_rpca_opts* opts, float* x, int32_t* iters, int32_t* status);
int tlsq_tls_from_vt_f64(const double* Vt, int64_t ncols, int64_t ldVt, int64_t n,
                         double* x, int64_t ldx);

/* ---- rpca_ga: Grassmann averages, src/robustPCA.jl:255-310 ---------------------------
 * X is d x N (ldX), a COLUMN is one observation; Q (d x r, ldQ) receives the r components, found one at a time
 * with deflation (:262-279).  The spherical average `μ` (:297) is chosen by opts->average:
 *   TLSQ_GA_MEAN          μ!                      (:312-320)  the default
 *   TLSQ_GA_TRIMMED_MEAN  entrywise_trimmed_mean  (:327-337)  with P = opts->trim
 *   TLSQ_GA_MEDIAN        entrywise_median        (:354-362)
 * q0 (d x r, ldq0; may be NULL) holds the start vector of each component — the reference draws `randn(d)` from
 * Julia's global RNG (:289), which no other program can reproduce; with q0 == NULL the library draws seeded
 * normals (opts->seed).  Returns TLSQ_MAXITER when a component used all `iters` iterations without dq < tol (the
 * `@warn "Reached maximum number of iterations"` of :306).  The entrywise averages need d*N < 2^31.
 * With a communicator (tlsq_comm_init) X holds this rank's COLUMNS (contiguous blocks in rank order: the entrywise averages
 * break ties by the column index within the whole row); a group handle (tlsq_create_multi) splits the columns of a host
 * matrix itself (at least 64 per GPU, otherwise the first GPU alone).  All three averages are available on shards. */
enum { TLSQ_GA_MEAN = 0, TLSQ_GA_TRIMMED_MEAN = 1, TLSQ_GA_MEDIAN = 2, TLSQ_GA_CALLBACK = 3 };
/* TLSQ_GA_CALLBACK: ANY spherical average of the host language - the reference's `μ = f` keyword (src/robustPCA.jl:286, :297:
 * `μᵢ = μ(q, w, U)`).  Once per iteration the library hands the weights w (N) and the unit columns U (d x N, ldU; copied to the
 * host once per component) to the caller's function on the calling thread; it writes the average to s (d; on entry the current q,
 * as in the reference, where the closure works in place on q) and returns 0.  One GPU only (a group handle uses its first GPU). */
typedef int (*tlsq_ga_avg_cb)(double* s, const double* w, const double* U, int64_t d, int64_t N, int64_t ldU, void* user);
typedef struct tlsq_ga_opts {
    double  tol;       /* NaN -> 1e-7   (:286) */
    int64_t iters;     /* <=0 -> 1000   (:286) */
    int32_t average;   /* TLSQ_GA_* */
    int32_t memory;    /* TLSQ_MEM_* for X, q0, Q */
    double  trim;      /* NaN -> 0.1    (:327) */
    uint64_t seed;
    tlsq_ga_avg_cb avg_cb;   /* TLSQ_GA_CALLBACK */
    void*   user;
} tlsq_ga_opts;
/* optional per-component reports (arrays of r entries; dq_hist is hist_capacity x r, column i = the `dq` of every
 * iteration of component i — what `verbose` prints at :300 — NaN padded) */
typedef struct tlsq_ga_info {
    int64_t* iters;
    int32_t* status;   /* 0 converged, 1 iteration limit */
    double*  dq;       /* last change */
    double*  dq_hist;
    int64_t  hist_capacity;
    double   ms_total, ms_loop;
    int64_t  passes;   /* total iterations = sweeps over U */
} tlsq_ga_info;
void tlsq_ga_opts_default(tlsq_ga_opts* o);
int tlsq_rpca_ga_f64(tlsq_handle h, const double* X, int64_t d, int64_t N, int64_t ldX, int64_t r,
                     const tlsq_ga_opts* opts, const double* q0, int64_t ldq0, double* Q, int64_t ldQ,
                     tlsq_ga_info* info);
/* Float32 observations (the reference's method is generic in the element type): X, q0, Q in float.  The panel is widened once on
 * the device and the iteration runs in the fp64 kernels (the library's small-arithmetic convention, as for ComplexF32 data); Q is
 * rounded to float on the way out.  Same options, report and return codes as tlsq_rpca_ga_f64. */
int tlsq_rpca_ga_f32(tlsq_handle h, const float* X, int64_t d, int64_t N, int64_t ldX, int64_t r,
                     const tlsq_ga_opts* opts, const float* q0, int64_t ldq0, float* Q, int64_t ldQ,
                     tlsq_ga_info* info);
/* the averages on their own (exported by the reference, src/TotalLeastSquares.jl:3; tested at
 * test/runtests.jl:466-490): s (d) = average of the columns of U (d x N, ldU) with weights w (N) */
int tlsq_ga_average_f64(tlsq_handle h, int average, double trim, const double* w, const double* U, int64_t d,
                        int64_t N, int64_t ldU, double* s, int memory);

/* ---- kernel-level entry points (DEVICE pointers; used by the parity tests and bench.py) -------
 * shrink sweep  (src/robustPCA.jl:188-192):  E = soft_th((D-A)+inv_mu*Y, thr) [max(E,0)]; Z=(D-E)+inv_mu*Y
 * update sweep  (src/robustPCA.jl:217-222):  [A=max(A,0)]; R=(D-A)-E; Y=Y+mu*R
 * All arrays contiguous M x N (ld = M), n = M*N elements. */
int tlsq_k_shrink_f64(tlsq_handle h, const double* D, const double* A, const double* Y, double* E,
                      double* Z, int64_t n, double inv_mu, double thr, int nonnegE);
int tlsq_k_update_f64(tlsq_handle h, const double* D, double* A, const double* E, double* Y,
                      double* R, int64_t n, double mu, int nonnegA);
int tlsq_k_shrink_f32(tlsq_handle h, const float* D, const float* A, const float* Y, float* E,
                      float* Z, int64_t n, float inv_mu, float thr, int nonnegE);
int tlsq_k_update_f32(tlsq_handle h, const float* D, float* A, const float* E, float* Y,
                      float* R, int64_t n, float mu, int nonnegA);
/* fused sweep used inside the loop for k >= 2: update of iteration k and shrink of iteration k+1 in one pass
 * (reads D, A, E, Y; writes R, Y, En = E_{k+1}, Zn = Z_{k+1}; 8 array passes instead of 11) */
int tlsq_k_update_shrink_f64(tlsq_handle h, const double* D, double* A, const double* E, double* Y, double* R,
                             double* En, double* Zn, int64_t n, double mu, int nonnegA, double inv_mu_next,
                             double thr_next, int nonnegE);
int tlsq_k_update_shrink_f32(tlsq_handle h, const float* D, float* A, const float* E, float* Y, float* R,
                             float* En, float* Zn, int64_t n, float mu, int nonnegA, float inv_mu_next,
                             float thr_next, int nonnegE);
/* the sweep of large panels (>= 2^26 elements): as above, but A = Tm * Vs' (Tm M x r ld M, Vs N x r ld N, r <= 32, M
 * even) is formed in registers and never stored: reads D, E, Y, writes R, Y, En, Zn (7 array passes) */
int tlsq_k_rebuild_update_shrink_f64(tlsq_handle h, const double* D, const double* Tm, const double* Vs, const double* E,
                                     double* Y, double* R, double* En, double* Zn, int64_t M, int64_t N, int64_t r,
                                     double mu, int nonnegA, double inv_mu_next, double thr_next, int nonnegE);
/* The E-free sweep every plain call runs (no E while the loop runs; src/robustPCA.jl:217-223 of iteration k and :188-192 of
 * iteration k+1 through the identities R_k = (Z_k - A_k) - Y_k / mu_k and Y_{k+1} = mu_k (Z_k - A_k)): reads D, Yin = Y_k
 * and Z = Z_k, writes Yout = Y_{k+1}, Z = Z_{k+1} in place and, when R is not NULL, R_k.  A_k = Tm * Vs' from its factors
 * (r <= 32, in registers) when A is NULL, read from A otherwise (clamped in place with nonnegA).  inv_mu = 1 / mu_k as the
 * element type rounds it.  sumsq (optional, 64 doubles, zeroed by the caller): their sum += ||R_k||_F^2. */
int tlsq_k_zsweep_f64(tlsq_handle h, const double* D, const double* Tm, const double* Vs, double* A, const double* Yin,
                      double* Yout, double* Z, double* R, int64_t M, int64_t N, int64_t r, double mu, double inv_mu,
                      int nonnegA, double inv_mu_next, double thr_next, int nonnegE, double* sumsq);
/* The same sweep and the Gram matrix G (N x N, ldG, both triangles) = Zout' Zout of the Z_{k+1} it writes, in one kernel
 * (fused.hip: the rows of Z_{k+1} feed the fp64 MFMA from LDS on their way to memory; split-K slabs summed in a fixed
 * order).  Zin = Z_k is read, Zout = Z_{k+1} written (the same buffer is allowed).  A_k always from its factors.  Serves
 * contiguous fp64 panels (ld = M) of N = 256 columns (even M >= 400000 rows) or N = 512 (>= 65536 rows: the kernel covers the
 * diagonal 256-column blocks, the off-diagonal block follows from the stored Zout), r <= 16, a threshold >= 0, 16-byte aligned, ldG = N;
 * TLSQ_ERR_UNSUPPORTED otherwise (rpca then runs k_zsweep and the Gram kernel one after the other).  hankel_y (optional): D is
 * the implicit Hankel matrix D[i, j] = hankel_y[i + j] for i < hankel_K, zero rows below (D itself is not read).
 * sumsq (optional, 72 doubles, zeroed by the caller): [0, 64) sum to ||R_k||_F^2, [64] = max |R_k[i, j]| as a bit pattern. */
int tlsq_k_zsweep_gram_f64(tlsq_handle h, const double* D, const double* Tm, const double* Vs, const double* Yin,
                           double* Yout, const double* Zin, double* Zout, double* R, int64_t M, int64_t N, int64_t r,
                           double mu, double inv_mu, int nonnegA, double inv_mu_next, double thr_next, int nonnegE,
                           double* sumsq, const double* hankel_y, int64_t hankel_K, double* G, int64_t ldG);
/* The E-free sweep for ranks 33..80 on an fp32 panel (sweeps.hip, k_zsweep_wide; rpca on BASELINE config 5): the statements of
 * k_zsweep (src/robustPCA.jl:217-222, then :188-192 of the next iteration) with A_k = T Vs' formed tile by tile on the fp32 MFMA
 * inside the sweep instead of read from a stored panel.  Z (M x N fp32, ld M) is the panel Z_k that T = Z Vg is taken from
 * (the fp32-MFMA product of opgram32.hip) AND the sweep's Z_k; Vg, Vs: N x r fp64 (ld N) - V_sel diag(g) and V_sel.
 * Writes Yout = Y_{k+1}, Zout = Z_{k+1} (Zout == Z: in place), R_k when R is not NULL, and Tm (M x r fp64, optional): the
 * factor T widened.  Serves M % 128 == 0, N % 128 == 0, M * N >= 2^26; TLSQ_ERR_UNSUPPORTED otherwise (rpca then stores A_k and
 * runs k_zsweep on it).  sumsq: as k_zsweep_gram (72 doubles, zeroed by the caller). */
int tlsq_k_zsweep_wide_f32(tlsq_handle h, const float* D, const double* Vg, const double* Vs, int64_t r, const float* Yin,
                           float* Yout, float* Z, float* Zout, float* R, double* Tm, int64_t M, int64_t N, float mu,
                           float inv_mu, int nonnegA, float inv_mu_next, float thr_next, int nonnegE, double* sumsq);
/* E = soft_th(D - A_prev + inv_mu * Y, thr) (:188-191), A_prev = Tm * Vs' (Aprev NULL, r <= 32; r = 0: zero) or read from
 * Aprev: the E a call returns, formed once after the E-free loop.  E may be the buffer Y lives in. */
int tlsq_k_final_e_f64(tlsq_handle h, const double* D, const double* Tm, const double* Vs, const double* Aprev,
                       const double* Y, double* E, int64_t M, int64_t N, int64_t r, double inv_mu, double thr, int nonnegA,
                       int nonnegE);
/* Matrix functions of a symmetric N x N device matrix (ld N, N <= 1024) by Newton-Schulz iterations on the MFMA GEMM
 * (matfun.hip): X = sign(C); W = B^(-1/2) for positive definite B with eigenvalues <= hi.  *iters = steps taken; returns
 * TLSQ_ERR_NOCONV when the iteration does not converge (an eigenvalue of C too close to zero / B not definite). */
int tlsq_k_matfun_sign_f64(tlsq_handle h, const double* C, int64_t N, double* X, int32_t* iters);
int tlsq_k_matfun_invsqrt_f64(tlsq_handle h, const double* B, int64_t N, double hi, double* W, int32_t* iters);
/* Rayleigh-Ritz of a nearly orthogonal block in one workgroup (subspace.hip, k_rr_small; p <= 32; device pointers): from
   B = Y'Y and Hg = Y'GY (p x p, ld p) the rotation C (p x p, ld p) with C'BC = I and C'HgC = diag(lam), by a symmetric
   orthonormalisation (Newton-Schulz inverse square root) and the eigenvector refinement of Ogita & Aishima started from the
   identity.  The first nt columns are the wanted ones; the other p - nt (pad columns) are orthonormalised and decoupled from
   them but not rotated among themselves, which is only acceptable while the pad block's eigenvalues (Gershgorin) stay
   below tau2.  status (8 doubles): [1] = 0 ok / 1 columns too far from orthogonal (C = diag(B)^-1/2) / 2 no convergence /
   3 a pad column may have reached tau2.  Replaces CholeskyQR2 + the Jacobi solver in warm subspace steps of the rpca loop
   (the `gesdd` work of src/robustPCA.jl:194). */
int tlsq_k_rr_small_f64(tlsq_handle h, const double* B, const double* Hg, int64_t p, int64_t nt, double tau2, double* C,
                        double* lam, double* status);
/* Y (N x p, ld N, fp64) = Z'(Z X) for an fp32 panel Z (M x N, ldZ) and X (N x p, ld N, fp64): the operator product of
 * large mode on the fp32 MFMA with fp64 fold-in (gemm.hip, op_gram_f32) - what the randomized hook's sketch is made of */
int tlsq_k_op_gram_f32(tlsq_handle h, const float* Z, int64_t M, int64_t N, int64_t ldZ, const double* X, int64_t p, double* Y);
/* G (N x N, ldG) = Z' Z for Z M x N (ldZ) — MFMA f64, deterministic split over rows */
int tlsq_k_gram_f64(tlsq_handle h, const double* Z, int64_t M, int64_t N, int64_t ldZ,
                    double* G, int64_t ldG);
/* the same for an fp32 panel, G in fp64.  mfma32 = 0: the panel is widened while staged, fp64 MFMA (what rpca uses up to
 * N = 2048); 1: fp32 MFMA, the 32-row partial sums folded into fp64 accumulators (large mode); -1: the library's choice */
int tlsq_k_gram_f32(tlsq_handle h, const float* Z, int64_t M, int64_t N, int64_t ldZ,
                    double* G, int64_t ldG, int mfma32);
/* C (M x Q, ldC) = Z (M x K, ldZ) * W (K x Q, ldW) */
int tlsq_k_gemm_nn_f64(tlsq_handle h, const double* Z, int64_t M, int64_t K, int64_t ldZ,
                       const double* W, int64_t Q, int64_t ldW, double* C, int64_t ldC);
/* C (M x Q, ldC) = T (M x K, ldT) * V' , V is Q x K (ldV) */
int tlsq_k_gemm_nt_f64(tlsq_handle h, const double* T, int64_t M, int64_t K, int64_t ldT,
                       const double* V, int64_t Q, int64_t ldV, double* C, int64_t ldC);
/* symmetric PSD eigen-decomposition of G (N x N): lam (N, descending), V (N x N, ldV; may be NULL) */
int tlsq_k_symeig_f64(tlsq_handle h, const double* G, int64_t N, int64_t ldG, double* lam,
                      double* V, int64_t ldV, int64_t* sweeps);
/* the same through the Cholesky-preconditioned route used inside the ALM loop (Jacobi on chol(G + delta I), no
 * eigenvector accumulation): eigenvectors of numerically-zero eigenvalues are returned as zero columns */
int tlsq_k_symeig_chol_f64(tlsq_handle h, const double* G, int64_t N, int64_t ldG, double* lam, double* V,
                           int64_t ldV, int64_t* sweeps);
/* R (N x N upper triangular, ldR) of the Householder TSQR factorisation Z = Q R of a tall panel Z (M x N, ldZ, M >= N):
 * LDS-staged Householder panel reduction over 256-row blocks, tree over the block factors, MFMA trailing updates.
 * R'R = Z'Z, and the singular values of R are those of Z to a few eps * sigma_max (what LAPACK's gesdd delivers at
 * src/robustPCA.jl:194) — unlike the Gram route, which squares the condition number. */
int tlsq_k_tsqr_f64(tlsq_handle h, const double* Z, int64_t M, int64_t N, int64_t ldZ, double* R, int64_t ldR);
/* thin SVD factors of Z through that route: S (N, descending), V (N x N, ldV; column i = right singular vector i):
 * TSQR, then one-sided Jacobi on R'.  Every singular value is accurate to a few eps * sigma_max. */
int tlsq_k_svd_r_f64(tlsq_handle h, const double* Z, int64_t M, int64_t N, int64_t ldZ, double* S, double* V,
                     int64_t ldV, int64_t* sweeps);
/* sigma_max of Z (M x N, ldZ) — the default `opnorm` */
int tlsq_k_opnorm_f64(tlsq_handle h, const double* Z, int64_t M, int64_t N, int64_t ldZ,
                      double* sigma_max);
/* max |x_i| over n contiguous elements — norm(Y, Inf) of src/robustPCA.jl:178 */
int tlsq_k_maxabs_f64(tlsq_handle h, const double* x, int64_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif /* TLSQ_H */
