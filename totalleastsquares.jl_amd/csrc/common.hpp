// Internal declarations shared by the HIP translation units of libtlsqhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <cmath>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../../include/tlsq.h"

namespace tlsq {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Comm;  // RCCL state (runtime.hip)
struct Stager;   // pinned staging pipeline of host-pointer calls (staging.hip)

struct Handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_b = nullptr;    // second stream: the Gram of the row chunks a fused sweep has finished (created on first use)
    hipEvent_t ev_b[9] = {};           // chunk c swept (0..7), Gram reduced (8); timing disabled
    hipEvent_t ev[34] = {};   // two banks of 16 phase marks (PhaseTimer) + [32] small readbacks + [33] Lanczos read-back
    std::string err;
    // grow-only device workspace, keyed by slot
    std::vector<DevBuf> ws;
    // pinned host scratch for small readbacks
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // pinned staging ring for small host->device uploads (index lists, scale factors): the copy is asynchronous
    // and the host may reuse its own buffer at once; see upload_async() in runtime.hip
    void* up_ring = nullptr;
    size_t up_bytes = 0, up_off = 0;
    // host-visible mailbox (coherent pinned memory, 32 KB) kernels write small results to; the host polls mailbox[0]
    double* mailbox = nullptr;       // host address
    size_t mailbox_bytes = 0;
    double* mailbox_dev = nullptr;   // device address of the same memory
    double mail_seq = 0.0;
    bool lz_multi_off = false;    // k_lanczos_multi timed out once on this handle (lanczos.hip)
    unsigned int lz_salt = 0;     // run counter: part of the granule tags of k_lanczos_multi
    void* lz_xch_clean = nullptr; // the granule buffer that has been cleared (WS_LZX)
    int lz_first_chunk = 0;       // hint for the next lanczos_begin (consumed there): pairs of its first chunk
    bool mail_counter_ready = false;   // the device-side arrival counter of k_ritz_finish has been cleared
    int64_t gram_tab2_nti = 0;
    // max |z| of an fp32 panel as a bit pattern, left by the kernel that wrote it (k_zsweep_wide) for the split Gram kernel that
    // reads it next (gram16.hip): absmax_panel == the panel it describes, nullptr = none.  The word lives in WS_H16S + 40.
    const void* absmax_panel = nullptr;
    bool gram_tab3_ready = false;      // the four-tile list of gram_offdiag_plan is on the device
    int64_t gram_tab_nti = 0;          // tile-order table in WS_GRAMTAB is the one for this many tile rows
    bool cert_ticket_ready = false;    // ... and the one of k_sq_norm
    Comm* comm = nullptr;
    int nranks = 1, rank = 0;
    // single-process multi-GPU group (tlsq_create_multi): the caller holds rank 0, subs[r - 1] is rank r.  multi_comm
    // is this rank's communicator of the group; it is attached as `comm` only while a group call runs (in_multi), so
    // that every other entry point uses the handle as a plain single-GPU one.
    std::vector<Handle*> subs;
    Comm* multi_comm = nullptr;
    int multi_n = 1, multi_rank = 0;
    bool in_multi = false;
    // warm start of the full Jacobi solver: WS_V holds the eigenvectors of the previous full decomposition
    int64_t warm_n = 0;      // its size (0 = nothing to reuse)
    // result of the last rpca_core call with ResolvedOpts::factors_out: A = out_Tm (M x out_r, ld M) * out_Vs' (N x out_r, ld N)
    // was left in factors (out_factors) instead of being written to the A panel
    bool out_factors = false;
    const double *out_Tm = nullptr, *out_Vs = nullptr;
    int64_t out_r = 0;
    int warm_uses = 0;       // consecutive warm starts (reset to a cold start now and then: drift control)
    Stager* stager = nullptr;   // created by the first large host <-> device transfer (staging.hip)
    // launches of the shape-dependent kernels since the last rpca_core entry (tlsq_rpca_info::kern_*)
    bool fused_warm_done = false;   // fused_zgram_warm has run on this handle's device
    int64_t kern_gram_h3 = 0, kern_zx_h = 0, kern_zty_h = 0, kern_zsweep_wide = 0, kern_fused_zgram = 0;
};

}  // namespace tlsq

struct tlsq_handle_s : tlsq::Handle {};

namespace tlsq {

int set_err(Handle* h, int code, const char* fmt, ...);

#define TLSQ_HIP(h, expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return tlsq::set_err((h), _e == hipErrorOutOfMemory ? TLSQ_ERR_OOM : TLSQ_ERR_HIP, \
                                 "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                                 __FILE__, __LINE__);                                       \
    } while (0)

#define TLSQ_TRY(expr)              \
    do {                            \
        int _s = (expr);            \
        if (_s < 0) return _s;      \
    } while (0)

// workspace slot: returns a device pointer of at least `bytes` (grow-only, contents not preserved on growth)
int ws_get(Handle* h, int slot, size_t bytes, void** out);


// ---- development switches (runtime.hip) ---------------------------------------------------------------------------
// Every alternative HIP path the library can be told to take is one entry of this list.  Values are
// set through tlsq_dev_set(name, value) - the tests and tools call it; a name that is not on the list is an error - and,
// only in a build with -DTLSQ_DEV_SWITCHES, from environment variables TLSQ_<NAME> read once at the first tlsq_create.
// The shipped build reads no environment variable at all: a leaked or mistyped variable cannot change which solver runs.
#define TLSQ_DEV_LIST_LIVE(X)                                                                                             \
    X(DEBUG) X(DEBUG_HASH) X(PHASE_TIMING) X(WS_POISON) X(FORCE_COMM) X(FORCE_LOCALGROUP) X(FAIL_RANK) X(NO_ZSWEEP)   \
    X(FUSED_REBUILD) X(RUS_ROWS) X(RUS_CT) X(SWEEP_GRID) X(RSKIP_MARGIN) X(MATFUN_DEFL) X(NO_QUINTIC)                 \
    X(NO_CERT_ASYNC) X(FAIL_CERT_AT) X(NO_SPEC_REBUILD) X(CERT_EARLY) X(CERT_PRIO) X(NO_SMALL_MM) X(COLD_Q)           \
    X(WARM_Q0) X(IMPLICIT_GRAM) X(LAZY_HANKEL) X(IMPLICIT_HANKEL) X(PAD) X(NO_MAILBOX) X(NO_FUSED_ZGRAM)              \
    X(FUSED_ZGRAM_MINROWS) X(FUSED_ABLATE) X(FUSED_ZGRAM_N512) X(OPGRAM_OLD) X(OPGRAM_H3) X(GRAM_H3)                  \
    X(NO_WIDE_SWEEP) X(NO_SLICED_EIG) X(SLICE_NORMWISE) X(HANKEL_STRUCT) X(SLICE_TARGET) X(COLD_TOP) X(COLD_TOL0)         \
    X(LZ_MULTI) X(RITZ_SORT) X(PAD_PROJECT) X(HOST_TRACE) X(SWEEP_TIMING_STRIDE) X(SLICE_MIDJ)
// Ablation switches: every one of them selects a path that was measured against its successor and is kept for that comparison
// (docs/HISTORY.md).  No committed test or tool uses them; the SHIPPED library does not accept them - 
// tlsq_dev_set answers TLSQ_ERR_ARG as for an unknown name, so dev_get() of these is always "not set" - only a build with
// -DTLSQ_DEV_SWITCHES (TLSQ_EXTRA_FLAGS of the build recipe) does.  48 live switches, 65 ablation switches.
#define TLSQ_DEV_LIST_ABLATION(X)                                                                                         \
    X(NO_FUSED_SWEEP) X(NO_FIRST_SHRINK) X(NO_FUSED_REBUILD) X(NO_REBUILD_STORE)                                      \
    X(NO_MAX_BOUND) X(LAST_GUESS) X(NO_POWER_LB) X(NO_POWER_START) X(POWER_LEVELS) X(FULL_EIG) X(NO_GRAM_DENSE)       \
    X(NO_MATFUN_ROUTE) X(MATFUN_SYM) X(MATFUN_COND) X(NO_DEFLATED_CERT) X(NO_DEEP_POWERS) X(NO_POWER_CERT)            \
    X(NO_CERT_OVERLAP) X(NO_DEFLATED_SVD) X(NO_FUSED_DEFLATE) X(NO_RR_FAST) X(NO_RR_BLOCKED) X(NO_U_POLISH)           \
    X(NO_GX_REUSE) X(COLD_CGS2) X(NO_ONEPASS) X(NO_CHOLQR) X(NO_BLOCKED_CGS2) X(JACOBI2) X(NO_JACOBI_REG)             \
    X(JACOBI_RPL) X(NO_CHOL) X(NO_SYMM_MFMA) X(GRAM_OLD) X(GRAM_RHO) X(GRAM_SPLIT) X(GRAM_ROWWISE) X(GRAM_F32MFMA)    \
    X(GEMM_WGS) X(HOOK_SKETCH) X(NO_F32_SKINNY) X(OVERLAP_CHUNKS) X(OVERLAP_LDS) X(OVERLAP_NOPRIO) X(NO_TSMM)         \
    X(NO_TSMM_SELV) X(TSMM_MAXR) X(NO_TSMM_SEL) X(UNHANKEL_FACTORS) X(GA_BLOCKS) X(GA_SOLO) X(NO_FUSED_GR)            \
    X(NO_HOOK_ZQ) X(GRAM_H3_FOLD) X(HOOK_CLASSIC) X(HOOK_POWER) X(HOOK_COLD) X(HOOK_CGS2) X(HOOK_ORTH_ALL)            \
    X(HOOK_PAD_REFRESH) X(COLD_GROW) X(SLICE_SCHED) X(SLICE_LEVELS) X(SLICE_L0) X(NO_MAXABS_ASYNC)
#define TLSQ_DEV_LIST(X) TLSQ_DEV_LIST_LIVE(X) TLSQ_DEV_LIST_ABLATION(X)
enum DevKey {
#define TLSQ_DEV_ENUM(n) DEV_##n,
    TLSQ_DEV_LIST(TLSQ_DEV_ENUM)
#undef TLSQ_DEV_ENUM
    DEV_COUNT
};
// the first DEV_LIVE_COUNT keys are the live ones
enum { DEV_LIVE_COUNT = 0
#define TLSQ_DEV_ONE(n) +1
    TLSQ_DEV_LIST_LIVE(TLSQ_DEV_ONE)
#undef TLSQ_DEV_ONE
};
const char* dev_get(DevKey k);   // the value as a string, nullptr when the switch is not set
inline bool dev_is(DevKey k, char c) {
    const char* e = dev_get(k);
    return e && e[0] == c;
}
// name with or without the TLSQ_ prefix; value == nullptr clears the switch.  TLSQ_ERR_ARG for a name not on the list.
int dev_set(const char* name, const char* value);
void dev_load_env();             // (no-op unless built with -DTLSQ_DEV_SWITCHES)
void host_mark(const char* tag);   // HOST_TRACE=1: a wall-clock mark of the host side of the call (runtime.hip)
void host_trace_dump();
// DEBUG_HASH=1: "[hash] <tag> <64-bit FNV-1a of the buffer>" on stderr after a stream synchronisation - two runs of the same
// call must print identical lines; the first line that differs names the kernel that is not reproducible
struct Handle;
void dbg_hash(Handle* h, const char* tag, const void* dev_ptr, size_t bytes, long long k = -1);

enum WsSlot {
    WS_D = 0, WS_A, WS_E, WS_Y, WS_Z, WS_R,      // M x N panels
    WS_G, WS_B, WS_V, WS_VG, WS_VS, WS_T,        // N x N / M x r
    WS_SLAB,                                     // split-K partial slabs
    WS_SCAL,                                     // small scalars / counters
    WS_LAM, WS_AUX0, WS_AUX1, WS_AUX2, WS_AUX3, WS_AUX4,
    WS_SX, WS_SQ, WS_SGQ, WS_SXN, WS_SGX, WS_SH, WS_SS, WS_SHB, WS_GD,   // subspace iteration panels
    WS_DT, WS_AT, WS_ET, WS_UT,
    WS_V2, WS_VC, WS_E2, WS_Z2, WS_BATCH0, WS_BATCH1, WS_BATCH2, WS_BATCH3, WS_BATCH4, WS_G2, WS_LZOP, WS_OPT, WS_OPW, WS_T32, WS_VS32, WS_H16, WS_L16, WS_H16S, WS_OPSC,
    WS_PW,   // persistent power-iteration vector of the cost evaluation
    WS_GRAMTAB,   // tile order of the Gram kernel (gemm.hip, gram_kc)
    WS_GRAMTAB2,  // ... of the fp32-MFMA Gram kernel (diagonal tiles included)
    WS_GRAMTAB3, WS_SLAB2,   // off-diagonal block of an N = 512 Gram matrix beside the fused sweep kernel: tile list, slabs
    WS_HKSUM,     // soft_hankel! on row shards: anti-diagonal sums and counts of the whole matrix (solver.hip)
    WS_CP1, WS_CP2, WS_CPART,   // power certificate: S^2, S^4, norm partials
    WS_T2, WS_VS2, WS_VS3, WS_T3,
    WS_MF0, WS_MF1, WS_MF2, WS_MF3, WS_MFP,   // matrix-function route (solver.hip): N x N iterates, partials of k_mf_stats      // rebuild factors of the E-free loop (the factors of A_{k-1} are kept: WS_T2/WS_VS2 and WS_T/WS_VS3 in turn)
    WS_QRW, WS_QRY, WS_QRT, WS_QRP, WS_QRG, WS_QRS,   // TSQR (tsqr.hip): working copy, reflectors, T factors, packed / gathered / stacked factors
    WS_GA_X, WS_GA_U, WS_GA_AUX, WS_GA_PART, WS_GA_MASK, WS_GA_KEYS, WS_GA_IDX, WS_GA_TMP, WS_GA_IO, WS_GA_Q, WS_GA_F32, WS_GA_X64, WS_GA_Q64,   // rpca_ga (grassmann.hip)
                                                             // two-level (precise) decomposition                                            // transposed problem (M < N)
    WS_G3,             // second Gram buffer of the speculative loop (solver.hip: the Gram of Z_{k+1} is queued while G_k is still read)
    WS_C32_F, WS_C32_D, WS_C32_A, WS_C32_E, WS_C32_U, WS_C32_V, WS_C32_S,   // ComplexF32 entry (api.hip): float staging, widened panels
    WS_CBS, WS_CBR,   // svd / opnorm callbacks on row shards (solver.hip): send block, gathered panel
    WS_VTOUT,   // the returned Vt of a host-pointer call on its way out (entry.hip)
    WS_SL_BUF, WS_SL_TAB,   // spectrum slicer in front of the accurate route's Jacobi (sliced.hip): N x N iterates, block-pair table
    WS_LZX,   // granule buffers of k_lanczos_multi (lanczos.hip): written by nothing else
    WS_UPOL, WS_UPB,   // orthonormal polish of the derived singular vectors (solver.hip): second M x d panel, d x d Gram + correction
    WS_COUNT
};

// An implicit Hankel source (src/robustPCA.jl:76-92): D[k, c] = x[k lag + l, d] with c = l Dch + d; channel d of the series
// starts ldx elements after channel d - 1.  The default describes one channel with lag 1: D[i, j] = y[i + j].
struct HankelGeom {
    int32_t lag = 1, Dch = 1;
    int64_t ldx = 0;
};
__host__ __device__ inline int64_t hankel_coff(const HankelGeom& g, int col) {   // offset of column `col` within row 0
    return g.Dch == 1 ? (int64_t)col : (int64_t)(col / g.Dch) + (int64_t)(col % g.Dch) * g.ldx;
}

// ---------------- sweeps.hip ----------------
template <typename T>
int launch_shrink(Handle* h, const T* D, const T* A, const T* Y, T* E, T* Z, int64_t n, T inv_mu,
                  T thr, int nonnegE);
template <typename T>
int launch_first_shrink(Handle* h, const T* D, T* Y, T* E, T* Z, int64_t n, T s, T inv_mu, T thr, int nonnegE);
template <typename T>
int launch_update(Handle* h, const T* D, T* A, const T* E, T* Y, T* R, int64_t n, T mu, int nonnegA);
// fused update(k) + shrink(k+1): R_k, Y_k, E_{k+1} (En), Z_{k+1} (Zn) in one pass
template <typename T>
// sumsq (optional, device, 64 doubles): their sum += ||R_k||_F^2 (atomic adds: a bound, never a result)
// R may be nullptr (residual not stored; launch_residual recomputes it if it is needed after all)
int launch_update_shrink(Handle* h, const T* D, T* A, const T* E, T* Y, T* R, T* En, T* Zn, int64_t n, T mu,
                         int nonnegA, T inv_mu_n, T thr_n, int nonnegE, double* sumsq = nullptr,
                         double* zero_slots = nullptr);   // zero_slots: 64 doubles cleared for the next sweep
// Y = D / s  (src/robustPCA.jl:181), contiguous n
template <typename T>
int launch_div_scalar(Handle* h, const T* D, T* Y, int64_t n, T s);
// out[0] = max |x_i|  (device scalar, as double bits in a uint64 slot)
template <typename T>
int launch_maxabs(Handle* h, const T* x, int64_t n, double* host_out);
template <typename T>
int launch_maxabs_begin(Handle* h, const T* x, int64_t n);   // ... queued on the second stream, read back by launch_maxabs_end
int launch_maxabs_end(Handle* h, double* host_out);
template <typename T>
int launch_clamp_nonneg(Handle* h, T* A, int64_t n);
template <typename T>
int launch_residual(Handle* h, const T* D, const T* A, const T* E, T* R, int64_t n);
template <typename T>
int launch_residual_hankel(Handle* h, const T* y, int64_t K, const T* A, const T* E, T* R, int64_t M, int64_t N,
                           HankelGeom hg = HankelGeom());   // R = (D - A) - E
// 64 doubles -> the handle's mailbox ([8..72)), published with sequence number seq
int launch_publish_slots(Handle* h, const double* slots, double seq);
// rebuild (A = Tm Vs', kept in registers) + update(k) + shrink(k+1): 7 panel passes, A is not stored
// E-free sweep and its companions (sweeps.hip: k_zsweep ...)
template <typename T>
int launch_zsweep(Handle* h, const T* D, const double* Tm, const double* Vs, T* A, const T* Yin, T* Yout, T* Z, T* R,
                  int64_t M, int64_t N, int64_t r, T mu, T inv_mu, int nonnegA, T inv_mu_n, T thr_n, int nonnegE,
                  double* sumsq, double* zero_slots, const T* hankel_y = nullptr, int64_t hankel_K = 0, int64_t row0 = 0,
                  int64_t row1 = 0, int maxslot = -1, HankelGeom hg = HankelGeom(), T* Zout = nullptr);   // Zout: Z_{k+1} out of place
template <typename T>
int launch_final_e(Handle* h, const T* D, const double* Tm, const double* Vs, const T* Aprev, const T* Y, T* E, int64_t M,
                   int64_t N, int64_t r, T inv_mu, T thr, int nonnegA, int nonnegE, const T* hankel_y = nullptr,
                   int64_t hankel_K = 0, HankelGeom hg = HankelGeom());
template <typename T>
int launch_residual_from_y(Handle* h, const T* Y1, const T* Y0, T* R, int64_t n, T inv_mu);
template <typename T>
int launch_z_from_y(Handle* h, const T* A, const T* Y1, T* Z, int64_t n, T inv_mu);
// the returned E without the factors of A_{k-1}: D - A - (Y1 - Y0) / mu, or D - Z + Y / mu; rounding noise snapped to zero
template <typename T>
int launch_e_from_residual(Handle* h, const T* D, const T* A, const T* Y1, const T* Y0, T* E, int64_t n, T inv_mu);
template <typename T>
int launch_e_from_z(Handle* h, const T* D, const T* Z, const T* Y, T* E, int64_t n, T inv_mu);
template <typename T>
bool rebuild_update_shrink_ok(const T* D, const T* E, T* Y, T* R, T* En, T* Zn, int64_t M, int64_t N, int64_t r);
template <typename T>
int launch_rebuild_update_shrink(Handle* h, const T* D, const double* Tm, const double* Vs, const T* E, T* Y, T* R,
                                 T* En, T* Zn, int64_t M, int64_t N, int64_t r, T mu, int nonnegA, T inv_mu_n, T thr_n,
                                 int nonnegE, double* sumsq, double* zero_slots = nullptr, const T* hankel_y = nullptr,
                                 int64_t hankel_K = 0, int64_t row0 = 0, int64_t row1 = 0,   // rows [row0, row1), default all
                                 size_t pad_lds = 0,   // unused dynamic LDS per workgroup (caps the residency per CU)
                                 HankelGeom hg = HankelGeom());
// A (M x N, ld M) = Tm Vs' (Tm M x r fp64, Vs N x r): the streaming-store form of the rebuild (r <= 32, even M)
template <typename T>
bool rebuild_store_ok(const T* A, int64_t M, int64_t N, int64_t ldA, int64_t r);
template <typename T>
int launch_rebuild_store(Handle* h, const double* Tm, const double* Vs, T* A, int64_t M, int64_t N, int64_t r);
// dst (N x M, ld ldd) = src' for src (M x N, ld lds)
template <typename T>
int launch_transpose(Handle* h, const T* src, int64_t lds, int64_t M, int64_t N, T* dst, int64_t ldd);
// out = a - b
template <typename T>
int launch_diff(Handle* h, const T* a, const T* b, T* out, int64_t n);
// Vt[p + j ld] = (T) V[j + order[p] N] for p < d, j < N (V: N x N fp64, ld N): the returned right singular vectors, sorted
template <typename T>
int launch_vt_out(Handle* h, const double* V, int64_t N, const int32_t* order_dev, int64_t d, T* Vt, int64_t ld);
// dst[i] = (Tdst) src[i]
template <typename TS, typename TD>
int launch_convert(Handle* h, const TS* src, TD* dst, int64_t n);

// a short column selection with one weight per column, passed to kernels by value (no upload, no extra command)
struct SelWeights {
    int32_t sel[32];
    double w[32];
};
// The same list in device memory, written by the last workgroup of k_ritz_finish from the Ritz values it has just formed
// (count of values >= 1/mu, their weights (sigma - 1/mu) / sigma: src/robustPCA.jl:198, :205-213) so that the factor product
// of the rebuild can be queued before the host has seen them.  ok = 0: the step is not one the host will accept as it stands
// (values out of order, a declined Rayleigh-Ritz kernel, non-finite numbers): the product is skipped.
struct SpecCtrl {
    int32_t ok, r;
    int32_t pad[2];
    SelWeights sw;
};

// ---------------- gemm.hip ----------------
// Cm[j + i*ldc] = sum_k Aop(i,k) * Bop(k,j), i<P, j<Q, k<K
//   A_KC: Aop(i,k) = A[k + i*lda]   else  A[i + k*lda]
//   B_KC: Bop(k,j) = B[k + j*ldb]   else  B[j + k*ldb]
// `splits`>1: deterministic split over K through slabs + fixed-order reduction.
// `symmetric`: only tiles with ti>=tj are computed, the reduction mirrors (Gram).
int gemm_f64(Handle* h, bool A_KC, bool B_KC, const double* A, int64_t lda, const double* B,
             int64_t ldb, double* C, int64_t ldc, int64_t P, int64_t Q, int64_t K, bool symmetric);
int gram_f64(Handle* h, const double* Z, int64_t M, int64_t N, int64_t ldZ, double* G, int64_t ldG);
// mixed-precision storage: operands / result may be fp32 in memory (x_f32 != 0), arithmetic is fp64 MFMA.
// instantiated operand type pairs (A,B): (f64,f64), (f32,f32), (f64,f32); layouts (KC,KC), (KC,MN), (MN,MN).
// skip (device, optional): a non-zero skip[0] turns the launches into no-ops.  normpart / normblocks (symmetric only):
// the reduction also leaves *normblocks partial sums of ||C||_F^2 in normpart (at most 2048; add them in order).
int gemm_mixed(Handle* h, bool A_KC, bool B_KC, const void* A, int a_f32, int64_t lda, const void* B, int b_f32,
               int64_t ldb, void* C, int c_f32, int64_t ldc, int64_t P, int64_t Q, int64_t K, bool symmetric,
               const double* skip = nullptr, double* normpart = nullptr, int* normblocks = nullptr);
// T (M x r, fp64) = Z (M x K, fp32 or fp64) * W (K x r), r <= 96: Z streamed once, MFMA fed from global memory
int tsmm_mixed(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* W, int64_t ldw, double* Tout, int64_t ldt,
               int64_t M, int64_t K, int64_t r);
// the same with W = V[:, sel] diag(w) (r <= 32) gathered and packed in one pass; Vs (optional, K x r) = V[:, sel]
int tsmm_sel(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* V, const SelWeights& sw, double* Vs, double* Tout,
             int64_t ldt, int64_t M, int64_t K, int64_t r);
// ... with the list (and r) read from device memory when the kernel runs (k_tsmm_selv only: K a multiple of 4); nct = 1 / 2
// accumulator tiles of 16 columns are compiled in: the launch does nothing when ctrl->ok == 0 or ctrl->r > 16 nct
int tsmm_sel_dev(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* V, const SpecCtrl* ctrl, int nct, double* Vs,
                 double* Tout, int64_t ldt, int64_t M, int64_t K);
// Y (N x p, fp64) = Z' * T  (Z: M x N fp32/fp64, T: M x p fp64): column dots for p <= 8, the tiled MFMA kernel beyond
int ztmm_mixed(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* Tm, int64_t ldt, double* Y, int64_t ldy,
               int64_t M, int64_t N, int64_t p);
// Y (N x p, fp64) = Z'(Z X) for an fp32 panel on the fp32 MFMA (fp64 fold-in), p <= 96: the large-mode operator product
int op_gram_f32(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* X, int64_t ldx, double* Y,
                int64_t ldy, int64_t p);
// ... its second form (opgram32.hip: the block through LDS, the panel straight into the MFMA fragments) for aligned shapes
bool op_gram_f32_fast_ok(const float* Z, int64_t ldz, int64_t M, int64_t N, int64_t p);
// ... its third form (opgram16.hip): every operand split in registers into two fp16 numbers, products on the fp16 MFMA - when the
// panel's maximum is known (zmax_bits: device word holding the float bits of max |Z|)
int op_gram_f32_h3(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* X, int64_t ldx, float* t32, double* Y,
                   int64_t ldy, int64_t p, const unsigned int* zmax_bits);
int tsmm_f32_h3(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* X, int64_t ldx, int64_t p, int lw,
                float* t32, const unsigned int* zmax_bits, unsigned int** tmax_out);   // (its first half alone: T32 = Z X)
int absmax_bits_f32(Handle* h, const float* Z, int64_t n, unsigned int* out_bits);   // (gram16.hip: max |z| as float bits)
// G = Z'Z of an fp32 panel on the fp16 MFMA at fp32 accuracy (two fp16 planes, three products; gram16.hip)
int gram_tile_table(Handle* h, int64_t nti, const int32_t** tab_out);   // (gemm.hip)
bool gram_h3_ok(const float* Z, int64_t ld, int64_t N, int64_t K);
int gram_h3(Handle* h, const float* Z, int64_t ld, double* G, int64_t ldg, int64_t N, int64_t K);
// the factors of A_k in fp32 for the wide sweep (ranks 33..80, fp32 panels; opgram32.hip) and that sweep (sweeps.hip)
bool wide_factors_ok(const float* Z, int64_t ldz, int64_t M, int64_t N, int64_t r);
int wide_factors_f32(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* Vg, const double* Vs,
                     int64_t r, double* Tm, const float** T32_out, const float** Vs32_out, int* lw_out);
int wide_factors_from_zq(Handle* h, const float* ZQ, int64_t M, int64_t N, int64_t p, const double* S_dev, const int32_t* cols_dev,
                         const double* g_dev, const double* Vs, int64_t r, double* Tm, const float** T32_out,
                         const float** Vs32_out, int* lw_out);
bool zsweep_wide_ok(int64_t M, int64_t N, int64_t r);
int launch_zsweep_wide(Handle* h, const float* D, const float* T32, int64_t ldt, const float* Vs32, int64_t r, const float* Yin,
                       float* Yout, float* Z, float* Zout, float* R, int64_t M, int64_t N, float mu, float inv_mu, int nonnegA,
                       float inv_mu_n, float thr_n, int nonnegE, double* sumsq, double* zero_slots, int maxslot,
                       bool leave_absmax = false);   // leave_absmax: max |Z_{k+1}| for the next split Gram (Handle::absmax_panel)
int op_gram_f32_fast(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const float* wt, float* t32, double* Y,
                     int64_t ldy, int64_t p);
// the Gram of a K-contiguous operand as plan / per-chunk launch / reduction (gemm.hip)
struct GramPlan {
    int64_t N = 0, K = 0, nti = 0, nsplit_o = 1, nsplit_d = 1, kchunk_o = 0, kchunk_d = 0;
    int nchunks = 1, z_f32 = 0;
    double* slab = nullptr;
    const int32_t* order = nullptr;
};
int gram_plan(Handle* h, int z_f32, int64_t N, int64_t K, int nchunks, GramPlan* pl);
int gram_launch_chunk(Handle* h, hipStream_t st, const GramPlan& pl, const void* Z, int64_t ld, int64_t rows, int c,
                      const double* skip = nullptr);
int gram_reduce(Handle* h, hipStream_t st, const GramPlan& pl, double* G, int64_t ldg, const double* skip = nullptr,
                double* normpart = nullptr, int* normblocks = nullptr);
// N = 512 beside the fused sweep kernel: the off-diagonal 256 x 256 block from the stored panel, and the reduction of both slab sets
int gram_offdiag_plan(Handle* h, int64_t N, int64_t K, GramPlan* pl);
int gram_offdiag_launch(Handle* h, hipStream_t st, const GramPlan& pl, const double* Z, int64_t ld, int64_t rows);
int gram_reduce2(Handle* h, hipStream_t st, const GramPlan& plA, const GramPlan& plB, double* G, int64_t ldg);
int gram_any(Handle* h, const void* Z, int z_f32, int64_t M, int64_t N, int64_t ldZ, double* G, int64_t ldG,
             int mfma32 = -1);   // fp32 panels: -1 library's choice, 0 fp64 MFMA on widened operands, 1 fp32 MFMA + fp64 fold-in

// ---------------- fused.hip ----------------
// The E-free sweep and the Gram matrix of the Z_{k+1} it writes in one kernel (fp64 panels of 256 / 512 columns, rank <= 16):
// fused_zgram_ok says whether a sweep qualifies, fused_zgram_plan sizes the K split and the slabs (a GramPlan: gram_reduce
// sums them), launch_fused_zgram queues the kernel on the handle's stream.  Arguments as launch_zsweep (Zin != Zout allowed);
// first = true: the first shrink instead (launch_first_shrink: Y_1 = D / s_div, Z_1) with the Gram of Z_1.
bool fused_zgram_ok(int64_t M, int64_t N, int64_t r, const void* D, const void* Yin, const void* Yout, const void* Zin,
                    const void* Zout, const void* R, bool hankel, double thr_n, HankelGeom hg = HankelGeom());
int fused_zgram_plan(Handle* h, int64_t M, int64_t N, GramPlan* pl);
int fused_zgram_finish(Handle* h, const GramPlan& pl, const double* Zout, int64_t M, int64_t N, double* G);   // slabs (+ the off-diagonal block at N = 512) -> G
bool fused_zgram_shape_ok(int64_t M, int64_t N, bool hankel);
int fused_zgram_warm(Handle* h);   // first-launch costs of the kernel's instantiations (once per handle)
int launch_fused_zgram(Handle* h, const GramPlan& pl, const double* D, const double* Tm, const double* Vs, const double* Yin,
                       double* Yout, const double* Zin, double* Zout, double* R, int64_t M, int64_t N, int64_t r, double mu,
                       double inv_mu, int nonnegA, double inv_mu_n, double thr_n, int nonnegE, double* sumsq, double* zero_slots,
                       const double* hankel_y, int64_t hankel_K, int maxslot, bool first = false, double s_div = 1.0,
                       bool gram_of_r = false);

// ---------------- matfun.hip ----------------
// sign function / inverse square root of small symmetric matrices by Newton-Schulz iterations (N x N fp64, ld N; products on
// the symmetric MFMA GEMM).  *ok = false: not converged within max_iters (an eigenvalue too close to zero) or not finite.
int matfun_sign(Handle* h, const double* C, int64_t N, double* X, double* W1, double* W2, int max_iters, int* iters, bool* ok);
int matfun_invsqrt(Handle* h, const double* B, int64_t N, double hi, double* Z, double* Y, double* T, double* W, int max_iters,
                   int* iters, bool* ok);
int matfun_trace_norm(Handle* h, const double* X, int64_t N, double* trace, double* norm_inf);
bool sq_norm_blk_ok(int64_t N);
int launch_sq_norm_blk(Handle* h, const double* S, int64_t N, double* mailbox_dev, unsigned int* ticket, double seq, int ntile);
int matfun_square(Handle* h, const double* A, double* C, int64_t N, bool* ok);   // C = A^2, symmetric A
int matfun_power_start(Handle* h, const double* G, int64_t N, double* P1, double* P2, int levels, const double** out, bool* ok);   // (G / tr G)^(2^levels)
int matfun_stats(Handle* h, const double* X, int64_t N, double out[3]);   // { ||X - I||_F^2, trace(X), ||X||_inf } of a symmetric X
int matfun_axpbi(Handle* h, const double* X, double* Y, int64_t N, double a, double b);   // Y = a X + b I
int matfun_mul(Handle* h, const double* A, const double* B, double* C, int64_t N);        // commuting symmetric A, B
// PhiT = ((I - Xs Xs') F + Xs diag(w) Xs')' for symmetric F, Y = F Xs (see matfun.hip)
int matfun_phi(Handle* h, const double* F, const double* Xs, const double* Y, const SelWeights& sw, int64_t r, int64_t N,
               double* PhiT);
int matfun_lin2(Handle* h, const double* X1, double a1, const double* X2, double a2, double b, double* Y, int64_t N);   // Y = a1 X1 + a2 X2 + b I

// ---------------- jacobi.hip ----------------
// One-sided block Jacobi on the square matrix G (N x N, ld N): on return B = G*V has orthogonal
// columns, lam_dev[i] = ||B[:,i]|| (unsorted), V orthogonal (ld N) unless want_v == false.
// B and V are workspace buffers owned by the caller (N*N each).
// async_small: when the matrix fits the single-workgroup path, do not read the sweep count back (no host sync).
int jacobi_mid_blocks_f64(Handle* h, const double* T, int64_t N, const std::vector<std::pair<int, int>>& blocks, double* W, double* lam,
                          int64_t* sweeps_out);   // one-sided Jacobi on the diagonal blocks of T, one workgroup each (k <= 96)
int symeig_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double* B, double* V,
               bool want_v, double* lam_dev, int64_t* sweeps_out, bool async_small = false,
               bool warm_v = false, bool two_sided = false,    // two_sided (N <= 64): rotate G itself (absolute accuracy)
               double rot_tol = 0.0);   // > 0 (N <= 96): rotation threshold |cos| of a column pair instead of 2 eps sqrt(N)
// warm_v: V holds the previous decomposition's eigenvectors (orthogonal): start from B = G*V.
// Vg[:,p] = g[p] * V[:,sel[p]], Vs[:,p] = V[:,sel[p]]  for p < r  (all N x r, ld N)
// selection + weights small enough to travel as kernel arguments (r <= 32)
int launch_gather_scale_arg(Handle* h, const double* V, int64_t N, const SelWeights& sw, int64_t r, double* Vg,
                            double* Vs);
// GD = scale (G - sum_k w_k X[:, sel_k] X[:, sel_k]') (subspace.hip)
int launch_deflate_sel(Handle* h, const double* G, int64_t ldG, const double* X, const SelWeights& sw, double* GD, int64_t N,
                       int64_t r, double scale);
int launch_gather_scale(Handle* h, const double* V, int64_t N, const int32_t* sel_dev,
                        const double* g_dev, int64_t r, double* Vg, double* Vs);

// ---------------- cholesky.hip ----------------
// L (N x N, ld N) = chol(G + delta I) with delta = 2 N eps max_i G_ii (stats_dev[1] = delta); T = N x N scratch.
int cholesky_shifted(Handle* h, const double* G, int64_t ldG, int64_t N, double* L, double* T, double* stats_dev);
int launch_normalize_cols(Handle* h, const double* B, int64_t N, double* V, double* sig);
// Eigen-decomposition of the PSD matrix G through its Cholesky factor: one-sided Jacobi on L (no eigenvector
// accumulation), V = normalised columns of the rotated L, sig_dev[i] = singular values of L = sqrt(lambda_i + delta).
// Columns belonging to numerically-zero eigenvalues come back as zero vectors.
int symeig_chol_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double* B, double* V, double* sig_dev,
                    double* delta_host, int64_t* sweeps_out);
// One-sided Jacobi on the columns of a square factor B (N x N, ld N, in place): V = normalised orthogonal columns,
// sig_dev = their norms (unsorted).  floor_rel * ||B||_F = norm below which a column takes no part (0: none).
int jacobi_factor_f64(Handle* h, double* B, int64_t N, double* V, double* sig_dev, double floor_rel,
                      int64_t* sweeps_out);

// ---------------- hankelop.hip ----------------
// products with the Hankel matrix H[k, j] = y[k + j] (k < K, j < n; one channel, lag 1) that is never stored: G = H'H from n
// lagged autocorrelation sums, Tm = H X as r FIR filters (r <= 32, n <= 1024)
bool hankel_structured_ok(int64_t K, int64_t n, int64_t r);
template <typename T>
int hankel_gram(Handle* h, const T* y, int64_t K, int64_t n, double* G);
template <typename T>
int hankel_times(Handle* h, const T* y, int64_t K, int64_t Kp, int64_t n, const double* X, int64_t ldx, int64_t r, double* Tm,
                 int64_t ldt);

// ---------------- sliced.hip ----------------
// The same decomposition with the spectrum cut into slices first (sign functions on the MFMA, Jacobi sweeps inside the slices,
// then over all pairs): N a multiple of 128 in [256, 1024].  *used = false: a guard declined - call symeig_chol_f64.
bool symeig_sliced_ok(int64_t N);
int symeig_sliced_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double* B, double* V, double* sig_dev, double* delta_host,
                      int64_t* sweeps_out, double lam_hi, int n_out, double val_out, double bulk_hi, bool* used);

// ... normwise on G itself (no factor): V and the eigenvalues lam_dev, certified to 8 N eps lam_hi - for the deflated panel of the
// returned `s` (see sliced.hip)
int symeig_sliced_normwise_f64(Handle* h, const double* G, int64_t N, double* V, double* lam_dev, int64_t* sweeps_out, double lam_hi,
                               int n_out, double val_out, double bulk_hi, bool* used);

// ---------------- tsqr.hip ----------------
// B (N x N, ld N) = R' (lower triangular) of the Householder TSQR factorisation Z = Q R; Z (M x N, ld ldz, fp32 when
// z_f32) is not modified.  On a handle with a communicator Z is the local row shard and R belongs to the whole matrix.
int tsqr_lt(Handle* h, const void* Z, int z_f32, int64_t M, int64_t N, int64_t ldz, double* B);
// R (N x N, upper triangular, ld ldR)
int tsqr_r(Handle* h, const double* Z, int64_t M, int64_t N, int64_t ldz, double* R, int64_t ldR);

// ---------------- lanczos.hip ----------------
// lambda_max(G) to relative accuracy rel_tol (residual bound of the Ritz pair); returns 1 (and the best
// estimate) if not reached within max_steps so the caller can fall back to the Jacobi solver.
// accept_below > 0: also stop (status 0) as soon as 2.5 * estimate < accept_below after >= 16 steps.
// stop_above > 0: also stop (status 0) as soon as the Ritz value (a lower bound of lambda_max) reaches stop_above.
// the two halves of lanczos_lmax_f64: begin queues the first steps and the read-back and records an event, finish
// waits for that event only (kernels queued in between keep running) and decides / continues
struct LanczosRun {
    const double* G = nullptr;
    int64_t N = 0, ldG = 0;
    double rel_tol = 0.0, accept_below = 0.0, stop_above = 0.0;
    int max_steps = 0, cap = 0, launched = 0, chunk = 0;
    double *st = nullptr, *ab = nullptr;
    size_t lds = 0;
    bool event_pending = false, trivial = false, unsupported = false;
    bool mail_ok = false, use_mail = false;   // read-back through the handle's mailbox (polled) instead of copy + event
    double seq = 0.0;
    const double* v0 = nullptr;   // start vector (device, N; need not be normalised); nullptr: fixed pseudo-random vector
    // k_lanczos_multi (N <= 1024): a chunk of steps per launch, the matrix in the LDS of 64 workgroups
    bool multi = false;
    double* xch = nullptr;
    size_t lds_multi = 0;
    int rows_per = 0;
    unsigned int salt = 0;
};
int lanczos_begin(Handle* h, LanczosRun& r, const double* G, int64_t N, int64_t ldG, double rel_tol, int max_steps,
                  double accept_below, double stop_above, const double* v0 = nullptr);
int lanczos_finish(Handle* h, LanczosRun& r, double* lmax, int* steps_used);
// the same for an operator given only as a product w = Op(q) on N-vectors (device pointers, handle's stream)
using LzApply = std::function<int(const double* q, double* w)>;
int lanczos_lmax_op(Handle* h, int64_t N, const LzApply& apply, double rel_tol, int max_steps, double* lmax,
                    int* steps_used, double accept_below = 0.0, double stop_above = 0.0);
int lanczos_lmax_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double rel_tol, int max_steps,
                     double* lmax, int* steps_used, double accept_below = 0.0, double stop_above = 0.0,
                     const double* v0 = nullptr);
// v (N) = the column of largest diagonal entry of the symmetric S (N x N, ld N); S /= sqrt(sum of the nb partial sums)
int launch_dominant_column(Handle* h, const double* S, int64_t N, double* v);
int launch_scale_by_norm(Handle* h, double* S, int64_t N, const double* part, int nb);

// best of nsteps lower bounds ||G v|| of lambda_max(G) along power steps on a vector that persists between calls
// (init: restart it); returns 1 when not applicable (the caller runs Lanczos)
int power_lower_bound(Handle* h, const double* G, int64_t N, int64_t ldG, bool init, int nsteps, double* lb_out);

// ---------------- subspace.hip ----------------
int subspace_max_block(int64_t N);
int launch_cgs2(Handle* h, double* Y, int64_t N, int64_t p, double* status_dev);
int launch_project_out(Handle* h, const double* X, int64_t c, double* Yb, int64_t pb, double* W, int64_t N);
int launch_orth(Handle* h, double* Y, double* tmp, double* W, int64_t N, int64_t p, double* status_dev,
                bool allow_cholqr, bool* used_cholqr, bool one_pass = false, int64_t c_start = 0);   // c_start: the columns in
                // front of it are orthonormal already (Gram-Schmidt path only): they are projected out of the rest, not touched
// Rayleigh-Ritz of a nearly orthogonal warm block without orthonormalisation pass and Jacobi sweeps (subspace.hip,
// k_rr_small): B = Y'Y, Hg = Y'GY (scratch, p x p), C (p x p) with X' = Y C the Ritz vectors, lam their Ritz values,
// status[1] != 0: not applicable (the caller falls back to CholeskyQR2 + Jacobi).  p <= 32.
int launch_rr_small(Handle* h, const double* Y, const double* GY, double* Bm, double* Hm, double* Cout, double* lam,
                    double* status, int64_t N, int64_t p, int64_t nt, double tau2);
int launch_rr_small_only(Handle* h, const double* Bm, const double* Hm, double* Cout, double* lam, double* status, int64_t p,
                         int64_t nt, double tau2);
int launch_ritz_resid(Handle* h, const double* GX, const double* X, const double* theta, int64_t N, int64_t p,
                      double* res);
int launch_rayleigh(Handle* h, const double* GX, const double* X, int64_t N, int64_t p, double* theta);
template <typename TA>
int launch_skinny_mm(Handle* h, const TA* A, int64_t lda, const double* X, int64_t ldx, double* Y, int64_t ldy,
                     int64_t R, int64_t K, int64_t p);
// GD = scale * (G - Vs Vg')
int launch_deflate(Handle* h, const double* G, int64_t ldG, const double* Vs, const double* Vg, double* GD, int64_t N,
                   int64_t r, double scale = 1.0);
// ||S^2||_F^2 of a symmetric S (N x N, ld N) as per-tile partial sums in the host-visible mailbox ([16 .. 16 + ntile)),
// published with sequence number seq
int launch_sq_norm(Handle* h, const double* S, int64_t N, double* mailbox_dev, unsigned int* ticket, double seq, int* ntile_out);

// complex.hip: ComplexF64 sweeps + realification (panels are interleaved re/im, n counts complex elements)
int launch_cshrink(Handle* h, const double* D, const double* A, const double* Y, double* E, double* Z, int64_t n,
                   double inv_mu, double thr);
int launch_cupdate(Handle* h, const double* D, const double* A, const double* E, double* Y, double* R, int64_t n,
                   double mu);
int launch_cdiv(Handle* h, const double* D, double* Y, int64_t n, double s);
int launch_realify(Handle* h, const double* Z, int64_t M, int64_t N, double* W);
int launch_unrealify(Handle* h, const double* AR, int64_t M, int64_t N, double* A);
int launch_pack_complex(Handle* h, const double* T, int64_t M, int64_t d, double* U);
int launch_cmaxabs(Handle* h, const double* x, int64_t n, double* host_out);
// batched.hip: one workgroup per tiny rpca problem
size_t rpca_small_lds_bytes(int64_t M, int64_t N, bool* in_lds, size_t esz = 8);
template <typename T>
int launch_rpca_small(Handle* h, const T* D, int64_t M, int64_t N, int64_t batch, double lambda, double tol,
                      double rho, int64_t iters, int64_t maxrank, bool nonnegA, bool nonnegE, bool nukeA, T* A,
                      T* E, T* S, T* Vt, int64_t* sv, int32_t* it, int32_t* st, T* cost, T* scratch);
int launch_symm_skinny(Handle* h, const double* G, int64_t ldG, const double* X, double* Y, int64_t N, int64_t p);
int launch_panel_tn(Handle* h, const double* A, const double* B, double* H, int64_t N, int64_t p,
                    const double* skip_status = nullptr);
int launch_deflate_vec(Handle* h, const double* Vs, const double* Vg, int64_t r, const double* q, double* c, double* w,
                       int64_t N);
// mailbox_dev != nullptr (and p <= 256): theta, res and status[0..1] are also written to the host-visible mailbox
// and published with sequence number seq (see k_ritz_finish)
int launch_ritz_finish(Handle* h, const double* Q, const double* GQ, const double* S, double* X, double* GX,
                       double* theta, double* res, int64_t N, int64_t p, const double* status = nullptr,
                       double* mailbox_dev = nullptr, unsigned int* arrivals = nullptr, double seq = 0.0,
                       SpecCtrl* ctrl = nullptr, double inv_mu = 0.0, int nukeA = 1, const double* sort_keys = nullptr,
                       const double* sort_guard = nullptr);
int launch_panel_rot2(Handle* h, const double* Q, const double* GQ, const double* S, double* X1, double* X2,
                      int64_t N, int64_t p);
int launch_fill_hash(Handle* h, double* X, int64_t n, unsigned int seed);
int launch_fill_gauss(Handle* h, double* X, int64_t n, unsigned int seed);
int launch_colsumsq(Handle* h, const double* B, int64_t rows, int64_t ld, int64_t cols, double* out);

// ---------------- hankel.hip ----------------
template <typename T>
int launch_hankel(Handle* h, const T* x, int64_t Nx, int64_t Dch, int64_t ldx, int64_t L,
                  int64_t lag, T* X, int64_t ldX);
template <typename T>
int launch_unhankel(Handle* h, const T* A, int64_t K, int64_t L, int64_t Dch, int64_t ldA,
                    int64_t lag, int64_t Nx, T* y, int64_t ldy);
// multi-GPU lowrankfilter: per-shard anti-diagonal sums / counts at a window offset, and the division after the all-reduce
template <typename T>
int launch_unhankel_partial(Handle* h, const T* A, int64_t K, int64_t L, int64_t Dch, int64_t ldA, int64_t lag,
                            int64_t Nw, int64_t off, double* sum, double* cnt, int64_t ldy);
template <typename T>
int launch_unhankel_finish(Handle* h, const double* sum, const double* cnt, int64_t n, T* y);
// y = unhankel(Tm Vs') for one channel and lag 1 without the panel: the same sums in the same order as launch_rebuild_store
// followed by launch_unhankel (bit-identical), reading only the factors (r <= 32)
template <typename T>
int launch_unhankel_factors(Handle* h, const double* Tm, int64_t ldT, const double* Vs, int64_t ldV, int64_t r, int64_t K,
                            int64_t L, int64_t Nx, T* y);
template <typename T>
int launch_soft_hankel(Handle* h, T* A, int64_t K, int64_t L, int64_t ldA, T eps, T* mean_ws);
template <typename T>
int launch_soft_toward(Handle* h, T* A, int64_t K, int64_t L, int64_t ldA, const T* m, T eps);

}  // namespace tlsq
