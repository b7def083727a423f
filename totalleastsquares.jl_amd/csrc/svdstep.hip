// The SVD step of the ALM loop on the small side (src/robustPCA.jl:193-204 - what gesdd does in the reference): the warm-started,
// certified subspace iteration on the Gram matrix / the implicit operator, its count certificate (matrix powers, Lanczos), and
// the block carried from one iteration to the next.  The kernels are in subspace.hip, jacobi.hip, lanczos.hip, matfun.hip;
// the ALM loop that calls this is rpca_core in solver.hip.  Split from solver.hip in round 5.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <thread>

#include "svdstep.hpp"

namespace tlsq {


// Y (N x p, ld N) = G X
static int op_apply(Handle* h, const GramOp& op, int64_t N, const double* X, double* Y, int64_t p) {
    if (p <= 0) return TLSQ_OK;
    if (!op.implicit()) return launch_symm_skinny(h, op.G, N, X, Y, N, p);
    // fp32 panels, blocks of more than 8 columns: both halves on the fp32 MFMA (gemm.hip, op_gram_f32) - the widening kernels
    // below run on the fp64 MFMA at half the rate (narrow blocks, the Lanczos vectors, are bandwidth-bound either way)
    if (op.z_f32 && op.lowp_ok && p > 8 && !dev_is(DEV_NO_F32_SKINNY, '1')) {
        for (int64_t c0 = 0; c0 < p; c0 += 96) {
            const int64_t pc = std::min<int64_t>(96, p - c0);
            TLSQ_TRY(op_gram_f32(h, (const float*)op.Z, op.ldZ, op.M, N, X + (size_t)c0 * N, N, Y + (size_t)c0 * N, N, pc));
        }
        TLSQ_TRY(comm_allreduce(h, Y, (size_t)N * p, ncclSum));
        return TLSQ_OK;
    }
    void* Tv;
    TLSQ_TRY(ws_get(h, WS_OPT, (size_t)op.M * std::min<int64_t>(p, 96) * 8, &Tv));
    for (int64_t c0 = 0; c0 < p; c0 += 96) {
        const int64_t pc = std::min<int64_t>(96, p - c0);
        TLSQ_TRY(tsmm_mixed(h, op.Z, op.z_f32, op.ldZ, X + (size_t)c0 * N, N, (double*)Tv, op.M, op.M, N, pc));
        TLSQ_TRY(ztmm_mixed(h, op.Z, op.z_f32, op.ldZ, (const double*)Tv, op.M, Y + (size_t)c0 * N, N, op.M, N, pc));
    }
    TLSQ_TRY(comm_allreduce(h, Y, (size_t)N * p, ncclSum));
    return TLSQ_OK;
}

// sigma_max of the panel behind an implicit operator (Lanczos on Z'Z through products); the same stopping rules
// as sigma_max_of_gram
int sigma_max_of_op(Handle* h, const GramOp& op, int64_t N, double rel_tol, double* out, double stop_above_sigma) {
    double lmax = 0.0;
    int steps = 0;
    const LzApply apply = [&](const double* q, double* w) -> int { return op_apply(h, op, N, q, w, 1); };
    const int st = lanczos_lmax_op(h, N, apply, rel_tol, 1000, &lmax, &steps, 0.0, stop_above_sigma * stop_above_sigma);
    if (st < 0) return st;
    *out = std::sqrt(lmax);   // (no dense fallback in large mode: the value after 1000 steps stands)
    return TLSQ_OK;
}



// ---- count certificate: lambda_max(GD) < margin for the deflated, scaled Gram matrix GD ---------------------------
// Matrix powers first: lambda_max(S) <= ||S^(2^k)||_F^(1/2^k) for symmetric S - rigorous, deterministic upper bounds
// from plain MFMA contractions (a Lanczos Ritz value is only a LOWER bound of lambda_max).  ||GD^2||_F comes from one
// fused kernel whose per-tile sums land in the host-visible mailbox (k_sq_norm in subspace.hip); ||GD^4||_F and, when
// that is still too coarse, a Lanczos run with the usual 1.5x safety factor follow synchronously - rarely.
// ||S^(2^levels)||_F^2 of the symmetric S = GD through the general MFMA GEMM (slab reduction with the norm by-product),
// partial sums read back and added on the host in order: the slow but general form (no mailbox, second squaring)
static int power_norm_sync(Handle* h, SubspaceState& st, int levels, double* out, int first_level = 0) {
    // out[l - first_level] = ||S^(2^l)||_F^2 for l = max(first_level, 1) .. levels (first_level 0: only the last one)
    const int64_t N = st.cert_N;
    void *P[2], *part;
    const size_t pslots = (size_t)std::max<int64_t>(4096, ((N + 31) / 32) * ((N + 31) / 32 + 1) / 2);
    TLSQ_TRY(ws_get(h, WS_CP1, (size_t)N * N * 8, &P[0]));
    if (levels >= 2) TLSQ_TRY(ws_get(h, WS_CP2, (size_t)N * N * 8, &P[1]));
    // N a multiple of 128 (round 4): squarings through k_small_mm_blk (8 us instead of ~30 through the split-K product), the norm
    // of S^(2^l) from k_sq_norm_blk applied to S^(2^(l-1)) - tile sums to the certificate's mailbox region, added in tile order
    {
        const int64_t ntl = (N + 31) / 32;
        const size_t need = (size_t)(kCertMailboxOffset + 16 + ntl * (ntl + 1) / 2) * 8;
        if (sq_norm_blk_ok(N) && h->mailbox && need <= h->mailbox_bytes && !dev_is(DEV_NO_MAILBOX, '1')) {
            void* scal;
            TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
            unsigned int* ticket = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(scal) + 336);   // (the one of power_cert_begin)
            if (!h->cert_ticket_ready) {
                TLSQ_HIP(h, hipMemsetAsync(ticket, 0, 4, h->stream));
                h->cert_ticket_ready = true;
            }
            const int l0 = first_level > 0 ? first_level : levels;
            const double* cur = st.cert_GD;
            volatile double* mb = h->mailbox + kCertMailboxOffset;
            bool mail_ok = true;
            for (int l = 1; l <= levels && mail_ok; ++l) {
                if (l >= l0) {
                    const double seq = (h->mail_seq += 1.0);
                    int ntile = 0;
                    TLSQ_TRY(launch_sq_norm(h, cur, N, h->mailbox_dev + kCertMailboxOffset, ticket, seq, &ntile));
                    const double t_poll = now_ms();
                    while (mb[0] != seq && now_ms() - t_poll < 2000.0) {
                    }
                    if (mb[0] != seq) {
                        mail_ok = false;
                        break;
                    }
                    double a = 0.0;
                    for (int t = 0; t < ntile; ++t) a += mb[16 + t];
                    out[l - l0] = a;
                }
                if (l < levels) {
                    double* dst = (double*)P[(l - 1) & 1];
                    bool sq_ok = false;
                    TLSQ_TRY(matfun_square(h, cur, dst, N, &sq_ok));
                    if (!sq_ok) {
                        mail_ok = false;
                        break;
                    }
                    cur = dst;
                }
            }
            if (mail_ok) return TLSQ_OK;
            h->mailbox_bytes = 0;   // (never seen in practice; classic read-backs from now on)
        }
    }
    TLSQ_TRY(ws_get(h, WS_CPART, pslots * 8 * (size_t)std::max(1, levels), &part));
    int nb = 0;
    const double* src = st.cert_GD;
    for (int l = 1; l <= levels; ++l) {
        double* dst = (double*)P[(l - 1) & 1];
        TLSQ_TRY(gemm_mixed(h, true, true, src, 0, N, src, 0, N, dst, 0, N, N, N, N, true, nullptr,
                            (double*)part + (size_t)(l - 1) * pslots, &nb));
        src = dst;
    }
    std::vector<double> hp(pslots * (size_t)levels);
    TLSQ_HIP(h, hipMemcpyAsync(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    const int l0 = first_level > 0 ? first_level : levels;
    for (int l = l0; l <= levels; ++l) {
        double a = 0.0;
        for (int i = 0; i < nb; ++i) a += hp[(size_t)(l - 1) * pslots + (size_t)i];
        out[l - l0] = a;
    }
    return TLSQ_OK;
}

int power_cert_begin(Handle* h, SubspaceState& st) {
    const bool no_mailbox = dev_is(DEV_NO_MAILBOX, '1');
    const int64_t N = st.cert_N;
    const int64_t nt = (N + 31) / 32;
    st.cert_seq = 0.0;
    st.cert_mb = nullptr;
    if (!(h->mailbox && !no_mailbox && (size_t)(kCertMailboxOffset + 16 + nt * (nt + 1) / 2) * 8 <= h->mailbox_bytes)) return TLSQ_OK;
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    unsigned int* ticket = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(scal) + 336);   // self-resetting
    if (!h->cert_ticket_ready) {   // (fresh workspace memory is not zero)
        TLSQ_HIP(h, hipMemsetAsync(ticket, 0, 4, h->stream));
        h->cert_ticket_ready = true;
    }
    st.cert_seq = (h->mail_seq += 1.0);
    const size_t mb_off = st.cert_async ? kCertMailboxOffset : 0;
    st.cert_mb = h->mailbox + mb_off;
    TLSQ_TRY(launch_sq_norm(h, st.cert_GD, N, h->mailbox_dev + mb_off, ticket, st.cert_seq, &st.cert_ntile));
    return TLSQ_OK;
}

int cert_finish(Handle* h, SubspaceState& st, bool* pass) {
    *pass = false;
    double lmax = 0.0;
    int steps = 0;
    if (!st.cert_power) {
        const int lst = lanczos_finish(h, st.cert, &lmax, &steps);
        if (lst < 0) return lst;
        ++st.n_lanczos_cert;
        *pass = lmax * 1.5 < st.cert_margin;
        return TLSQ_OK;
    }
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    double a = -1.0;
    if (st.cert_seq != 0.0) {
        volatile double* mb = st.cert_mb ? st.cert_mb : h->mailbox;
        const double t_poll = now_ms();
        while (mb[0] != st.cert_seq && now_ms() - t_poll < 2000.0) {
        }
        if (mb[0] == st.cert_seq) {
            a = 0.0;
            for (int t = 0; t < st.cert_ntile; ++t) a += mb[16 + t];   // tile order: reproducible
        } else {
            h->mailbox_bytes = 0;   // never seen in practice; classic read-backs from now on
        }
    }
    if (a < 0.0) TLSQ_TRY(power_norm_sync(h, st, 1, &a));
    const double b1 = std::isfinite(a) ? std::pow(a, 0.25) : std::numeric_limits<double>::infinity();
    if (dbg) fprintf(stderr, "  power certificate: bound1=%.4f margin=%.6f\n", b1, st.cert_margin);
    if (b1 < st.cert_margin) {
        ++st.n_power;
        *pass = true;
        return TLSQ_OK;
    }
    if (!std::isfinite(a)) return TLSQ_OK;   // NaN / inf in the deflated matrix: not certified
    // one more squaring tightens the bound from rank^(1/4) to rank^(1/8) above lambda_max
    double b = 0.0;
    TLSQ_TRY(power_norm_sync(h, st, 2, &b));
    const double b2 = std::isfinite(b) ? std::pow(b, 0.125) : std::numeric_limits<double>::infinity();
    if (dbg) fprintf(stderr, "  power certificate: bound2=%.4f\n", b2);
    if (b2 < st.cert_margin) {
        ++st.n_power_l2;
        *pass = true;
        return TLSQ_OK;
    }
    // Still too coarse: the tail is flat and close to the mark (late iterations of a Hankel filter: hundreds of values
    // at 0.7x the threshold).  Three more squarings bring the bound to rank^(1/64) above lambda_max, still rigorous;
    // a Lanczos run (a LOWER bound, hence the 1.5x safety factor) screens first where an N^3 product is not small.
    const bool no_deep = dev_is(DEV_NO_DEEP_POWERS, '1');
    bool lanczos_done = false;
    if (st.cert_N > 1024 || no_deep) {
        const int lst = lanczos_lmax_f64(h, st.cert_GD, st.cert_N, st.cert_N, 0.02, 48, &lmax, &steps, st.cert_margin);
        if (lst < 0) return lst;
        ++st.n_lanczos_cert;
        lanczos_done = true;
        if (lmax * 1.5 < st.cert_margin) {
            *pass = true;
            st.cert_tail = 0.0;
            return TLSQ_OK;
        }
        st.cert_tail = lmax;
        if (lmax >= st.cert_margin || no_deep) return TLSQ_OK;   // an eigenvalue above the mark: the count is wrong
    }
    double c[3] = {0.0, 0.0, 0.0};
    TLSQ_TRY(power_norm_sync(h, st, 5, c, 3));
    const double inf = std::numeric_limits<double>::infinity();
    const double b3 = std::isfinite(c[0]) ? std::pow(c[0], 1.0 / 16.0) : inf;
    const double b4 = std::isfinite(c[1]) ? std::pow(c[1], 1.0 / 32.0) : inf;
    const double b5 = std::isfinite(c[2]) ? std::pow(c[2], 1.0 / 64.0) : inf;
    if (dbg) fprintf(stderr, "  power certificate: bound3=%.4f bound4=%.4f bound5=%.4f\n", b3, b4, b5);
    if (std::min(b3, std::min(b4, b5)) < st.cert_margin) {
        ++st.n_power_l2;
        *pass = true;
        st.cert_tail = 0.0;
        return TLSQ_OK;
    }
    if (lanczos_done) return TLSQ_OK;
    const int lst = lanczos_lmax_f64(h, st.cert_GD, st.cert_N, st.cert_N, 0.02, 48, &lmax, &steps, st.cert_margin);
    if (lst < 0) return lst;
    ++st.n_lanczos_cert;
    *pass = lmax * 1.5 < st.cert_margin;
    st.cert_tail = *pass ? 0.0 : lmax;
    return TLSQ_OK;
}

// one stream-ordered upload of an index list and a weight list of the same length r into `aux`
// (layout: int32 sel[r], padding to 8 bytes, double w[r]); returns the two device pointers
static int upload_sel_weights(Handle* h, void* aux, const std::vector<int32_t>& sel, const std::vector<double>& w,
                              int32_t** dsel, double** dw) {
    const size_t r = sel.size();
    const size_t off = ((r * 4 + 7) / 8) * 8;
    std::vector<char> buf(off + r * 8);
    memcpy(buf.data(), sel.data(), r * 4);
    memcpy(buf.data() + off, w.data(), r * 8);
    TLSQ_TRY(upload_async(h, aux, buf.data(), buf.size()));
    *dsel = (int32_t*)aux;
    *dw = (double*)((char*)aux + off);
    return TLSQ_OK;
}

// Vg = V[:, sel] * diag(w), Vs = V[:, sel] for host-side sel / w: as kernel arguments when short, through `aux` otherwise
int gather_scale_host(Handle* h, const double* V, int64_t N, const std::vector<int32_t>& sel,
                             const std::vector<double>& w, void* aux, double* Vg, double* Vs) {
    const int64_t r = (int64_t)sel.size();
    if (r <= 32) {
        SelWeights sw;
        for (int64_t i = 0; i < 32; ++i) {
            sw.sel[i] = i < r ? sel[i] : 0;
            sw.w[i] = i < r ? w[i] : 0.0;
        }
        return launch_gather_scale_arg(h, V, N, sw, r, Vg, Vs);
    }
    int32_t* dsel;
    double* dw;
    TLSQ_TRY(upload_sel_weights(h, aux, sel, w, &dsel, &dw));
    return launch_gather_scale(h, V, N, dsel, dw, r, Vg, Vs);
}

// upload a column selection and gather X = V[:, sel]
int gather_cols(Handle* h, const double* V, int64_t N, const std::vector<int32_t>& sel, double* X) {
    const int64_t r = (int64_t)sel.size();
    if (r == 0) return TLSQ_OK;
    void* aux;
    TLSQ_TRY(ws_get(h, WS_AUX0, (size_t)r * 16 + 64, &aux));
    TLSQ_TRY(upload_async(h, aux, sel.data(), (size_t)r * 4));
    TLSQ_TRY(launch_gather_scale(h, V, N, (const int32_t*)aux, nullptr, r, nullptr, X));
    return TLSQ_OK;
}

// Try to get the sigma_i >= inv_mu pairs of G from the block carried in st.  *ok = false -> caller must run
// the full solver.  On success V_out (N x p) / s describe the Ritz pairs (all p of them; the wanted ones are
// converged, the rest only bound the count).
int svd_subspace(Handle* h, const GramOp& op, int64_t N, double inv_mu, SubspaceState& st,
                        double** V_out, SmallSvd& s, int64_t* sweeps, bool* ok) {
    *ok = false;
    st.fail = SubspaceState::FAIL_NONE;
    st.cert_pending = false;
    st.cert_tail = 0.0;
    const bool hook = st.hook_rank > 0;
    const bool cold = hook || !st.valid;
    if (hook) {
        st.p = std::min<int64_t>(std::min<int64_t>(st.hook_rank + 10, subspace_max_block(N)), N);
        if (st.p < st.hook_rank || st.p < 3) return TLSQ_OK;   // block too large for this path: full solver + truncation
    } else if (cold) {
        // no block yet (first ALM iteration): start from a pseudo-random block of 10 + 8 columns — 10 is the
        // reference's initial rank guess `sv = 10` (src/robustPCA.jl:184)
        if (!st.allow_cold) return TLSQ_OK;
        st.p = std::min<int64_t>(std::min<int64_t>(std::max<int64_t>(18, st.cold_p), subspace_max_block(N)), N);
        if (st.p < 3) return TLSQ_OK;
    }
    if (st.p < 3) return TLSQ_OK;
    st.fail = SubspaceState::FAIL_NONE;
    const int64_t p = st.p;
    void *X, *Q, *GQ, *XN, *GX, *H, *S, *HB, *lam, *aux, *GD;
    {   // the block lives in WS_SX across calls and may grow: reserve the largest block once (ws_get reallocates)
        const int64_t pcap = std::max<int64_t>(p, std::min<int64_t>(subspace_max_block(N), N));
        TLSQ_TRY(ws_get(h, WS_SX, (size_t)N * pcap * 8, &X));
        TLSQ_TRY(ws_get(h, WS_SXN, (size_t)N * pcap * 8, &XN));
    }
    if (cold) TLSQ_TRY(launch_fill_hash(h, (double*)X, N * p, hook ? (unsigned int)(st.hook_seed * 2654435761ull + 77u) : 0x9E3779B9u));
    TLSQ_TRY(ws_get(h, WS_SQ, (size_t)N * p * 8, &Q));
    TLSQ_TRY(ws_get(h, WS_SGQ, (size_t)N * p * 8, &GQ));
    TLSQ_TRY(ws_get(h, WS_SXN, (size_t)N * p * 8, &XN));
    TLSQ_TRY(ws_get(h, WS_SGX, (size_t)N * p * 8, &GX));
    TLSQ_TRY(ws_get(h, WS_SH, (size_t)p * p * 8, &H));
    TLSQ_TRY(ws_get(h, WS_SS, (size_t)p * p * 8, &S));
    TLSQ_TRY(ws_get(h, WS_SHB, (size_t)p * p * 8, &HB));
    TLSQ_TRY(ws_get(h, WS_LAM, (size_t)std::max<int64_t>(N, 3 * p + 32) * 8, &lam));
    TLSQ_TRY(ws_get(h, WS_AUX0, (size_t)p * 16 + 64, &aux));
    double* theta_dev = (double*)lam;
    double* res_dev = theta_dev + p;
    double* stat_dev = res_dev + p;
    double* lamH_dev = stat_dev + 8;
    std::vector<double> host((size_t)2 * p + 8);
    if (hook && !dev_is(DEV_HOOK_CLASSIC, '1')) {
        // The randomized hook as a block power method (round 5).  The reference's hook is `svd(Z, sv)` with a user function of
        // the rsvd kind (src/robustPCA.jl:195-197; test/runtests.jl:388-398 uses rank sv, two power iterations): Q = orth((Z Z')^q Z Omega),
        // then the SVD of Q'Z.  Here on the small side:  X <- orth(G X) `npow` times (2 on the panel operator, 3 on an explicit Gram matrix), then ONE Rayleigh-Ritz step
        // H = Q'(G Q): three products with the panel pair instead of the five of the two-step form below, one p x p eigenproblem
        // instead of two, and CholeskyQR2 (a handful of multi-workgroup launches) for every orthonormalisation: a block is
        // orthonormalised after EACH product, so its condition number is sigma_1^2 / sigma_p^2 of the panel, not the fourth
        // power a random block has after two products (which needed the column-sequential CGS2: six 0.26 ms one-workgroup
        // launches per orthonormalisation at 4096 x 74).  The start block is the previous iteration's sorted Ritz block when
        // the caller says it is still in WS_SX (hook_carry) - Z_k changes little from one ALM iteration to the next, so the
        // block enters already near-invariant - with the last few pad columns refreshed from the hash generator, so that a
        // direction absent from the carried block can still enter; a cold call (iteration 2 after a full decomposition on
        // another route, HOOK_COLD=1) starts from a random block like the reference's Omega.
        // A Cholesky factorisation that breaks down (status[1]) repeats the call from a random block with CGS2.
        // (products before the Rayleigh-Ritz step: two on the panel operator - three products in all, the work of rsvd with two power
        //  iterations; three on an explicit Gram matrix, where a product is a 12 us kernel and the extra half power that rsvd's
        //  Q'Z step has over a Rayleigh-Ritz step on G^2 X is cheaper to exceed than to argue about: tools/fuzz_misc.py compares
        //  with the oracle's rsvd to 1e-5)
        const int npow_default = op.implicit() ? 2 : 3;
        const int npow = [npow_default] {
            const char* e = dev_get(DEV_HOOK_POWER);
            const int v = e ? atoi(e) : npow_default;
            return v >= 1 && v <= 6 ? v : npow_default;
        }();
        const unsigned int seed32 = (unsigned int)(st.hook_seed * 2654435761ull + 77u);
        int64_t carry = dev_is(DEV_HOOK_COLD, '1') ? 0 : std::min<int64_t>(st.hook_carry, p);
        // (the last pad columns come from the hash generator at every call, so that a direction absent from the carried block can
        //  still enter; HOOK_PAD_REFRESH=0 carries all of them - measured: the same seven Jacobi sweeps either way, the Ritz vectors
        //  inside the dominant cluster rotate by O(1) from one ALM iteration to the next)
        if (carry > 0 && !dev_is(DEV_HOOK_PAD_REFRESH, '0'))
            carry = std::max<int64_t>(0, std::min<int64_t>(carry, p - std::max<int64_t>(2, (p - st.hook_rank) / 2)));
        st.hook_carry = 0;
        double* stat2 = lamH_dev + p;          // 3 status words per intermediate orthonormalisation (npow <= 6)
        std::vector<double> host2((size_t)3 * 6);
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool chol = attempt == 0 && !dev_is(DEV_HOOK_CGS2, '1');
            if (attempt > 0) carry = 0;
            if (carry < p) TLSQ_TRY(launch_fill_hash(h, (double*)X + (size_t)carry * N, N * (p - carry), seed32 + (unsigned int)attempt));
            double* cur = (double*)Q;
            double* other = (double*)XN;
            const double* src = (const double*)X;
            bool used_any = false;
            for (int t = 0; t < npow; ++t) {
                ++st.steps;
                TLSQ_TRY(op_apply(h, op, N, src, cur, p));
                bool used = false;
                // (a carried block consists of Ritz vectors: their images are nearly orthogonal, and the column scaling of the
                //  second product does not hurt the scaled Cholesky factor - only the last product's block is orthonormalised;
                //  the refreshed pad columns are projected against the blocks in front of them first, HOOK_ORTH_ALL=1 for every product)
                const bool skip_orth = chol && carry >= st.hook_rank && t + 1 < npow && !dev_is(DEV_HOOK_ORTH_ALL, '1');
                if (skip_orth) {
                    TLSQ_HIP(h, hipMemsetAsync(stat2 + 3 * t, 0, 24, h->stream));
                } else {
                    TLSQ_TRY(launch_orth(h, cur, (double*)GQ, (double*)H, N, p, t + 1 < npow ? stat2 + 3 * t : stat_dev, chol, &used, false));
                }
                used_any = used_any || used;
                src = cur;
                std::swap(cur, other);
            }
            const double* Qf = src;            // the orthonormal block
            TLSQ_TRY(op_apply(h, op, N, Qf, (double*)GQ, p));
            TLSQ_TRY(launch_panel_tn(h, Qf, (const double*)GQ, (double*)H, N, p));
            int64_t sw = 0;
            // (sketch products carry the fp32 rounding of the block, ~1e-7: Ritz vectors orthogonal to 1e-9 lose nothing, and the
            //  Jacobi sweeps from there to 4e-15 - three of eight at p = 74, 0.14 ms each - are saved)
            TLSQ_TRY(symeig_f64(h, (const double*)H, p, p, (double*)HB, (double*)S, true, lamH_dev, &sw, true, false, true,
                                op.implicit() && op.lowp_ok ? 1e-9 : 0.0));
            if (sweeps) *sweeps += sw;
            TLSQ_TRY(launch_ritz_finish(h, Qf, (const double*)GQ, (const double*)S, (double*)X, (double*)GX, theta_dev, res_dev, N, p));
            TLSQ_HIP(h, hipMemcpyAsync(host.data(), theta_dev, (size_t)(2 * p + 3) * 8, hipMemcpyDeviceToHost, h->stream));
            if (npow > 1) TLSQ_HIP(h, hipMemcpyAsync(host2.data(), stat2, (size_t)3 * (npow - 1) * 8, hipMemcpyDeviceToHost, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            bool failed = false;
            if (used_any) {
                failed = host[(size_t)(2 * p + 1)] != 0.0;
                for (int t = 0; t + 1 < npow; ++t) failed = failed || host2[(size_t)(3 * t + 1)] != 0.0;
            }
            bool finite = true;
            for (int64_t i = 0; i < 2 * p; ++i) finite = finite && std::isfinite(host[(size_t)i]);
            if (dev_get(DEV_DEBUG) != nullptr) {
                int sd = -1;
                (void)hipMemcpy(&sd, (char*)h->ws[WS_SCAL].p + 136, 4, hipMemcpyDeviceToHost);
                fprintf(stderr, "  hook: p %lld carry %lld npow %d %s, %d Jacobi sweeps%s\n", (long long)p, (long long)carry, npow,
                        chol ? "CholeskyQR2" : "CGS2", sd, failed ? " - factorisation broke down, again from a random block with CGS2" : "");
            }
            if ((failed || !finite) && attempt == 0) continue;
            if (!finite) {
                st.fail = SubspaceState::FAIL_NUMERIC;
                return TLSQ_OK;
            }
            break;
        }
        s.sigma.resize((size_t)p);
        for (int64_t i = 0; i < p; ++i) s.sigma[(size_t)i] = std::sqrt(std::max(host[(size_t)i], 0.0));
        s.ncols = p;
        sort_desc(s);
        st.hook_zq = nullptr;
        if (op.implicit() && op.z_f32 && op.lowp_ok && p > 8 && p <= 80 && !dev_is(DEV_NO_F32_SKINNY, '1') &&
            !dev_is(DEV_OPGRAM_OLD, '1') && op_gram_f32_fast_ok((const float*)op.Z, op.ldZ, op.M, N, p) && op.ldZ == op.M) {
            st.hook_zq = (const float*)h->ws[WS_OPT].p;   // (the T32 of the last op_apply: Z Qf, 16 ceil(p / 16) columns)
            st.hook_zq_lw = 16 * (int)((p + 15) / 16);
            st.hook_S = (const double*)S;
            st.hook_order = s.order;
        }
        bool sorted = true;
        for (int64_t i = 0; i < p; ++i) sorted = sorted && s.order[(size_t)i] == (int32_t)i;
        if (!sorted) {
            std::vector<double> sg((size_t)p);
            for (int64_t i = 0; i < p; ++i) sg[(size_t)i] = s.sigma[(size_t)s.order[(size_t)i]];
            TLSQ_HIP(h, hipMemcpyAsync(XN, X, (size_t)N * p * 8, hipMemcpyDeviceToDevice, h->stream));
            TLSQ_TRY(upload_async(h, aux, s.order.data(), (size_t)p * 4));
            TLSQ_TRY(launch_gather_scale(h, (const double*)XN, N, (const int32_t*)aux, nullptr, p, nullptr, (double*)X));
            s.sigma = sg;
            std::iota(s.order.begin(), s.order.end(), 0);
        }
        s.ncols = std::min<int64_t>(st.hook_rank, p);   // rank-sv truncation, like `svd(Z, sv)`
        st.hook_carry = p;
        *V_out = (double*)X;
        *ok = true;
        return TLSQ_OK;
    }
    const int max_steps = hook ? 2 : (cold ? 30 : 10) + st.extra_steps;
    const int64_t ntop = cold ? p : std::min<int64_t>(st.ntop, p);
    int64_t svp = 0;
    bool conv = false;
    double prev_maxres = 0.0;
    bool gx_valid = false;    // WS_SGX holds G X for the block in WS_SX (see the top of the loop)
    bool x_settled = false;   // nothing that writes the block X has been queued since the host read the last step's results
    bool force_cgs2 = cold;   // a random block is far too ill-conditioned for CholeskyQR2
    bool cgs2_sticky = false;
    const bool no_onepass = dev_is(DEV_NO_ONEPASS, '1');
    const bool no_rr_fast = dev_is(DEV_NO_RR_FAST, '1');
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    // (declined: for the rest of this call the block is not what k_rr_small is made for.  A kernel that keeps declining -
    //  near-degenerate pairs of Ritz values, as a Hankel filter has them - is not tried in the next 2, 4, 8, 16 calls.)
    bool rr_fast_declined = false;
    if (st.rr_skip > 0) {
        --st.rr_skip;
        rr_fast_declined = true;
    }
    for (int step = 0; step < max_steps; ++step) {
        ++st.steps;
        st.spec.launched = false;
        // Q = orth([G^q X_top, G X_pad]): the block is kept sorted, its first `nt` columns are the dominant
        // vectors; q-1 extra multiplications of those columns cost one skinny GEMM each and raise their
        // convergence factor to the q-th power (a whole step costs ~15 GEMMs).  The pad columns get a single
        // multiplication so that they keep tracking the top of the tail spectrum.  Cold (random) start: every
        // column, q = 2 (higher powers would make the random block too ill-conditioned for CGS2).
        // (second and later steps: G X is already there - the Rayleigh-Ritz finish of the previous step formed it with the Ritz
        //  vectors, GX = (G Q) S - as long as the block has not been re-ordered since: one product less per step, of six in
        //  a step of the randomized hook, 1.4 ms each at 65536 x 4096)
        if (gx_valid) TLSQ_HIP(h, hipMemcpyAsync(Q, GX, (size_t)N * p * 8, hipMemcpyDeviceToDevice, h->stream));
        else TLSQ_TRY(op_apply(h, op, N, (const double*)X, (double*)Q, p));
        gx_valid = false;
        // (cold, round 6: after the first step the block is sorted and its count known - only the counted columns take the extra
        //  multiplications from then on, and more of them: every column to the power 4 is what the pad columns' condition number
        //  allows, (lambda_pad / lambda_1)^4 ~ 1e-7 at C2; the counted columns alone take 7 and the second step lands under the
        //  residual bound that used to need a third - 100 us of the first ALM iteration.  COLD_TOP=0: as before)
        const bool cold_top = cold && !hook && step >= 1 && svp >= 1 && svp <= p - 2 && !dev_is(DEV_COLD_TOP, '0');
        const int64_t nt_step = cold_top ? svp : cold ? p : std::min<int64_t>(step == 0 ? ntop : svp, p);   // leading columns treated as wanted
        {
            const int64_t nt = nt_step;
            // (cold: 2 on the random block; from the second step on the block consists of Ritz vectors and takes a higher power
            //  as well as any warm block - one step less to the residual bound, TLSQ_COLD_Q)
            const int cold_q = [] { const char* e = dev_get(DEV_COLD_Q); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 7 ? v : 4; }();
            const int q = cold ? ((step == 0 || force_cgs2) ? 2 : cold_top ? std::max(cold_q, 7) : cold_q) : st.q_warm;   // adapted below: a multiplication of the top columns costs ~12 us, a step ~200
            // Cold start of a wide block (large mode: 76 columns of length 4096 at BASELINE config 5): the random block is
            // orthonormalised after its FIRST product as well (round 5) - G X has the condition number of G on the block, its second
            // product the square of it, which CholeskyQR2 cannot take and the column-sequential CGS2 needs 0.26 ms per 16 columns
            // for (2.6 ms per cold step at p = 76, four cold rounds while the block grows to the rank).  With the intermediate
            // pass both orthonormalisations are CholeskyQR2 (~0.3 ms each); a breakdown of the final one repeats the step with
            // CGS2 as before (a breakdown of the intermediate one leaves G X as it is: the final one then decides).
            const bool cold_chol = cold && step == 0 && force_cgs2 && !cgs2_sticky && q == 2 && nt == p && (N > 2048 || p > 32) &&
                                   !dev_is(DEV_COLD_CGS2, '1') && !dev_is(DEV_NO_CHOLQR, '1') &&
                                   // (the status-guarded finish of the mailbox path leaves X alone when the factorisation breaks down)
                                   h->mailbox && !dev_is(DEV_NO_MAILBOX, '1') && p <= 512 && (size_t)(2 * p + 10) * 8 <= h->mailbox_bytes;
            if (cold_chol) {
                bool used0 = false;
                TLSQ_TRY(launch_orth(h, (double*)Q, (double*)GQ, (double*)H, N, p, lamH_dev + p, true, &used0, false));
                force_cgs2 = false;
            }
            // the extra multiplications ping-pong between Q and GQ; an odd count ends in GQ and is copied back
            bool in_q = true;
            for (int t = 1; t < q && nt > 0; ++t) {
                TLSQ_TRY(op_apply(h, op, N, (const double*)(in_q ? Q : GQ), (double*)(in_q ? GQ : Q), nt));
                in_q = !in_q;
            }
            if (!in_q) TLSQ_HIP(h, hipMemcpyAsync(Q, GQ, (size_t)N * nt * 8, hipMemcpyDeviceToDevice, h->stream));
        }
        dbg_hash(h, "sub.chain", Q, (size_t)N * p * 8);
        bool used_cholqr = false;
        // Warm single-block panels: the columns of Q = [G^q X_top, G X_pad] are images of Ritz vectors - nearly orthogonal, and
        // Q'GQ nearly diagonal once they are normalised.  Orthonormalisation, Rayleigh quotient and its eigenvectors then
        // come from two reductions over the panel and ONE workgroup of p x p products (subspace.hip, k_rr_small) instead of
        // CholeskyQR2, H = Q'GQ and the Jacobi solver; anything that kernel declines (panel too far from orthogonal, a
        // cluster of Ritz values with internal coupling) repeats the step on the classic path.
        const bool rr_fast = !cold && !force_cgs2 && !rr_fast_declined && !no_rr_fast && p <= 32 && !op.implicit();
        // one CholeskyQR pass when the previous step on this block cleared the one-pass pivot bound with room to spare
        const bool one_pass = !cold && !force_cgs2 && !no_onepass && p <= 32 && st.chol_p == p && st.chol_piv >= 0.5;   // (32 = CQ_PMAX: single-block panels)
        if (rr_fast) {
            TLSQ_TRY(op_apply(h, op, N, (const double*)Q, (double*)GQ, p));
            TLSQ_TRY(launch_rr_small(h, (const double*)Q, (const double*)GQ, (double*)H, (double*)HB, (double*)S, lamH_dev, stat_dev, N, p,
                                     nt_step, inv_mu * inv_mu));
            dbg_hash(h, "sub.GQ", GQ, (size_t)N * p * 8);
            dbg_hash(h, "sub.S", S, (size_t)p * p * 8);
        } else {
        TLSQ_TRY(launch_orth(h, (double*)Q, (double*)GQ, (double*)H, N, p, stat_dev, !force_cgs2, &used_cholqr, one_pass));
        dbg_hash(h, "sub.orth", Q, (size_t)N * p * 8);
        // Rayleigh-Ritz: H = Q' (G Q)
        TLSQ_TRY(op_apply(h, op, N, (const double*)Q, (double*)GQ, p));
        TLSQ_TRY(launch_panel_tn(h, (const double*)Q, (const double*)GQ, (double*)H, N, p));
        dbg_hash(h, "sub.GQ", GQ, (size_t)N * p * 8);
        dbg_hash(h, "sub.H", H, (size_t)p * p * 8);
        int64_t sw = 0;
        // (the Ritz vectors of a random block's first step only sort the block and count, and the step itself leaves them 1e-3
        //  from their limits: rotations down to 1e-5 instead of 2 eps sqrt(p); COLD_TOL0=0: full accuracy)
        const double rot_tol0 = cold && !hook && step == 0 && force_cgs2 && p <= 96 && !dev_is(DEV_COLD_TOL0, '0') ? 1e-5 : 0.0;
        TLSQ_TRY(symeig_f64(h, (const double*)H, p, p, (double*)HB, (double*)S, true, lamH_dev, &sw, true, false, true, rot_tol0));
        dbg_hash(h, "sub.S", S, (size_t)p * p * 8);
        if (sweeps) *sweeps += sw;
        }
        // X' = Q S,  G X' = (G Q) S
        // (straight into X: the old block is not an input any more; it only has to be permuted afterwards when the
        // Ritz values did not come out in descending order)
        const bool no_mailbox = dev_is(DEV_NO_MAILBOX, '1');
        const bool mail = h->mailbox && !no_mailbox && p <= 512 && (size_t)(2 * p + 10) * 8 <= h->mailbox_bytes;
        // (the eigenvalues of the p x p problem come in no order from the Jacobi solver, and from k_rr_small in the order of the
        //  block it was given - two Ritz values of a warm block change places every ten iterations or so: the kernel writes the
        //  pairs sorted by them.  The guard: the status word of the stage that may have declined)
        const double* sort_keys = p <= 512 && !dev_is(DEV_RITZ_SORT, '0') ? lamH_dev : nullptr;
        const double* sort_guard = (rr_fast || used_cholqr) ? stat_dev : nullptr;
        if (mail) {
            void* scal;
            TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
            unsigned int* arrivals = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(scal) + 320);   // self-resetting
            if (!h->mail_counter_ready) {   // (fresh workspace memory is not zero)
                TLSQ_HIP(h, hipMemsetAsync(arrivals, 0, 4, h->stream));
                h->mail_counter_ready = true;
            }
            const double seq = (h->mail_seq += 1.0);
            const bool spec_now = st.spec.enable && rr_fast && step == 0 && (N & 3) == 0 && nt_step <= 32 && p <= 32;
            SpecCtrl* ctrl = spec_now ? reinterpret_cast<SpecCtrl*>(reinterpret_cast<char*>(scal) + 2048) : nullptr;
            TLSQ_TRY(launch_ritz_finish(h, (const double*)Q, (const double*)GQ, (const double*)S, (double*)X,
                                        (double*)GX, theta_dev, res_dev, N, p, stat_dev, h->mailbox_dev, arrivals, seq, ctrl,
                                        inv_mu, st.spec.nukeA ? 1 : 0, sort_keys, sort_guard));
            if (spec_now) {
                if (st.spec.before_launch) st.spec.before_launch();
                st.spec.nct = nt_step <= 16 ? 1 : 2;
                TLSQ_TRY(tsmm_sel_dev(h, st.spec.Z, st.spec.z_f32, st.spec.ldz, (const double*)X, ctrl, st.spec.nct, st.spec.Vs,
                                      st.spec.Tout, st.spec.M, st.spec.M, N));
                st.spec.launched = true;
            }
            // poll the flag (the kernel publishes it once every workgroup has delivered); generous time-out, then the
            // classic read-back
            volatile double* mb = h->mailbox;
            const double t_poll = now_ms();
            bool got = false;
            for (;;) {
                if (mb[0] == seq) {
                    got = true;
                    break;
                }
                if (now_ms() - t_poll > 2000.0) break;
            }
            if (got) {
                for (int64_t i = 0; i < 2 * p + 5; ++i) host[(size_t)i] = mb[8 + i];
                st.spec.dev_ok = st.spec.launched && host[(size_t)(2 * p + 3)] != 0.0;
                st.spec.dev_r = (int64_t)host[(size_t)(2 * p + 4)];
            } else {
                st.spec.launched = false;
                // never seen in practice; do not pay the time-out again on this handle
                h->mailbox_bytes = 0;
                TLSQ_HIP(h, hipMemcpyAsync(host.data(), theta_dev, (size_t)(2 * p + 3) * 8, hipMemcpyDeviceToHost,
                                           h->stream));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            }
        } else {
            TLSQ_TRY(launch_ritz_finish(h, (const double*)Q, (const double*)GQ, (const double*)S, (double*)X,
                                        (double*)GX, theta_dev, res_dev, N, p, nullptr, nullptr, nullptr, 0.0, nullptr, 0.0, 1, sort_keys, sort_guard));
            TLSQ_HIP(h, hipMemcpyAsync(host.data(), theta_dev, (size_t)(2 * p + 3) * 8, hipMemcpyDeviceToHost,
                                       h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        }
        dbg_hash(h, "sub.X", X, (size_t)N * p * 8);
        dbg_hash(h, "sub.theta", theta_dev, (size_t)(2 * p) * 8);
        if (rr_fast && host[2 * p + 1] != 0.0) {
            // k_rr_small declined (it left X = the normalised columns of Q: same span): same step again on the classic path
            if (dbg) fprintf(stderr, "  subspace step %d: fused Rayleigh-Ritz declined (status %.0f, ||B - I|| = %.2e)\n", step, host[2 * p + 1], host[2 * p + 2]);
            rr_fast_declined = true;
            ++st.n_rr_declined;
            st.rr_streak = std::min(st.rr_streak + 1, 4);
            st.rr_skip = 1 << st.rr_streak;
            --step;
            --st.steps;
            continue;
        }
        if (rr_fast) {
            ++st.n_rr_fast;
            st.rr_streak = 0;
        }
        st.chol_piv = used_cholqr && host[2 * p + 1] == 0.0 ? host[2 * p + 2] : 0.0;
        st.chol_p = p;
        if (used_cholqr && one_pass && host[2 * p + 1] == 0.0 && host[2 * p + 2] < 0.25) {
            // the one-pass guess was wrong (the columns have moved closer together since the last step): Q is only
            // orthonormal to ~1e-10; same step again with both passes
            --step;
            --st.steps;
            continue;
        }
        if (used_cholqr && host[2 * p + 1] != 0.0) {
            // the panel was too ill-conditioned for CholeskyQR2 (it left Q alone): same step again with CGS2
            force_cgs2 = true;
            cgs2_sticky = true;
            --step;
            --st.steps;
            continue;
        }
        // a cold start is random only once: from the second step on the block consists of Ritz vectors, whose images under
        // G^q are nearly orthogonal again (different norms do not hurt the Cholesky factor) - CholeskyQR2 (18 us instead of 60)
        // unless it has already failed on this block
        const bool cold_cgs2 = dev_is(DEV_COLD_CGS2, '1');
        if (cold && !hook && !cgs2_sticky && !cold_cgs2) force_cgs2 = false;   // (the randomized hook keeps its two plain passes)
        s.sigma.resize((size_t)p);
        double tmax = 0.0;
        bool finite = true;
        for (int64_t i = 0; i < p; ++i) {
            const double t = host[i];
            if (!std::isfinite(t) || !std::isfinite(host[p + i])) finite = false;
            tmax = std::max(tmax, t);
            s.sigma[i] = std::sqrt(std::max(t, 0.0));
        }
        if (!finite) {
            st.valid = false;
            st.fail = SubspaceState::FAIL_NUMERIC;
            break;
        }
        s.ncols = p;
        // Pad columns far below the threshold may stand in any order (round 6): their Ritz values are the rounding noise of G in the
        // late iterations - 1e-12 of the top - and the eigenvalue estimates the device sorted them by (k_ritz_finish) differ from the
        // Rayleigh quotients read here in the third digit; nothing looks at their order (the counted columns come first, sorted),
        // but a strict sort cost two panel copies, two gathers and the speculative factor product every time two of them swapped.
        bool pads_only = !dev_is(DEV_RITZ_SORT, '0');
        for (int64_t i = 0; i + 1 < p && pads_only; ++i)
            if (s.sigma[(size_t)i] < s.sigma[(size_t)i + 1] && !(s.sigma[(size_t)i + 1] < 0.5 * inv_mu)) pads_only = false;
        if (pads_only) {
            s.order.resize((size_t)p);
            std::iota(s.order.begin(), s.order.end(), 0);
        } else {
            sort_desc(s);
        }
        // keep the block sorted by Ritz value: X = X'[:, order]
        {
            std::vector<double> res_sorted((size_t)p), th_sorted((size_t)p), sg_sorted((size_t)p);
            for (int64_t i = 0; i < p; ++i) {
                res_sorted[i] = host[p + s.order[i]];
                th_sorted[i] = host[s.order[i]];
                sg_sorted[i] = s.sigma[s.order[i]];
            }
            bool sorted = true;
            for (int64_t i = 0; i < p; ++i) sorted = sorted && s.order[i] == (int32_t)i;
            x_settled = sorted && mail;   // (a re-ordering is queued on the main stream: the block is in flux until that has run)
            if (!sorted) {
                TLSQ_HIP(h, hipMemcpyAsync(XN, X, (size_t)N * p * 8, hipMemcpyDeviceToDevice, h->stream));
                TLSQ_TRY(upload_async(h, aux, s.order.data(), (size_t)p * 4));
                TLSQ_TRY(launch_gather_scale(h, (const double*)XN, N, (const int32_t*)aux, nullptr, p, nullptr,
                                             (double*)X));
                // ... and G X with it (two small launches against one operator product saved in the next step)
                TLSQ_HIP(h, hipMemcpyAsync(XN, GX, (size_t)N * p * 8, hipMemcpyDeviceToDevice, h->stream));
                TLSQ_TRY(launch_gather_scale(h, (const double*)XN, N, (const int32_t*)aux, nullptr, p, nullptr,
                                             (double*)GX));
            }
            for (int64_t i = 0; i < p; ++i) {
                host[i] = th_sorted[i];
                host[p + i] = res_sorted[i];
                s.sigma[i] = sg_sorted[i];
            }
            std::iota(s.order.begin(), s.order.end(), 0);
            gx_valid = !dev_is(DEV_NO_GX_REUSE, '1');
        }
        svp = 0;
        for (int64_t i = 0; i < p; ++i) svp += (s.sigma[i] >= inv_mu) ? 1 : 0;
        if (hook) {
            if (step + 1 < max_steps) continue;
            s.ncols = std::min<int64_t>(st.hook_rank, p);   // rank-sv truncation, like `svd(Z, sv)`
            *V_out = (double*)X;
            *ok = true;
            return TLSQ_OK;
        }
        if (svp > p - 2) {  // the block may not contain every sigma >= 1/mu: full solver, or a larger block
            st.fail = SubspaceState::FAIL_SMALL;
            break;
        }
        bool good = true;
        double maxres = 0.0;
        for (int64_t i = 0; i < svp; ++i) {
            good = good && (host[p + i] <= 2e-13 * tmax);
            maxres = std::max(maxres, host[p + i]);
        }
        if (dbg) {
            int sd = -1;
            void* scal = h->ws[WS_SCAL].p;
            (void)hipMemcpy(&sd, (char*)scal + 136, 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "  [small eig sweeps %d]", sd);
            if (rr_fast) {
                double st8[8];
                (void)hipMemcpy(st8, stat_dev, 64, hipMemcpyDeviceToHost);
                fprintf(stderr, " [rr: its %.0f delta %.2e delta_tt %.2e k_tp %.2e blocked %.0f piv %.2e]", st8[3], st8[2], st8[5], st8[6], st8[7], st8[0]);
            }
        }
        if (dbg)
            fprintf(stderr, "  subspace step %d: p=%lld ntop=%lld svp=%lld maxres/tmax=%.3e tail/tau=%.3f cold=%d\n", step,
                    (long long)p, (long long)ntop, (long long)svp, maxres / tmax,
                    svp < p ? s.sigma[s.order[svp]] / inv_mu : 0.0, (int)cold);
        if (good) {
            conv = true;
            if (cold && !hook) {
                // The panel changes more between the first two ALM iterations (Y is still zero in the first) than it ever does
                // again: with the default count the first warm step misses the residual bound and a second one (~190 us on the
                // classic path) follows.  Two more multiplications of the top columns (~13 us) avoid that; the count relaxes
                // by itself afterwards (below).  TLSQ_WARM_Q0 overrides.
                const char* e = dev_get(DEV_WARM_Q0);
                const int v = e ? atoi(e) : 0;
                st.q_warm = std::max(st.q_warm, v >= 1 && v <= 7 ? v : 5);
                // (... and the relaxation stops at 3: probing further down costs a failed step sooner or later)
                if (!e) st.q_floor = std::max(st.q_floor, 3);
            }
            if (!cold && !hook) {
                // a warm block that needed a second step just missed the residual bound after the first one: two more
                // multiplications of its top columns next time are far cheaper than another step; relax again later
                // ... and when a single step landed far below the bound (the spectral gap behind the block grows by
                // rho^2 per ALM iteration) give a multiplication back - but never return to a count that has failed
                if (step >= 1) {
                    st.q_floor = std::max(st.q_floor, std::min(st.q_warm + 1, 3));
                    st.q_warm = std::min(7, st.q_warm + 2);
                } else if (st.q_warm > st.q_floor && maxres <= 0.02 * 2e-13 * tmax) {
                    st.q_warm -= 1;
                }
            }
            break;
        }
        // (the fused Rayleigh-Ritz kernel leaves clusters of Ritz values unresolved - fine for pad columns, not for a cluster
        //  that reaches into the wanted pairs: the next step of this call goes through the Jacobi solver)
        if (rr_fast) rr_fast_declined = true;
        // hopeless (no spectral gap behind the block): stop early and let the full solver run
        if (step >= 4 && prev_maxres > 0.0 && maxres > 0.5 * prev_maxres) break;
        prev_maxres = maxres;
    }
    if (!conv) {
        if (st.fail == SubspaceState::FAIL_NONE) st.fail = SubspaceState::FAIL_NOCONV;
        return TLSQ_OK;
    }
    {   // count window: a Ritz value this close to the threshold cannot be trusted to fall on the right side
        const double tau2 = inv_mu * inv_mu;
        st.dlam = st.noise_rel * host[0];   // (sorted: host[0] is the largest Ritz value)
        bool in_window = !(tau2 > 2.0 * st.dlam);
        for (int64_t i = 0; i < p && !in_window; ++i) in_window = std::fabs(host[i] - tau2) <= st.dlam;
        if (in_window) {
            st.fail = SubspaceState::FAIL_WINDOW;
            return TLSQ_OK;
        }
    }
    if (st.skip_certificate) {
        *V_out = (double*)X;
        *ok = true;
        return TLSQ_OK;
    }
    // ---- certificate: lambda_max(G - X_r Theta_r X_r') must be clearly below (1/mu)^2 ----
    void *Vg = nullptr, *Vs = nullptr;
    // (explicit G, at most 32 deflated columns, N < 1024: the deflation kernel reads the columns of X itself)
    const bool no_fused_defl = dev_is(DEV_NO_FUSED_DEFLATE, '1');
    const bool fused_deflate = !op.implicit() && svp <= 32 && N < 1024 && !no_fused_defl;
    SelWeights defl_sw;
    if (svp > 0) {
        std::vector<int32_t> sel((size_t)svp);
        std::vector<double> th((size_t)svp);
        for (int64_t i = 0; i < svp; ++i) {
            sel[i] = s.order[i];
            th[i] = host[sel[i]];
        }
        if (fused_deflate) {
            for (int64_t i = 0; i < 32; ++i) {
                defl_sw.sel[i] = i < svp ? sel[(size_t)i] : 0;
                defl_sw.w[i] = i < svp ? th[(size_t)i] : 0.0;
            }
        } else {
            TLSQ_TRY(ws_get(h, WS_VG, (size_t)N * svp * 8, &Vg));
            TLSQ_TRY(ws_get(h, WS_VS, (size_t)N * svp * 8, &Vs));
            TLSQ_TRY(gather_scale_host(h, (const double*)X, N, sel, th, aux, (double*)Vg, (double*)Vs));
        }
    }
    const double tau2 = inv_mu * inv_mu;
    // everything below is scaled by 1 / tau^2: the question is lambda_max(GD) < margin = 1 - dlam / tau^2
    st.cert_margin = (1.0 - st.dlam / tau2) * (1.0 - 1e-9);
    if (!op.implicit()) {
        TLSQ_TRY(ws_get(h, WS_GD, (size_t)N * N * 8, &GD));
        const bool no_power = dev_is(DEV_NO_POWER_CERT, '1');
        // the two dense squarings cost N^3 flops against ~16 N^2 loads for a Lanczos run: matrix powers up to N = 1024
        st.cert_power = N <= 1024 && !no_power;
        // Asynchronous form: the host has just read this step's results from the mailbox, so everything the certificate reads
        // (G, the block X) is complete - its two kernels go to the second stream and run beside whatever the caller queues next
        const bool async = st.cert_async && st.defer_certificate && st.cert_power && fused_deflate && svp > 0 && h->stream_b &&
                           x_settled && h->mailbox && h->mailbox_bytes >= 32768 && !dev_is(DEV_NO_MAILBOX, '1');
        st.cert_async = async;
        st.cert_GD = (const double*)GD;
        st.cert_N = N;
        st.cert_launch = nullptr;
        if (async) {
            const double* Gp = op.G;
            const double* Xp = (const double*)X;
            double* GDp = (double*)GD;
            const double sc = 1.0 / tau2;
            SubspaceState* stp = &st;
            st.cert_launch = [h, Gp, Xp, GDp, N, svp, sc, defl_sw, stp]() -> int {
                StreamScope on_b(h, h->stream_b);
                TLSQ_TRY(launch_deflate_sel(h, Gp, N, Xp, defl_sw, GDp, N, svp, sc));
                return power_cert_begin(h, *stp);
            };
        } else {
            if (fused_deflate && svp > 0)
                TLSQ_TRY(launch_deflate_sel(h, op.G, N, (const double*)X, defl_sw, (double*)GD, N, svp, 1.0 / tau2));
            else
                TLSQ_TRY(launch_deflate(h, op.G, N, (const double*)Vs, (const double*)Vg, (double*)GD, N, svp, 1.0 / tau2));
            if (st.cert_power) TLSQ_TRY(power_cert_begin(h, st));
            else TLSQ_TRY(lanczos_begin(h, st.cert, (const double*)GD, N, N, 0.02, 48, st.cert_margin, 0.0));
        }
        if (st.defer_certificate) {
            st.cert_pending = true;
            *V_out = (double*)X;
            *ok = true;   // tentatively: svd_subspace_certify has the last word
            return TLSQ_OK;
        }
        bool pass = false;
        TLSQ_TRY(cert_finish(h, st, &pass));
        if (!pass) {
            st.fail = SubspaceState::FAIL_CERT;   // ambiguous: the accurate route decides (or a larger block)
            return TLSQ_OK;
        }
    } else {
        // the deflated operator as a product: w = (G q - Vs (Vg' q)); Lanczos bound, unscaled
        void* cv;
        TLSQ_TRY(ws_get(h, WS_SH, (size_t)std::max<int64_t>(p * p, svp) * 8, &cv));
        const LzApply apply = [&](const double* q, double* w) -> int {
            TLSQ_TRY(op_apply(h, op, N, q, w, 1));
            if (svp > 0) TLSQ_TRY(launch_deflate_vec(h, (const double*)Vs, (const double*)Vg, svp, q, (double*)cv, w, N));
            return TLSQ_OK;
        };
        double lmax = 0.0;
        int steps = 0;
        const int lst = lanczos_lmax_op(h, N, apply, 0.02, 48, &lmax, &steps, tau2);
        if (lst < 0) return lst;
        if (!(lmax * 1.5 + st.dlam < tau2)) {
            st.fail = SubspaceState::FAIL_CERT;
            return TLSQ_OK;
        }
    }
    *V_out = (double*)X;
    *ok = true;
    return TLSQ_OK;
}

// Second half of a deferred count certificate (SubspaceState::defer_certificate): waits for the certificate's numbers
// only - whatever the caller queued behind them keeps running.
int svd_subspace_certify(Handle* h, SubspaceState& st, double inv_mu, bool* ok) {
    (void)inv_mu;
    *ok = false;
    st.cert_pending = false;
    bool pass = false;
    TLSQ_TRY(cert_finish(h, st, &pass));
    if (!pass) {
        st.fail = SubspaceState::FAIL_CERT;
        return TLSQ_OK;
    }
    *ok = true;
    return TLSQ_OK;
}

// Factors of the thresholded low-rank matrix A = Z * V[:,sel] * diag(g) * V[:,sel]' (r = sel.size() columns):
// Tm (M x r, ld M, fp64, WS_T) = Z * V[:,sel] * diag(g) and Vs (N x r, ld N, WS_VS) = V[:,sel].  r = 0: both nullptr.

// Carry the dominant block (svp + pad Ritz/eigen vectors, sorted) to the next ALM iteration: WS_SX = V[:, top].
// Called after rebuild_lowrank (which has finished reading V, and V may alias WS_SX).
int carry_block(Handle* h, const double* V, int64_t N, const SmallSvd& s, int64_t svp, int64_t pmax,
                       SubspaceState& sub) {
    const int64_t pad_min = [] { const char* e = dev_get(DEV_PAD); return (int64_t)(e ? atoi(e) : 4); }();
    int64_t pad = std::max<int64_t>(pad_min, svp / 4);
    // up to 64 (96) columns the p x p Rayleigh-Ritz problem is solved in a single launch (k_jacobi_small / _mid);
    // beyond that it costs ~1 ms per step: give up some padding to stay below when the rank allows
    if (svp + pad > 64 && svp + pad_min <= 64) pad = 64 - svp;
    else if (svp + pad > 96 && svp + pad_min <= 96) pad = 96 - svp;   // (k_jacobi_mid: one launch up to 96 as well)
    int64_t want = std::min<int64_t>(N, svp + pad);
    if (N > kFullEigMaxN) want = std::min(want, pmax);   // large mode has no other solver: keep what fits
    if (want > pmax || want < 3) {
        sub.valid = false;
        // rank beyond the largest block: cold starts would only find that out again - the dense solver serves the
        // next iterations until the rank fits (this function is called after every one of them)
        sub.allow_cold = want < 3;
        return TLSQ_OK;
    }
    sub.allow_cold = true;
    // the sorted vectors we have (a subspace result only carries p of them); any missing pad columns are
    // pseudo-random — the next iteration's CGS2 orthogonalises them against the rest
    const int64_t have = std::min<int64_t>(want, s.ncols);
    std::vector<int32_t> keep((size_t)have);
    bool identity = true;
    for (int64_t p = 0; p < have; ++p) {
        keep[p] = s.order[p];
        identity = identity && keep[p] == (int32_t)p;
    }
    if (identity && h->ws[WS_SX].p && V == (const double*)h->ws[WS_SX].p) {
        // the usual case: V is the sorted block the subspace solver left in WS_SX - its leading columns stay where
        // they are, only missing pad columns are (re)filled
        if (have < want) {
            double* Xb = (double*)h->ws[WS_SX].p;
            TLSQ_TRY(launch_fill_hash(h, Xb + (size_t)N * have, N * (want - have), 0x85EBCA6Bu));
            // (round 6: the fresh pad columns are taken out of the span of the block first.  A random column is mostly dominant
            //  subspace after its first product, the block's Gram matrix 2.1 away from I, and whether the fused Rayleigh-Ritz
            //  kernel still converged on it in the second ALM iteration of BASELINE config 2 - or that iteration and the next
            //  two went through the Jacobi path, +0.36 ms - hung on the last bit of ||D||_2.  PAD_PROJECT=0: as before)
            if (have <= 512 && !dev_is(DEV_PAD_PROJECT, '0')) {
                void* W;
                TLSQ_TRY(ws_get(h, WS_SH, (size_t)std::max<int64_t>(want * want, have * (want - have)) * 8, &W));
                TLSQ_TRY(launch_project_out(h, Xb, have, Xb + (size_t)N * have, want - have, (double*)W, N));
            }
        }
        sub.p = want;
        sub.ntop = svp;
        sub.valid = true;
        return TLSQ_OK;
    }
    void *tmp, *X;
    TLSQ_TRY(ws_get(h, WS_SXN, (size_t)N * want * 8, &tmp));
    TLSQ_TRY(gather_cols(h, V, N, keep, (double*)tmp));          // out of place (V may be WS_SX itself)
    if (have < want) TLSQ_TRY(launch_fill_hash(h, (double*)tmp + (size_t)N * have, N * (want - have), 0x85EBCA6Bu));
    TLSQ_TRY(ws_get(h, WS_SX, (size_t)N * want * 8, &X));
    TLSQ_HIP(h, hipMemcpyAsync(X, tmp, (size_t)N * want * 8, hipMemcpyDeviceToDevice, h->stream));
    sub.p = want;
    sub.ntop = svp;
    sub.valid = true;
    return TLSQ_OK;
}


}  // namespace tlsq
