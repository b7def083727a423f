"""ctypes binding of libtlsqhip.so — mirrors include/tlsq.h one to one.

There is deliberately no fallback: if the shared object is missing or a GPU call fails, an exception
is raised (the product path never routes through oracle/ or any CPU implementation).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TLSQ_LIB") or os.path.join(_HERE, "libtlsqhip.so")  # TLSQ_LIB: development builds

TLSQ_OK, TLSQ_MAXITER = 0, 1
TLSQ_ERR_ARG, TLSQ_ERR_HIP, TLSQ_ERR_OOM, TLSQ_ERR_COMM, TLSQ_ERR_UNSUPPORTED, TLSQ_ERR_NOCONV = -1, -2, -3, -4, -5, -6
TLSQ_ERR_NONFINITE = -7   # the input contains Infs or NaNs (the reference: ArgumentError from LAPACK's chkfinite)
MEM_HOST, MEM_DEVICE = 0, 1
SVD_FULL, SVD_RANDOMIZED, SVD_CALLBACK = 0, 1, 2
OPNORM_EXACT, OPNORM_POWER, OPNORM_CALLBACK = 0, 1, 2
UNIQUE_ID_BYTES = 128
GA_MEAN, GA_TRIMMED_MEAN, GA_MEDIAN, GA_CALLBACK = 0, 1, 2, 3

ON_ITER = C.CFUNCTYPE(None, C.c_int64, C.c_double, C.c_int64, C.c_void_p)
SVD_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                     C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p)
OPNORM_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)


class RpcaOpts(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("maxrank", C.c_int64), ("iters", C.c_int64),
                ("tol", C.c_double), ("rho", C.c_double),
                ("nonnegA", C.c_int32), ("nonnegE", C.c_int32), ("hankel", C.c_int32),
                ("nukeA", C.c_int32), ("svd_mode", C.c_int32), ("opnorm_mode", C.c_int32),
                ("opnorm_mvps", C.c_int32), ("memory", C.c_int32),
                ("m_global", C.c_int64), ("seed", C.c_uint64),
                ("on_iter", ON_ITER), ("user", C.c_void_p), ("svd_cb", SVD_CB), ("opnorm_cb", OPNORM_CB),
                ("phase_timing", C.c_int32), ("reserved0", C.c_int32)]


class RpcaInfo(C.Structure):
    _fields_ = [("iters_done", C.c_int64), ("converged", C.c_int32), ("tsqr_iterations", C.c_int32),
                ("final_cost", C.c_double), ("final_mu", C.c_double), ("d_norm", C.c_double),
                ("cost_hist", C.POINTER(C.c_double)), ("svp_hist", C.POINTER(C.c_int64)),
                ("hist_capacity", C.c_int64), ("jacobi_sweeps", C.c_int64),
                ("ms_total", C.c_double), ("ms_loop", C.c_double), ("ms_h2d", C.c_double),
                ("ms_d2h", C.c_double), ("ms_shrink", C.c_double), ("ms_update", C.c_double),
                ("ms_gram", C.c_double), ("ms_eig", C.c_double), ("ms_rebuild", C.c_double),
                ("ms_opnorm", C.c_double),
                ("eig_full", C.c_int64), ("eig_fast", C.c_int64), ("subspace_steps", C.c_int64),
                ("residual_stores_skipped", C.c_int64),
                ("hbm_bytes_sweeps", C.c_double), ("hbm_bytes", C.c_double),
                ("sweeps_timed", C.c_int64), ("hbm_bytes_sweeps_timed", C.c_double),
                ("kern_gram_h3", C.c_int64), ("kern_zx_h", C.c_int64), ("kern_zty_h", C.c_int64),
                ("kern_zsweep_wide", C.c_int64), ("kern_fused_zgram", C.c_int64)]


GA_AVG_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64,
                        C.c_int64, C.c_int64, C.c_void_p)


class GaOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("iters", C.c_int64), ("average", C.c_int32), ("memory", C.c_int32),
                ("trim", C.c_double), ("seed", C.c_uint64), ("avg_cb", GA_AVG_CB), ("user", C.c_void_p)]


class GaInfo(C.Structure):
    _fields_ = [("iters", C.POINTER(C.c_int64)), ("status", C.POINTER(C.c_int32)), ("dq", C.POINTER(C.c_double)),
                ("dq_hist", C.POINTER(C.c_double)), ("hist_capacity", C.c_int64),
                ("ms_total", C.c_double), ("ms_loop", C.c_double), ("passes", C.c_int64)]


# every symbol include/tlsq.h declares (tests check that the .so exports all of them)
EXPORTS = [
    "tlsq_version", "tlsq_dev_set", "tlsq_rpca_opts_default", "tlsq_create", "tlsq_create_multi", "tlsq_ngpus", "tlsq_destroy", "tlsq_last_error",
    "tlsq_stream", "tlsq_synchronize", "tlsq_comm_unique_id", "tlsq_comm_init", "tlsq_comm_destroy", "tlsq_comm_size",
    "tlsq_rpca_f64", "tlsq_rpca_f32",
    "tlsq_hankel_f64", "tlsq_unhankel_f64", "tlsq_soft_hankel_f64",
    "tlsq_hankel_f32", "tlsq_unhankel_f32", "tlsq_soft_hankel_f32",
    "tlsq_lowrankfilter_f64", "tlsq_lowrankfilter_f32", "tlsq_tls_f64", "tlsq_rtls_f64", "tlsq_tls_f32", "tlsq_rtls_f32", "tlsq_tls_from_vt_f64",
    "tlsq_rpca_batched_f64", "tlsq_rtls_batched_f64", "tlsq_rpca_batched_f32", "tlsq_rtls_batched_f32", "tlsq_rpca_c64", "tlsq_rpca_c64_svd", "tlsq_rpca_c32_svd",
    "tlsq_ga_opts_default", "tlsq_rpca_ga_f64", "tlsq_rpca_ga_f32", "tlsq_ga_average_f64",
    "tlsq_k_shrink_f64", "tlsq_k_update_f64", "tlsq_k_shrink_f32", "tlsq_k_update_f32",
    "tlsq_k_update_shrink_f64", "tlsq_k_update_shrink_f32", "tlsq_k_rebuild_update_shrink_f64",
    "tlsq_k_zsweep_f64", "tlsq_k_zsweep_gram_f64", "tlsq_k_zsweep_wide_f32", "tlsq_k_final_e_f64", "tlsq_k_matfun_sign_f64", "tlsq_k_matfun_invsqrt_f64", "tlsq_k_rr_small_f64",
    "tlsq_k_gram_f64", "tlsq_k_gram_f32", "tlsq_k_op_gram_f32", "tlsq_k_gemm_nn_f64", "tlsq_k_gemm_nt_f64", "tlsq_k_symeig_f64", "tlsq_k_symeig_chol_f64",
    "tlsq_k_opnorm_f64", "tlsq_k_maxabs_f64", "tlsq_k_tsqr_f64", "tlsq_k_svd_r_f64",
]

_lib = None


def load():
    """Load libtlsqhip.so; raises if it has not been built (python totalleastsquares.jl_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing — build it with `python totalleastsquares.jl_amd/build.py` "
            "(hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32, dbl, flt = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_float
    P = C.POINTER
    lib.tlsq_version.restype = C.c_char_p
    lib.tlsq_dev_set.argtypes = [C.c_char_p, C.c_char_p]
    lib.tlsq_rpca_opts_default.argtypes = [P(RpcaOpts)]
    lib.tlsq_rpca_opts_default.restype = None
    lib.tlsq_create.argtypes = [i32, P(vp)]
    lib.tlsq_create_multi.argtypes = [i32, P(i32), P(vp)]
    lib.tlsq_ngpus.argtypes = [vp]
    lib.tlsq_destroy.argtypes = [vp]
    lib.tlsq_last_error.argtypes = [vp]
    lib.tlsq_last_error.restype = C.c_char_p
    lib.tlsq_stream.argtypes = [vp]
    lib.tlsq_stream.restype = vp
    lib.tlsq_synchronize.argtypes = [vp]
    lib.tlsq_comm_unique_id.argtypes = [C.c_char_p]
    lib.tlsq_comm_init.argtypes = [vp, i32, i32, C.c_char_p]
    lib.tlsq_comm_destroy.argtypes = [vp]
    lib.tlsq_comm_size.argtypes = [vp, P(i32)]
    lib.tlsq_rpca_f64.argtypes = [vp, vp, i64, i64, i64, P(RpcaOpts), vp, i64, vp, i64, vp, i64, vp,
                                  vp, i64, P(i64), P(RpcaInfo)]
    lib.tlsq_rpca_f32.argtypes = lib.tlsq_rpca_f64.argtypes
    for suf, sc in (("f64", dbl), ("f32", flt)):
        getattr(lib, "tlsq_hankel_" + suf).argtypes = [vp, vp, i64, i64, i64, i64, i64, vp, i64, i32]
        getattr(lib, "tlsq_unhankel_" + suf).argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, vp, i64, i32]
        getattr(lib, "tlsq_soft_hankel_" + suf).argtypes = [vp, vp, i64, i64, i64, sc, i32]
        getattr(lib, "tlsq_k_shrink_" + suf).argtypes = [vp, vp, vp, vp, vp, vp, i64, sc, sc, i32]
        getattr(lib, "tlsq_k_update_" + suf).argtypes = [vp, vp, vp, vp, vp, vp, i64, sc, i32]
        getattr(lib, "tlsq_k_update_shrink_" + suf).argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, sc, i32, sc, sc, i32]
    lib.tlsq_k_zsweep_f64.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, dbl, dbl, i32, dbl, dbl, i32, vp]
    lib.tlsq_k_zsweep_gram_f64.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, dbl, dbl, i32, dbl, dbl, i32, vp,
                                           vp, i64, vp, i64]
    lib.tlsq_k_zsweep_wide_f32.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, i64, flt, flt, i32, flt, flt, i32, vp]
    lib.tlsq_k_final_e_f64.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, dbl, dbl, i32, i32]
    lib.tlsq_k_matfun_sign_f64.argtypes = [vp, vp, i64, vp, vp]
    lib.tlsq_k_matfun_invsqrt_f64.argtypes = [vp, vp, i64, dbl, vp, vp]
    lib.tlsq_k_rr_small_f64.argtypes = [vp, vp, vp, i64, i64, dbl, vp, vp, vp]
    lib.tlsq_k_rebuild_update_shrink_f64.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, dbl, i32, dbl, dbl,
                                                     i32]
    lib.tlsq_lowrankfilter_f64.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, P(RpcaOpts), vp, i64,
                                           P(RpcaInfo)]
    lib.tlsq_lowrankfilter_f32.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, P(RpcaOpts), vp, i64,
                                           P(RpcaInfo)]
    lib.tlsq_tls_f64.argtypes = [vp, vp, i64, i64, i64, i64, vp, i64, i32]
    lib.tlsq_rtls_f64.argtypes = [vp, vp, i64, i64, i64, vp, i64, i64, P(RpcaOpts), vp, i64, P(RpcaInfo)]
    lib.tlsq_tls_f32.argtypes = lib.tlsq_tls_f64.argtypes
    lib.tlsq_rtls_f32.argtypes = lib.tlsq_rtls_f64.argtypes
    lib.tlsq_tls_from_vt_f64.argtypes = [vp, i64, i64, i64, vp, i64]
    lib.tlsq_rpca_c64.argtypes = [vp, vp, i64, i64, i64, P(RpcaOpts), vp, i64, vp, i64, vp, P(i64), P(RpcaInfo)]
    lib.tlsq_rpca_c32_svd.argtypes = [vp, vp, i64, i64, i64, P(RpcaOpts), vp, i64, vp, i64, vp, i64, vp, vp, i64, P(i64),
                                      P(RpcaInfo)]
    lib.tlsq_rpca_c64_svd.argtypes = [vp, vp, i64, i64, i64, P(RpcaOpts), vp, i64, vp, i64, vp, i64, vp, vp, i64, P(i64),
                                      P(RpcaInfo)]
    lib.tlsq_rpca_batched_f64.argtypes = [vp, vp, i64, i64, i64, P(RpcaOpts), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tlsq_rtls_batched_f64.argtypes = [vp, vp, vp, i64, i64, i64, i64, P(RpcaOpts), vp, vp, vp]
    lib.tlsq_rpca_batched_f32.argtypes = lib.tlsq_rpca_batched_f64.argtypes
    lib.tlsq_rtls_batched_f32.argtypes = lib.tlsq_rtls_batched_f64.argtypes
    lib.tlsq_ga_opts_default.argtypes = [P(GaOpts)]
    lib.tlsq_ga_opts_default.restype = None
    lib.tlsq_rpca_ga_f64.argtypes = [vp, vp, i64, i64, i64, i64, P(GaOpts), vp, i64, vp, i64, P(GaInfo)]
    lib.tlsq_rpca_ga_f32.argtypes = lib.tlsq_rpca_ga_f64.argtypes
    lib.tlsq_ga_average_f64.argtypes = [vp, C.c_int, dbl, vp, vp, i64, i64, i64, vp, C.c_int]
    lib.tlsq_k_gram_f64.argtypes = [vp, vp, i64, i64, i64, vp, i64]
    lib.tlsq_k_gram_f32.argtypes = [vp, vp, i64, i64, i64, vp, i64, C.c_int]
    lib.tlsq_k_op_gram_f32.argtypes = [vp, vp, i64, i64, i64, vp, i64, vp]
    lib.tlsq_k_gemm_nn_f64.argtypes = [vp, vp, i64, i64, i64, vp, i64, i64, vp, i64]
    lib.tlsq_k_gemm_nt_f64.argtypes = [vp, vp, i64, i64, i64, vp, i64, i64, vp, i64]
    lib.tlsq_k_symeig_f64.argtypes = [vp, vp, i64, i64, vp, vp, i64, P(i64)]
    lib.tlsq_k_symeig_chol_f64.argtypes = [vp, vp, i64, i64, vp, vp, i64, P(i64)]
    lib.tlsq_k_tsqr_f64.argtypes = [vp, vp, i64, i64, i64, vp, i64]
    lib.tlsq_k_svd_r_f64.argtypes = [vp, vp, i64, i64, i64, vp, vp, i64, P(i64)]
    lib.tlsq_k_opnorm_f64.argtypes = [vp, vp, i64, i64, i64, P(dbl)]
    lib.tlsq_k_maxabs_f64.argtypes = [vp, vp, i64, P(dbl)]
    for name in EXPORTS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        if name not in ("tlsq_version", "tlsq_last_error", "tlsq_stream", "tlsq_rpca_opts_default",
                        "tlsq_ga_opts_default"):
            fn.restype = C.c_int
    _lib = lib
    return lib


_dev_shadow = {}   # what this process has set through dev_set (the library has no getter): name without prefix -> string


def _dev_key(name):
    return name[5:] if name.startswith("TLSQ_") else name


def dev_set(name, value):
    """tlsq_dev_set: select a development switch of the library by name (value None clears it); raises on unknown names.
    Process-wide state of the library: set it between solves, never while another thread is inside one."""
    lib = load()
    v = None if value is None else str(value).encode()
    if lib.tlsq_dev_set(name.encode(), v) != 0:
        raise ValueError(f"unknown tlsq development switch {name!r}")
    if value is None:
        _dev_shadow.pop(_dev_key(name), None)
    else:
        _dev_shadow[_dev_key(name)] = str(value)


# environment variables that look like switches but are not the library's (read by the Python / Julia hosts themselves)
_NOT_SWITCHES = {"TLSQ_LIB", "TLSQ_NGPUS", "TLSQ_DEVICE", "TLSQ_EXTRA_FLAGS", "TLSQ_BUILD_INCREMENTAL"}


def dev_from_env(environ=None):
    """Tools and tests that are steered from the shell call this explicitly: every TLSQ_<NAME> variable of the environment that
    names a switch of the library is handed to tlsq_dev_set; a variable the library does not know (one a harness set, a typo)
    is reported with a warning and skipped.  The library itself never looks at the environment."""
    import warnings
    environ = os.environ if environ is None else environ
    applied = {}
    for k, v in environ.items():
        if k.startswith("TLSQ_") and k not in _NOT_SWITCHES:
            try:
                dev_set(k, v)
            except ValueError:
                warnings.warn(f"{k} is not a development switch of libtlsqhip (ignored)")
                continue
            applied[k] = v
    return applied


class dev_switches:
    """with dev_switches(NO_ZSWEEP=1, ...): the switches are set inside the block and put back to what they were before it
    (nested and overlapping uses keep the outer setting)."""

    def __init__(self, **kw):
        self.kw = kw
        self.prev = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.prev[k] = _dev_shadow.get(_dev_key(k))
            dev_set(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            dev_set(k, self.prev.get(k))
        return False
