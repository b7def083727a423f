"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as baggepinnen/TotalLeastSquares.jl
(paths relative to /root/reference):

    rpca(D; λ, maxrank, iters, tol, ρ, verbose, nonnegA, nonnegE, hankel, nukeA)   src/robustPCA.jl:156-239
    lowrankfilter(y, n; sv, lag, tol, ...)                                         src/robustPCA.jl:119-128
    hankel(x, L, lag) / unhankel(A[, lag, N, D]) / ishankel(A)                     src/robustPCA.jl:76-106, 28-68
    tls!(Ay, n) / tls!(s, n) / rtls(A, y)                                          src/TotalLeastSquares.jl:63-69, 152-156

The Julia shim (julia/TotalLeastSquaresHIP.jl) binds the same C entry points with `ccall`; Julia is not
present in the build image, so this ctypes mirror is the host side that is actually exercised by tests.
numpy arrays are passed as HOST pointers (column-major copies are made when needed); every matrix
operation runs in libtlsqhip.so on the GPU — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings
from collections import namedtuple

import numpy as np

from . import _lib as L

SVD = namedtuple("SVD", ["U", "S", "Vt"])
SVD.V = property(lambda s: s.Vt.conj().T)


class TlsqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tlsq error {code}: {msg}")
        self.code = code


def _f(a, dtype=np.float64):
    """column-major host array of the given dtype"""
    return np.asfortranarray(np.asarray(a, dtype=dtype))


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class RpcaReport:
    """Python view of tlsq_rpca_info.  The fields are read out of the C struct on first use (a solve of the headline config takes
    9 ms: twenty conversions per call are not free)."""

    def __init__(self, info, cost, svp):
        self._raw = (info, cost, svp)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        d = self.__dict__
        if "iters_done" not in d:
            info, cost, svp = d["_raw"]
            d["iters_done"] = int(info.iters_done)
            d["converged"] = bool(info.converged)
            d["final_cost"] = float(info.final_cost)
            d["final_mu"] = float(info.final_mu)
            d["d_norm"] = float(info.d_norm)
            d["cost_hist"] = cost[: d["iters_done"]].tolist()
            d["svp_hist"] = svp[: d["iters_done"]].tolist()
            d["jacobi_sweeps"] = int(info.jacobi_sweeps)
            d["eig_full"], d["eig_fast"] = int(info.eig_full), int(info.eig_fast)
            d["subspace_steps"] = int(info.subspace_steps)
            d["residual_stores_skipped"] = int(info.residual_stores_skipped)
            d["tsqr_iterations"] = int(info.tsqr_iterations)
            d["hbm_bytes_sweeps"], d["hbm_bytes"] = float(info.hbm_bytes_sweeps), float(info.hbm_bytes)
            d["sweeps_timed"], d["hbm_bytes_sweeps_timed"] = int(info.sweeps_timed), float(info.hbm_bytes_sweeps_timed)
            d["kern"] = {k[5:]: int(getattr(info, k)) for k, _ in info._fields_ if k.startswith("kern_")}
            d["ms"] = {k[3:]: float(getattr(info, k)) for k, _ in info._fields_ if k.startswith("ms_")}
        try:
            return d[name]
        except KeyError:
            raise AttributeError(name) from None


class Engine:
    """One handle = one GPU (tlsq_create)."""

    def __init__(self, device: int = 0, *, ngpus: int | None = None, devices=None):
        """Engine(device) = tlsq_create: one GPU.  Engine(ngpus=n) / Engine(devices=[...]) = tlsq_create_multi: one
        handle over n GPUs of this process; rpca / lowrankfilter on host arrays are row-sharded over them."""
        self.lib = L.load()
        h = C.c_void_p()
        if ngpus is not None or devices is not None:
            devs = list(devices) if devices is not None else list(range(int(ngpus)))
            arr = (C.c_int * len(devs))(*devs)
            st = self.lib.tlsq_create_multi(len(devs), arr, C.byref(h))
            device = devs[0]
        else:
            st = self.lib.tlsq_create(int(device), C.byref(h))
        if st != 0:
            raise TlsqError(st, "tlsq_create failed (no MI355X visible? there is no CPU fallback)")
        self.h = h
        self.device = device
        self.ngpus = int(self.lib.tlsq_ngpus(h))
        self.nranks, self.rank = 1, 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tlsq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ----------------------------------------------------------------------------------
    def _check(self, st):
        if st < 0:
            raise TlsqError(st, self.lib.tlsq_last_error(self.h).decode())
        return st

    def stream(self):
        return self.lib.tlsq_stream(self.h)

    def synchronize(self):
        self._check(self.lib.tlsq_synchronize(self.h))

    @staticmethod
    def _hook_modes(svd, opnorm):
        """Map the reference's `svd` / `opnorm` keyword hooks (src/robustPCA.jl:168-169) onto the library's modes.
        None = LinearAlgebra.svd! / opnorm.  "randomized" (or "rsvd", "rsvd_fnkz", "tsvd") = rank-sv randomized SVD
        for k >= 2 on the GPU;  "power" / ("power", mvps) = the rnorm(x, mvps) estimator on the GPU.  Any other
        CALLABLE is the user's own function: svd(Z, sv) -> (U, S, Vt) (or an object with .U, .S, .Vt), opnorm(X) ->
        float; it runs on the host through the C callback (the panel is copied out for every call)."""
        svd_mode, opn_mode, mvps = L.SVD_FULL, L.OPNORM_EXACT, 10
        if svd is not None:
            if isinstance(svd, str) and svd.lower() in ("randomized", "rsvd", "rsvd_fnkz", "tsvd"):
                svd_mode = L.SVD_RANDOMIZED
            elif callable(svd):
                svd_mode = L.SVD_CALLBACK
            else:
                raise TlsqError(L.TLSQ_ERR_UNSUPPORTED, "svd hook: a callable, or svd='randomized'")
        if opnorm is not None:
            if isinstance(opnorm, tuple) and len(opnorm) == 2 and str(opnorm[0]).lower() in ("power", "rnorm"):
                opn_mode, mvps = L.OPNORM_POWER, int(opnorm[1])
            elif isinstance(opnorm, str) and opnorm.lower() in ("power", "rnorm"):
                opn_mode = L.OPNORM_POWER
            elif callable(opnorm):
                opn_mode = L.OPNORM_CALLBACK
            else:
                raise TlsqError(L.TLSQ_ERR_UNSUPPORTED, "opnorm hook: a callable, or opnorm=('power', mvps)")
        return svd_mode, opn_mode, mvps

    @staticmethod
    def _wrap_hooks(svd, opnorm, dt, errbox):
        """ctypes callbacks around Python callables (kept alive by the caller for the duration of the call)"""
        ct = C.c_float if dt == np.float32 else C.c_double

        def view(p, rows, cols, ld):   # column-major (rows x cols, ld) host matrix behind a raw pointer
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(cols, ld)).T[:rows]

        svd_cb = opn_cb = None
        if callable(svd):
            def _svd(Zp, M, N, ldZ, sv, Up, ldU, Sp, Vtp, ldVt, kout, user):
                try:
                    r = svd(np.array(view(Zp, M, N, ldZ)), int(sv))
                    U, S, Vt = (r.U, r.S, r.Vt) if hasattr(r, "Vt") else r
                    k = min(len(S), min(M, N))
                    view(Up, M, min(M, N), ldU)[:, :k] = np.asarray(U)[:, :k]
                    np.ctypeslib.as_array(C.cast(Sp, C.POINTER(ct)), shape=(min(M, N),))[:k] = np.asarray(S)[:k]
                    view(Vtp, min(M, N), N, ldVt)[:k, :] = np.asarray(Vt)[:k, :]
                    kout[0] = k
                    return 0
                except Exception as e:   # noqa: BLE001  (must not propagate through the C frames)
                    errbox.append(e)
                    return 1
            svd_cb = L.SVD_CB(_svd)
        if callable(opnorm):
            def _opn(Xp, M, N, ldX, user):
                try:
                    return float(opnorm(np.array(view(Xp, M, N, ldX))))
                except Exception as e:   # noqa: BLE001
                    errbox.append(e)
                    return float("nan")
            opn_cb = L.OPNORM_CB(_opn)
        return svd_cb, opn_cb

    def make_opts(self, *, lam=None, maxrank=None, iters=None, tol=None, rho=None, nonnegA=False,
                  nonnegE=False, hankel=False, nukeA=True, memory=L.MEM_HOST, m_global=0,
                  svd_mode=L.SVD_FULL, opnorm_mode=L.OPNORM_EXACT, opnorm_mvps=10, seed=0, on_iter=None,
                  svd_cb=None, opnorm_cb=None, phase_timing=False):
        o = L.RpcaOpts()
        self.lib.tlsq_rpca_opts_default(C.byref(o))
        if lam is not None:
            o.lambda_ = float(lam)
        if maxrank is not None:
            o.maxrank = int(min(maxrank, 2 ** 62))
        if iters is not None:
            o.iters = int(iters)
        if tol is not None:
            o.tol = float(tol)
        if rho is not None:
            o.rho = float(rho)
        o.nonnegA, o.nonnegE, o.hankel, o.nukeA = int(nonnegA), int(nonnegE), int(hankel), int(nukeA)
        o.memory = memory
        o.m_global = int(m_global)
        o.svd_mode, o.opnorm_mode = svd_mode, opnorm_mode
        o.opnorm_mvps, o.seed = int(opnorm_mvps), int(seed)
        if on_iter is not None:
            o.on_iter = on_iter
        o.phase_timing = int(bool(phase_timing))
        if svd_cb is not None:
            o.svd_cb = svd_cb
        if opnorm_cb is not None:
            o.opnorm_cb = opnorm_cb
        return o

    @staticmethod
    def _info(capacity, cost_history=True):
        cost = np.zeros(max(capacity, 1), dtype=np.float64)
        svp = np.zeros(max(capacity, 1), dtype=np.int64)
        info = L.RpcaInfo()
        if cost_history:   # asking for it makes the library evaluate opnorm(residual) exactly in every iteration
            info.cost_hist = cost.ctypes.data_as(C.POINTER(C.c_double))
        info.svp_hist = svp.ctypes.data_as(C.POINTER(C.c_int64))
        info.hist_capacity = capacity
        return info, cost, svp

    # -- multi-GPU --------------------------------------------------------------------------------
    def unique_id(self) -> bytes:
        buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
        st = self.lib.tlsq_comm_unique_id(buf)
        if st != 0:
            raise TlsqError(st, "tlsq_comm_unique_id failed (librccl not loadable?)")
        return buf.raw

    def comm_size(self) -> int:
        """ranks of the attached communicator as RCCL reports them (1 without one)"""
        n = C.c_int(0)
        self._check(self.lib.tlsq_comm_size(self.h, C.byref(n)))
        return int(n.value)

    def comm_init(self, nranks: int, rank: int, uid: bytes):
        self._check(self.lib.tlsq_comm_init(self.h, nranks, rank, C.create_string_buffer(uid, L.UNIQUE_ID_BYTES)))
        self.nranks, self.rank = nranks, rank

    # -- rpca -------------------------------------------------------------------------------------
    def rpca(self, D, *, lam=None, maxrank=None, iters=1000, tol=None, rho=None, verbose=False,
             nonnegA=False, nonnegE=False, hankel=False, nukeA=True, svd=None, opnorm=None,
             want_U=True, want_s=True, cost_history=True, return_report=False, m_global=0, **kwargs):
        """A, E, s, sv = rpca(D; ...) — src/robustPCA.jl:156-239.  Unknown kwargs are swallowed like the
        reference's `kwargs...` (:170).  `svd`/`opnorm` hooks: only the defaults run on the GPU.
        want_U / want_s = False skip the left vectors / the whole returned SVD (s is None then: no extra
        decomposition after the loop); cost_history=False lets the library settle only `cost < tol` per iteration
        (what a plain, non-verbose Julia call observes)."""
        svd_mode, opn_mode, mvps = self._hook_modes(svd, opnorm)
        if int(iters) < 1:   # (the reference leaves `s` undefined and throws for iters = 0, src/robustPCA.jl:185,238)
            raise ValueError("iters must be >= 1")
        D = np.asarray(D)
        if np.iscomplexobj(D):
            if svd_mode != L.SVD_FULL or opn_mode != L.OPNORM_EXACT:
                raise TlsqError(L.TLSQ_ERR_UNSUPPORTED, "hook modes are not available for complex data")
            return self._rpca_complex(D, lam=lam, maxrank=maxrank, iters=iters, tol=tol, rho=rho, verbose=verbose,
                                      nonnegA=nonnegA, nonnegE=nonnegE, hankel=hankel, nukeA=nukeA,
                                      return_report=return_report)
        dt = np.float32 if D.dtype == np.float32 else np.float64
        Df = _f(D, dt)
        M, N = Df.shape
        d = min(max(m_global, M), N)
        A = np.empty((M, N), dtype=dt, order="F")
        E = np.empty((M, N), dtype=dt, order="F")
        U = np.empty((M, d), dtype=dt, order="F") if (want_U and want_s) else None
        S = np.empty(d, dtype=dt) if want_s else None
        Vt = np.empty((d, N), dtype=dt, order="F") if want_s else None
        cb = None
        if verbose:
            def _print(k, cost, svp, user):
                print(f"{k} cost: {float(f'{cost:.4g}')}")                 # :226
            cb = L.ON_ITER(_print)
        errbox = []
        svd_cb, opn_cb = self._wrap_hooks(svd, opnorm, dt, errbox)
        o = self.make_opts(lam=lam, maxrank=maxrank, iters=iters, tol=tol, rho=rho, nonnegA=nonnegA,
                           nonnegE=nonnegE, hankel=hankel, nukeA=nukeA, m_global=m_global, on_iter=cb,
                           svd_mode=svd_mode, opnorm_mode=opn_mode, opnorm_mvps=mvps, seed=kwargs.get("seed", 0),
                           svd_cb=svd_cb, opnorm_cb=opn_cb)
        info, cost, svp = self._info(int(iters), cost_history)
        sv = C.c_int64(0)
        fn = self.lib.tlsq_rpca_f32 if dt == np.float32 else self.lib.tlsq_rpca_f64
        st = fn(
            self.h, _ptr(Df), M, N, M, C.byref(o), _ptr(A), M, _ptr(E), M,
            _ptr(U) if U is not None else None, M, _ptr(S) if want_s else None, _ptr(Vt) if want_s else None, d,
            C.byref(sv), C.byref(info))
        if errbox:
            raise errbox[0]          # the user's hook raised: surface its own exception
        st = self._check(st)
        rep = RpcaReport(info, cost, svp)
        if verbose and rep.converged:
            print("converged")                                             # :229
        if st == L.TLSQ_MAXITER:                                           # :232
            warnings.warn(f"Maximum number of iterations reached, cost: {rep.final_cost}, tol: "
                          f"{tol if tol is not None else math.sqrt(np.finfo(dt).eps)}")
        s = SVD(U, S, Vt) if want_s else None
        if return_report:
            return A, E, s, int(sv.value), rep
        return A, E, s, int(sv.value)

    def _rpca_complex(self, D, *, lam, maxrank, iters, tol, rho, verbose, nonnegA, nonnegE, hankel, nukeA,
                      return_report):
        """Complex data (src/robustPCA.jl:3-7); `s` = (U, S, Vt) of the last Z like the real path (:194, :238).  ComplexF32
        input (numpy complex64) stays ComplexF32 - eltype-generic like the reference - through tlsq_rpca_c32_svd."""
        c32 = np.asarray(D).dtype == np.complex64
        ct, rt = (np.complex64, np.float32) if c32 else (np.complex128, np.float64)
        Df = np.asfortranarray(D, dtype=ct)
        M, N = Df.shape
        d = min(M, N)
        A = np.empty((M, N), dtype=ct, order="F")
        E = np.empty((M, N), dtype=ct, order="F")
        S = np.empty(d, dtype=rt)
        U = np.empty((M, d), dtype=ct, order="F")
        Vt = np.empty((d, N), dtype=ct, order="F")
        cb = None
        if verbose:
            cb = L.ON_ITER(lambda k, cost, svp, user: print(f"{k} cost: {float(f'{cost:.4g}')}"))
        o = self.make_opts(lam=lam, maxrank=maxrank, iters=iters, tol=tol, rho=rho, nonnegA=nonnegA, nonnegE=nonnegE,
                           hankel=hankel, nukeA=nukeA, on_iter=cb)
        info, cost, svp = self._info(int(iters))
        sv = C.c_int64(0)
        fn = self.lib.tlsq_rpca_c32_svd if c32 else self.lib.tlsq_rpca_c64_svd
        st = self._check(fn(self.h, _ptr(Df), M, N, M, C.byref(o), _ptr(A), M, _ptr(E), M,
                            _ptr(U), M, _ptr(S), _ptr(Vt), d, C.byref(sv), C.byref(info)))
        rep = RpcaReport(info, cost, svp)
        if st == L.TLSQ_MAXITER:
            warnings.warn(f"Maximum number of iterations reached, cost: {rep.final_cost}")
        s = SVD(U, S, Vt)
        return (A, E, s, int(sv.value), rep) if return_report else (A, E, s, int(sv.value))

    def rpca_device(self, dD, M, N, dA, dE, *, dU=None, dS=None, dVt=None, iters=1000, m_global=0,
                    want_hist=True, dtype=np.float64, **optkw):
        """rpca on DEVICE-resident column-major panels (raw device addresses as ints): D (M x N, ld M) in,
        A, E (M x N) out, optional U (M x d), S (d), Vt (d x N); dtype float64 or float32 (all panels).  Nothing
        crosses PCIe except O(N) scalars.  Returns (sv, RpcaReport, status)."""
        o = self.make_opts(iters=iters, memory=L.MEM_DEVICE, m_global=m_global, **optkw)
        info, cost, svp = self._info(int(iters))
        if not want_hist:   # no per-iteration cost requested: the library only settles cost < tol (see tlsq.h)
            info.cost_hist = None
        sv = C.c_int64(0)
        d = min(max(m_global, M), N)
        vp = lambda x: C.c_void_p(int(x)) if x else None
        fn = self.lib.tlsq_rpca_f32 if np.dtype(dtype) == np.float32 else self.lib.tlsq_rpca_f64
        st = self._check(fn(self.h, vp(dD), M, N, M, C.byref(o), vp(dA), M, vp(dE), M,
                            vp(dU), M, vp(dS), vp(dVt), d, C.byref(sv), C.byref(info)))
        return int(sv.value), RpcaReport(info, cost, svp), st

    # -- hankel family ----------------------------------------------------------------------------
    def hankel(self, x, L_, lag=1):
        """src/robustPCA.jl:76-92.  float64 / float32 run on the GPU as they are; integer (and bool) series - the
        reference's own test uses `hankel(1:10, 3)`, test/runtests.jl:293 - are embedded in float64 on the device and
        converted back: the kernel only copies, so the result is exact for |x| < 2^53."""
        x = np.asarray(x)
        if x.dtype.kind in "iub":
            if x.size and np.abs(x.astype(np.float64)).max() >= 2.0 ** 53:
                raise ValueError("hankel: integers beyond 2^53 do not survive the float64 embedding")
            return np.rint(self.hankel(x.astype(np.float64), L_, lag)).astype(x.dtype)
        dt = np.float32 if x.dtype == np.float32 else np.float64
        x2 = _f(x.reshape(x.shape[0], -1), dt)
        Nx, Dch = x2.shape
        assert L_ <= Nx / 2, f"L has to be less than N/2 = {Nx / 2}"        # :79
        assert lag <= L_, "lag must be <= L"                                # :80
        K = (Nx - L_) // lag + 1
        X = np.empty((K, L_ * Dch), dtype=dt, order="F")
        fn = self.lib.tlsq_hankel_f32 if dt == np.float32 else self.lib.tlsq_hankel_f64
        self._check(fn(self.h, _ptr(x2), Nx, Dch, Nx, L_, lag, _ptr(X), K, L.MEM_HOST))
        return X

    def unhankel(self, A, lag=None, N=None, D=1):
        """src/robustPCA.jl:28-39 and :53-68."""
        A = np.asarray(A)
        dt = np.float32 if A.dtype == np.float32 else np.float64
        Af = _f(A, dt)
        K, LD = Af.shape
        if lag is None:
            lag, D = 1, 1
            N = LD + K - 1
        if lag == 1 and D == 1:
            N = LD + K - 1                                                  # :54 -> unhankel(A)
        y = np.empty((N, D), dtype=dt, order="F")
        fn = self.lib.tlsq_unhankel_f32 if dt == np.float32 else self.lib.tlsq_unhankel_f64
        self._check(fn(self.h, _ptr(Af), K, LD, K, lag, N, D, _ptr(y), N, L.MEM_HOST))
        return y[:, 0].copy() if D == 1 else y

    def soft_hankel_(self, A, eps):
        """In place soft_hankel! — src/robustPCA.jl:9-21.  Returns the (column-major) result."""
        Af = _f(A)
        K, L_ = Af.shape
        self._check(self.lib.tlsq_soft_hankel_f64(self.h, _ptr(Af), K, L_, K, float(eps), L.MEM_HOST))
        if isinstance(A, np.ndarray) and A.dtype == np.float64:
            A[...] = Af
        return Af

    def lowrankfilter(self, y, n=None, *, sv=0, lag=1, tol=1e-3, svd=None, return_report=False,
                      cost_history=True, **kw):
        """src/robustPCA.jl:119-128.  cost_history=False: like a plain Julia call, nobody looks at the
        per-iteration cost, so the library only settles `cost < tol` (same result, less work)."""
        opnorm = kw.pop("opnorm", None)
        svd_mode, opn_mode, mvps = self._hook_modes(svd, opnorm)
        y = np.asarray(y)
        dt = np.float32 if y.dtype == np.float32 else np.float64      # eltype(y) as in the reference (generic T)
        y = np.asarray(y, dtype=dt)
        y2 = _f(y.reshape(y.shape[0], -1), dt)
        Nx, Dch = y2.shape
        if n is None:
            n = min(Nx // 20, 2000)
        assert n <= Nx / 2, f"L has to be less than N/2 = {Nx / 2}"
        assert lag <= n, "lag must be <= L"
        iters = int(kw.pop("iters", 1000))
        verbose = kw.pop("verbose", False)
        allowed = {k: kw[k] for k in ("lam", "maxrank", "rho", "nonnegA", "nonnegE", "hankel", "nukeA", "phase_timing")
                   if k in kw}
        cb = None
        if verbose:
            cb = L.ON_ITER(lambda k, cost, svp, user: print(f"{k} cost: {float(f'{cost:.4g}')}"))
        errbox = []
        svd_cb, opn_cb = self._wrap_hooks(svd, opnorm, dt, errbox)
        o = self.make_opts(iters=iters, tol=tol, on_iter=cb, svd_mode=svd_mode, opnorm_mode=opn_mode,
                           opnorm_mvps=mvps, seed=kw.pop("seed", 0), svd_cb=svd_cb, opnorm_cb=opn_cb, **allowed)
        info, cost, svp = self._info(iters, cost_history)
        yf = np.empty((Nx, Dch), dtype=dt, order="F")
        fn = self.lib.tlsq_lowrankfilter_f32 if dt == np.float32 else self.lib.tlsq_lowrankfilter_f64
        st = fn(self.h, _ptr(y2), Nx, Dch, Nx, int(n), int(lag), int(sv), C.byref(o), _ptr(yf), Nx, C.byref(info))
        if errbox:
            raise errbox[0]
        st = self._check(st)
        rep = RpcaReport(info, cost, svp)
        if st == L.TLSQ_MAXITER:
            warnings.warn(f"Maximum number of iterations reached, cost: {rep.final_cost}, tol: {tol}")
        out = yf[:, 0].copy() if y.ndim == 1 else yf
        return (out, rep) if return_report else out

    # -- tls! / rtls ------------------------------------------------------------------------------
    def tls_(self, Ay, n):
        """tls!(Ay, n) or tls!(s::SVD, n) — src/TotalLeastSquares.jl:63-69."""
        if isinstance(Ay, SVD):
            Vt = _f(Ay.Vt)
            nc = Vt.shape[1]
            if Vt.shape[0] != nc:
                raise TlsqError(L.TLSQ_ERR_ARG, "tls!(s, n) needs the full ncols x ncols Vt")
            x = np.empty((n, nc - n), dtype=np.float64, order="F")
            st = self.lib.tlsq_tls_from_vt_f64(_ptr(Vt), nc, nc, n, _ptr(x), n)
            if st < 0:
                raise TlsqError(st, "tls_from_vt failed")
            return x
        Ay = np.asarray(Ay)
        dt = np.float32 if Ay.dtype == np.float32 else np.float64     # generic element type like the reference (:63)
        Af = _f(Ay, dt)
        M, nc = Af.shape
        x = np.empty((n, nc - n), dtype=dt, order="F")
        fn = self.lib.tlsq_tls_f32 if dt == np.float32 else self.lib.tlsq_tls_f64
        self._check(fn(self.h, _ptr(Af), M, nc, M, n, _ptr(x), n, L.MEM_HOST))
        return x

    def tls(self, A, y):
        """tls(A, y) — src/TotalLeastSquares.jl:48-55: the out-of-place form, x = tls!([A y], size(A, 2)).
        A vector y gives a vector x (Julia's `-V21/V22` with a 1 x 1 V22 is n x 1; the reference's tests use `vec`)."""
        A = np.asarray(A)
        dt = np.float32 if A.dtype == np.float32 else np.float64
        A = np.asarray(A, dtype=dt)
        yv = np.asarray(y, dtype=dt)
        x = self.tls_(np.hstack([A, yv.reshape(A.shape[0], -1)]), A.shape[1])
        return x[:, 0].copy() if yv.ndim == 1 else x

    def rtls(self, A, y, *, return_report=False, **kw):
        """rtls(A, y; kwargs...) — src/TotalLeastSquares.jl:152-156; the kwargs go to rpca as in the reference."""
        A = np.asarray(A)
        dt = np.float32 if A.dtype == np.float32 else np.float64
        A = _f(A, dt)
        yv = np.asarray(y, dtype=dt)
        y2 = _f(yv.reshape(yv.shape[0], -1), dt)
        M, n = A.shape
        q = y2.shape[1]
        iters = int(kw.pop("iters", 1000))
        kw.pop("nukeA", None)
        kw.pop("verbose", None)
        o = self.make_opts(iters=iters, **{k: v for k, v in kw.items()
                                           if k in ("lam", "maxrank", "tol", "rho", "nonnegA", "nonnegE", "hankel")})
        info, cost, svp = self._info(iters)
        x = np.empty((n, q), dtype=dt, order="F")
        fn = self.lib.tlsq_rtls_f32 if dt == np.float32 else self.lib.tlsq_rtls_f64
        st = self._check(fn(self.h, _ptr(A), M, n, M, _ptr(y2), q, M, C.byref(o), _ptr(x), n, C.byref(info)))
        rep = RpcaReport(info, cost, svp)
        if st == L.TLSQ_MAXITER:
            warnings.warn(f"Maximum number of iterations reached, cost: {rep.final_cost}")
        out = x[:, 0].copy() if yv.ndim == 1 else x
        return (out, rep) if return_report else out

    # -- batched tiny problems -----------------------------------------------
    def rpca_batched(self, D, *, iters=1000, **kw):
        """rpca on every D[b] of a (batch, M, N) stack, N <= 32, one workgroup per problem.  Returns
        A, E (batch, M, N), S (batch, N), Vt (batch, N, N), sv, iters, status, cost (batch,).  A float32 stack is
        solved in float32 (default tol sqrt(eps(Float32))), everything else in float64."""
        D = np.asarray(D)
        dt = np.float32 if D.dtype == np.float32 else np.float64
        D = D.astype(dt, copy=False)
        fn = self.lib.tlsq_rpca_batched_f32 if dt == np.float32 else self.lib.tlsq_rpca_batched_f64
        B, M, N = D.shape
        Df = np.ascontiguousarray(np.transpose(D, (0, 2, 1)))          # each problem column-major
        A = np.empty_like(Df)
        E = np.empty_like(Df)
        S = np.empty((B, N), dtype=dt)
        Vt = np.empty((B, N, N), dtype=dt)
        sv = np.empty(B, dtype=np.int64)
        it = np.empty(B, dtype=np.int32)
        stt = np.empty(B, dtype=np.int32)
        cost = np.empty(B, dtype=dt)
        o = self.make_opts(iters=int(iters), **{k: v for k, v in kw.items()
                                                if k in ("lam", "maxrank", "tol", "rho", "nonnegA", "nonnegE", "nukeA")})
        st = self._check(fn(self.h, _ptr(Df), M, N, B, C.byref(o), _ptr(A), _ptr(E), _ptr(S), _ptr(Vt), _ptr(sv),
                            _ptr(it), _ptr(stt), _ptr(cost)))
        if st == L.TLSQ_MAXITER:
            if int((stt == 1).sum()):
                warnings.warn(f"Maximum number of iterations reached in {int((stt == 1).sum())} of {B} problems")
            if int((stt == 2).sum()):
                warnings.warn(f"{int((stt == 2).sum())} of {B} problems contain Infs or NaNs (status 2, NaN results)")
        return (np.transpose(A, (0, 2, 1)), np.transpose(E, (0, 2, 1)), S, np.transpose(Vt, (0, 2, 1)), sv, it, stt,
                cost)

    def rtls_batched(self, A, y, *, iters=1000, return_status=False, **kw):
        """x[b] = rtls(A[b], y[b]) for a (batch, M, n) stack A and a (batch, M) or (batch, M, q) stack y (float32 when
        both stacks are float32, float64 otherwise)."""
        A = np.asarray(A)
        yv = np.asarray(y)
        dt = np.float32 if (A.dtype == np.float32 and yv.dtype == np.float32) else np.float64
        A = A.astype(dt, copy=False)
        yv = yv.astype(dt, copy=False)
        fn = self.lib.tlsq_rtls_batched_f32 if dt == np.float32 else self.lib.tlsq_rtls_batched_f64
        B, M, n = A.shape
        y3 = yv.reshape(B, M, -1)
        q = y3.shape[2]
        Af = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
        yf = np.ascontiguousarray(np.transpose(y3, (0, 2, 1)))
        x = np.empty((B, q, n), dtype=dt)
        it = np.empty(B, dtype=np.int32)
        stt = np.empty(B, dtype=np.int32)
        o = self.make_opts(iters=int(iters), **{k: v for k, v in kw.items()
                                                if k in ("lam", "maxrank", "tol", "rho", "nonnegA", "nonnegE")})
        st = self._check(fn(self.h, _ptr(Af), _ptr(yf), M, n, q, B, C.byref(o), _ptr(x), _ptr(it), _ptr(stt)))
        if st == L.TLSQ_MAXITER:
            if int((stt == 1).sum()):
                warnings.warn(f"Maximum number of iterations reached in {int((stt == 1).sum())} of {B} problems")
            if int((stt == 2).sum()):
                warnings.warn(f"{int((stt == 2).sum())} of {B} problems contain Infs or NaNs (status 2, NaN results)")
        out = np.transpose(x, (0, 2, 1))
        out = out[:, :, 0].copy() if yv.ndim == 2 else out
        return (out, it, stt) if return_status else out

    # ---- rpca_ga (src/robustPCA.jl:255-310) and the spherical averages (:312-362) -------------------------------
    @staticmethod
    def _average_code(mu):
        """The reference's `μ` keyword is a function (`μ!`, `entrywise_trimmed_mean`, `entrywise_median`); here it is
        named by one of those functions' Python counterparts below, their names, or None (= μ!)."""
        name = getattr(mu, "__name__", mu)
        table = {None: L.GA_MEAN, "mu_": L.GA_MEAN, "μ!": L.GA_MEAN, "mean": L.GA_MEAN,
                 "entrywise_trimmed_mean": L.GA_TRIMMED_MEAN, "entrywise_median": L.GA_MEDIAN}
        # this module's own mu_ / entrywise_* (and their names) select the device averages; ANY other callable is the
        # reference's `μ = f` with a user function: it runs on the host through the C callback (TLSQ_GA_CALLBACK)
        own = mu is None or isinstance(mu, str) or getattr(mu, "__module__", "").endswith("engine")
        if own and name in table:
            return table[name]
        if callable(mu):
            return L.GA_CALLBACK
        raise TlsqError(L.TLSQ_ERR_UNSUPPORTED, f"rpca_ga: average {name!r} is neither a device average nor a callable")

    def rpca_ga(self, X, r=None, *, tol=1e-7, iters=1000, mu=None, P=0.1, q0=None, seed=0, verbose=False,
                return_report=False):
        """Q = rpca_ga(X, r; tol, iters, μ): X is d x N with the observations in its columns; returns Q (d x r).
        q0 (d x r) gives the start vector of every component (the reference draws randn(d), :289)."""
        Xa = np.asarray(X)
        # Float32 observations take the fp32 entry (the panel travels as float, is widened on the device, Q comes back in float);
        # a caller's own average always sees float64 (the callback's signature)
        f32 = Xa.dtype == np.float32 and self._average_code(mu) != L.GA_CALLBACK
        dt = np.float32 if f32 else np.float64
        Xf = _f(Xa, dt)
        d, N = Xf.shape
        r = min(d, N) if r is None else int(r)
        Q = np.zeros((d, r), dtype=dt, order="F")
        o = L.GaOpts()
        self.lib.tlsq_ga_opts_default(C.byref(o))
        o.tol, o.iters, o.average, o.trim, o.seed = float(tol), int(iters), self._average_code(mu), float(P), int(seed)
        errbox = []
        if o.average == L.GA_CALLBACK:
            def _avg(sp, wp, Up, dd, NN, ldU, user):
                # `μᵢ = μ(q, w, U)` (src/robustPCA.jl:297): s holds the current q on entry; the result is what the function
                # returns (or s itself when it works in place and returns None)
                try:
                    sv = np.ctypeslib.as_array(sp, shape=(dd,))
                    wv = np.ctypeslib.as_array(wp, shape=(NN,))
                    Uv = np.ctypeslib.as_array(Up, shape=(NN, ldU)).T[:dd]      # column-major d x N view
                    out = mu(sv, wv, Uv)
                    if out is not None and out is not sv:
                        sv[:] = np.asarray(out, dtype=np.float64).reshape(dd)
                    return 0
                except BaseException as e:   # noqa: BLE001 - must not unwind through the C frames
                    errbox.append(e)
                    return 1
            keep_cb = L.GA_AVG_CB(_avg)
            o.avg_cb = keep_cb
        it = np.zeros(max(r, 1), dtype=np.int64)
        stt = np.zeros(max(r, 1), dtype=np.int32)
        dq = np.zeros(max(r, 1))
        cap = int(iters) if (verbose or return_report) else 0
        hist = np.full((max(r, 1), max(cap, 1)), np.nan)
        info = L.GaInfo()
        info.iters = it.ctypes.data_as(C.POINTER(C.c_int64))
        info.status = stt.ctypes.data_as(C.POINTER(C.c_int32))
        info.dq = dq.ctypes.data_as(C.POINTER(C.c_double))
        info.dq_hist = hist.ctypes.data_as(C.POINTER(C.c_double)) if cap else None
        info.hist_capacity = cap
        q0f = None if q0 is None else _f(np.asarray(q0, dtype=dt).reshape(d, -1), dt)
        if q0f is not None and q0f.shape[1] < r:
            raise TlsqError(L.TLSQ_ERR_ARG, "rpca_ga: q0 needs one column per component")
        fn = self.lib.tlsq_rpca_ga_f32 if f32 else self.lib.tlsq_rpca_ga_f64
        st = fn(self.h, _ptr(Xf), d, N, d, r, C.byref(o), _ptr(q0f) if q0f is not None else None, d, _ptr(Q), d, C.byref(info))
        if errbox:
            raise errbox[0]          # the user's average raised: surface its own exception
        st = self._check(st)
        if verbose:
            for i in range(r):
                for k in range(int(it[i])):
                    print(f"Change at iteration {k + 1}: {hist[i, k]}")            # :300
                if not stt[i]:
                    print(f"Converged after {int(it[i])} iterations")               # :302
        if st == L.TLSQ_MAXITER:
            warnings.warn("Reached maximum number of iterations")                  # :306
        if return_report:
            return Q, {"iters": it[:r].tolist(), "status": stt[:r].tolist(), "dq": dq[:r].tolist(),
                       "dq_hist": [hist[i, : int(it[i])].tolist() for i in range(r)] if cap else None,
                       "ms_loop": float(info.ms_loop), "ms_total": float(info.ms_total), "passes": int(info.passes)}
        return Q

    def _average(self, code, s, w, U, P=float("nan")):
        Uf = _f(U)
        d, N = Uf.shape
        wf = np.ascontiguousarray(np.asarray(w, dtype=np.float64).reshape(-1))
        if wf.size != N:
            raise TlsqError(L.TLSQ_ERR_ARG, "average: w needs one weight per column of U")
        out = np.zeros(d)
        self._check(self.lib.tlsq_ga_average_f64(self.h, code, float(P), _ptr(wf), _ptr(Uf), d, N, d, _ptr(out),
                                                  L.MEM_HOST))
        s[...] = out.reshape(np.shape(s))
        return s

    def mu_(self, s, w, U):
        """μ!(s,w,U), src/robustPCA.jl:312-320 — writes into s and returns it."""
        return self._average(L.GA_MEAN, s, w, U)

    def entrywise_trimmed_mean(self, s, w, U, P=0.1):
        """src/robustPCA.jl:327-337"""
        return self._average(L.GA_TRIMMED_MEAN, s, w, U, P)

    def entrywise_median(self, s, w, U):
        """src/robustPCA.jl:354-362"""
        return self._average(L.GA_MEDIAN, s, w, U)


def ishankel(A):
    """src/robustPCA.jl:94-106 — exact test of constant anti-diagonals (host-side test helper)."""
    A = np.asarray(A)
    K, L_ = A.shape
    F = np.fliplr(A)
    for off in range(-(K - 1), L_):
        v = np.diagonal(F, off)
        if np.any(v != v[0]):
            return False
    return True


_default = None


def default_engine() -> Engine:
    global _default
    if _default is None:
        _default = Engine(0)
    return _default


def rpca(D, **kw):
    return default_engine().rpca(D, **kw)


def lowrankfilter(y, n=None, **kw):
    return default_engine().lowrankfilter(y, n, **kw)


def hankel(x, L_, lag=1):
    return default_engine().hankel(x, L_, lag)


def unhankel(A, lag=None, N=None, D=1):
    return default_engine().unhankel(A, lag, N, D)


def soft_hankel_(A, eps):
    return default_engine().soft_hankel_(A, eps)


def tls_(Ay, n):
    return default_engine().tls_(Ay, n)


def tls(A, y):
    return default_engine().tls(A, y)


def rpca_ga(X, r=None, **kw):
    return default_engine().rpca_ga(X, r, **kw)


def mu_(s, w, U):
    return default_engine().mu_(s, w, U)


def entrywise_trimmed_mean(s, w, U, P=0.1):
    return default_engine().entrywise_trimmed_mean(s, w, U, P)


def entrywise_median(s, w, U):
    return default_engine().entrywise_median(s, w, U)


def rtls(A, y, **kw):
    return default_engine().rtls(A, y, **kw)
