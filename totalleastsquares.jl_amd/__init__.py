"""tlsq-hip: MI355X-native robust PCA / low-rank recovery (drop-in for the rpca / lowrankfilter /
tls! / rtls path of baggepinnen/TotalLeastSquares.jl).  See DESIGN.md and INTEGRATION.md.

The directory name contains a dot, so import it through the repo-root shim:  `import tlsq_amd`.
"""
from . import _lib  # noqa: F401
from . import workloads  # noqa: F401
from ._lib import dev_set, dev_from_env, dev_switches  # noqa: F401
from .engine import (SVD, Engine, TlsqError, default_engine, hankel, ishankel, lowrankfilter, rpca,  # noqa: F401
                     rtls, soft_hankel_, tls, tls_, unhankel, rpca_ga, mu_, entrywise_trimmed_mean,
                     entrywise_median)

__all__ = ["SVD", "Engine", "TlsqError", "default_engine", "hankel", "ishankel", "lowrankfilter", "rpca",
           "rtls", "soft_hankel_", "tls", "tls_", "unhankel", "rpca_ga", "mu_", "entrywise_trimmed_mean", "entrywise_median"]
