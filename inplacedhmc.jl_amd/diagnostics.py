"""Diagnostics on the engine's outputs: the reference's Diagnostics.EBFMI and summarize_tree_statistics
(src/diagnostics.jl:28-32, 61-127) plus the effective-sample-size estimator the reference lacks but
BASELINE.json's parity criterion names (Geyer initial-positive-sequence, per coordinate)."""
from dataclasses import dataclass

import numpy as np

ACCEPTANCE_QUANTILES = [0.05, 0.25, 0.5, 0.75, 0.95]  # src/diagnostics.jl:35
MAX_DIRECTIONS_DEPTH = 32                              # src/tree.jl:132


def EBFMI(tree_statistics):
    """Energy Bayesian fraction of missing information: mean(abs2, diff(pi)) / var(pi) (src/diagnostics.jl:28-32).
    A 2-D array (chains x draws) gives one value per chain."""
    ts = np.asarray(tree_statistics)
    if ts.ndim == 2:
        return np.array([EBFMI(r) for r in ts])
    pis = ts["pi"].astype(np.float64)
    return float(np.mean(np.diff(pis) ** 2) / np.var(pis, ddof=1))


def is_divergent(left, right):
    return left == right                               # src/tree.jl:285


@dataclass
class TreeStatisticsSummary:                           # src/diagnostics.jl:44-55
    N: int
    a_mean: float
    a_quantiles: np.ndarray
    termination_counts: dict
    depth_counts: np.ndarray

    def __str__(self):
        pct = lambda v: "%d%%" % round(100.0 * v / max(self.N, 1))
        return ("Hamiltonian Monte Carlo sample of length %d\n  acceptance rate mean: %.2f, 5/25/50/75/95%%: %s\n"
                "  termination: %s\n  depth: %s" % (
                    self.N, self.a_mean, " ".join("%.2f" % q for q in self.a_quantiles),
                    ", ".join("%s => %s" % (k, pct(v)) for k, v in sorted(self.termination_counts.items())),
                    ", ".join("%d => %s" % (d, pct(c)) for d, c in enumerate(self.depth_counts))))


def summarize_tree_statistics(tree_statistics):
    """src/diagnostics.jl:61-101"""
    ts = np.asarray(tree_statistics).ravel()
    left, right = ts["term_left"], ts["term_right"]
    maxd = int(np.sum((left == 1) & (right == 0)))      # REACHED_MAX_DEPTH, src/tree.jl:300
    div = int(np.sum(is_divergent(left, right)))
    counts = np.bincount(ts["depth"], minlength=1)
    nz = np.nonzero(counts)[0]
    counts = counts[: (nz[-1] + 1) if len(nz) else 0]
    return TreeStatisticsSummary(len(ts), float(ts["acceptance_rate"].mean()),
                                 np.quantile(ts["acceptance_rate"], ACCEPTANCE_QUANTILES),
                                 {"max_depth": maxd, "divergence": div, "turning": len(ts) - maxd - div}, counts)


def summary_from_counters(counters):
    """TreeStatisticsSummary from the device-side integer counters (Engine.diag_counters(), possibly summed over ranks):
    the library's idhmc_tree_summary_from_counters.  Quantiles come from a 1024-bin histogram (within 1/1024 of the sample
    quantile; the reference prints two digits)."""
    import ctypes as C
    from . import _lib
    cn = np.ascontiguousarray(counters, dtype=np.uint64)
    if cn.shape != (_lib.DIAG_COUNTERS,):
        raise ValueError("expected %d counters" % _lib.DIAG_COUNTERS)
    out = _lib.TreeSummary()
    _lib.check(_lib.load().idhmc_tree_summary_from_counters(cn.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(out)))
    depth = np.array(out.depth_counts[:], dtype=np.int64)
    nz = np.nonzero(depth)[0]
    depth = depth[: (nz[-1] + 1) if len(nz) else 0]
    return TreeStatisticsSummary(int(out.N), float(out.a_mean), np.array(out.a_quantiles[:]),
                                 {"max_depth": int(out.max_depth), "divergence": int(out.divergence), "turning": int(out.turning)},
                                 depth)


def ess(x):
    """Effective sample size of each column of x (draws x dims): Geyer's initial positive sequence."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    n = x.shape[0]
    xc = x - x.mean(axis=0)
    nfft = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(xc, n=nfft, axis=0)
    acov = np.fft.irfft(f * np.conj(f), n=nfft, axis=0)[:n] / n
    var = acov[0]
    out = np.empty(x.shape[1])
    for d in range(x.shape[1]):
        rho = acov[:, d] / var[d] if var[d] > 0 else np.zeros(n)
        s, k = 0.0, 0
        while 2 * k + 1 < n:
            pair = rho[2 * k] + rho[2 * k + 1]
            if pair < 0:
                break
            s += pair
            k += 1
        tau = max(2.0 * s - 1.0, 1.0 / n)
        out[d] = n / tau
    return out


def rhat_from_moments(mean, var, n):
    """Potential scale reduction per coordinate from the engine's device-side running moments (Engine.moments():
    per-chain mean and unbiased variance over n draws, shape chains x dims) -- at 65 536 chains the draws themselves
    never leave the device, the 2 x chains x dims moments do.  Gelman-Rubin: W = mean of chain variances,
    B/n = variance of chain means, R^2 = ((n-1)/n W + B/n) / W."""
    mean, var = np.asarray(mean, dtype=np.float64), np.asarray(var, dtype=np.float64)
    n = float(np.min(n)) if np.ndim(n) else float(n)
    W = var.mean(axis=0)
    Bn = mean.var(axis=0, ddof=1)
    return np.sqrt(((n - 1.0) / n * W + Bn) / W)


def ess_from_moments(mean, var, n, cap=False):
    """Total effective sample size per coordinate from the same moments, by replicated batch means with every chain as
    one batch (Vats, Flegal & Jones): the variance of the chain means estimates sigma^2 / ESS_chain directly, so
    ESS_total = chains * V / var(chain means) with V = (n-1)/n W + B/n the pooled posterior variance.  No
    autocorrelation estimate is involved and nothing saturates: antithetic chains (NUTS on a Gaussian) report more than
    chains * n, as they should.  Relative standard error ~ sqrt(2 / (chains - 1)).  cap=True limits it to chains * n
    (round 1's behaviour, which made every Gaussian run report exactly the cap)."""
    mean, var = np.asarray(mean, dtype=np.float64), np.asarray(var, dtype=np.float64)
    n = float(np.min(n)) if np.ndim(n) else float(n)
    C = mean.shape[0]
    W = var.mean(axis=0)
    Bn = mean.var(axis=0, ddof=1)
    V = (n - 1.0) / n * W + Bn
    e = C * V / Bn
    return np.minimum(e, C * n) if cap else e
