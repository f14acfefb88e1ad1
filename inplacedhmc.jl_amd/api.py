"""The reference's user-level surface for the hot path, same names / defaults / output shapes
(reference src/InplaceDHMC.jl:3-11 exports, src/mcmc.jl:109-159, src/warmup.jl:217-234,361-389,
src/NUTS.jl:204-220, src/stepsize.jl:16-38,173-193,251-259, src/hamiltonian.jl:33-74), driving the HIP
engine through the C ABI.  Julia keyword arguments become Python keyword arguments; `Val`/type
parameters become plain values (TuningNUTS{Diagonal} -> TuningNUTS(M="Diagonal")).

FindLocalOptimum runs the engine's own device L-BFGS behind the reference's stage contract (the reference's
optimiser, QuasiNewtonMethods.proptimize!, is not in the reference tree: include/idhmc.h,
idhmc_find_local_optimum).  Not mirrored: per-step progress reports (LogProgressReport reports per stage),
Symmetric (dense) metrics.
"""
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

from . import engine as _e


# ---- option structs -----------------------------------------------------------------------------
@dataclass(frozen=True)
class NUTS:
    """reference NUTS(; max_depth = 10, min_D = -1000.0, turn_statistic_configuration = Val(:generalized))"""
    max_depth: int = 10
    min_delta: float = -1000.0
    turn_statistic_configuration: str = "generalized"

    def __post_init__(self):
        if self.turn_statistic_configuration != "generalized":
            raise ValueError("only the generalized turn statistic is supported (as in the reference)")
        if not (0 < self.max_depth <= 15):
            raise ValueError("0 < max_depth <= 15")
        if not (self.min_delta < 0):
            raise ValueError("min_delta < 0")


@dataclass(frozen=True)
class DualAveraging:
    """reference DualAveraging(; d = 0.8, g = 0.05, k = 0.75, t0 = 10)"""
    delta: float = 0.8
    gamma: float = 0.05
    kappa: float = 0.75
    t0: int = 10


@dataclass(frozen=True)
class FixedStepsize:
    """reference FixedStepsize: adaptation placeholder that leaves eps unchanged"""


@dataclass(frozen=True)
class InitialStepsizeSearch:
    """reference InitialStepsizeSearch(; a_min=0.25, a_max=0.75, e0=1.0, C=2.0, maxiter_crossing=400, maxiter_bisect=400)"""
    a_min: float = 0.25
    a_max: float = 0.75
    eps0: float = 1.0
    C: float = 2.0
    maxiter_crossing: int = 400
    maxiter_bisect: int = 400

    def __len__(self):      # Base.length(::InitialStepsizeSearch) = 0, src/warmup.jl:201
        return 0


@dataclass(frozen=True)
class FindLocalOptimum:
    """reference FindLocalOptimum(magnitude_penalty=1e-4, iterations=50), src/warmup.jl:137-150: maximise
    l(q) - magnitude_penalty/2 sum(q^2) from the initial position; restarts and failure as :162-172."""
    magnitude_penalty: float = 1e-4
    iterations: int = 50

    def __len__(self):
        return 0


@dataclass(frozen=True)
class TuningNUTS:
    """reference TuningNUTS{M}(N, stepsize_adaptation, lambda = 5/N); M in {"Nothing", "Diagonal"}"""
    N: int
    stepsize_adaptation: object = field(default_factory=DualAveraging)
    M: str = "Diagonal"
    lam: Optional[float] = None

    def __post_init__(self):
        if self.M not in ("Nothing", "Diagonal"):
            raise ValueError("M must be 'Nothing' or 'Diagonal' (Symmetric metrics are not on the hot path)")
        if self.lam is not None and abs(self.lam - 5.0 / self.N) > 1e-15:
            raise ValueError("the engine uses the reference default lambda = 5/N")

    def __len__(self):
        return self.N


class NoProgressReport:
    """reference NoProgressReport (src/reporting.jl:6): reports nothing"""


class LogProgressReport:
    """reference LogProgressReport (src/reporting.jl:41-48): messages go to the `logging` framework (logger
    "InplaceDHMC", level INFO) -- the reference uses Julia's `@info`.  All chains of a device advance together, so a
    *step* is one NUTS transition of every chain; steps are reported at stage granularity (the driver loops run on
    the device without host synchronisation inside a stage)."""

    def __init__(self, chain_id=None, step_interval=100, time_interval_s=1000.0):
        self.chain_id, self.step_interval, self.time_interval_s = chain_id, step_interval, time_interval_s


def report(reporter, message, **meta):
    """reference report(reporter, message; meta...) (src/reporting.jl:22,63-66)"""
    if reporter is None or isinstance(reporter, NoProgressReport):
        return
    import logging
    if getattr(reporter, "chain_id", None) is not None:
        meta = dict(chain_id=reporter.chain_id, **meta)
    logging.getLogger("InplaceDHMC").info("%s%s", message, "".join("  %s = %s" % kv for kv in meta.items()))


def default_warmup_stages(local_optimization=FindLocalOptimum(), stepsize_search=InitialStepsizeSearch(),
                          M="Diagonal", stepsize_adaptation=DualAveraging(), init_steps=75, middle_steps=25,
                          doubling_stages=5, terminating_steps=50):
    """reference default_warmup_stages (src/warmup.jl:361-372)"""
    mid = tuple(TuningNUTS(middle_steps << d, stepsize_adaptation, M) for d in range(doubling_stages))
    return (local_optimization, stepsize_search, TuningNUTS(init_steps, stepsize_adaptation, "Nothing")) + mid + \
        (TuningNUTS(terminating_steps, stepsize_adaptation, "Nothing"),)


def fixed_stepsize_warmup_stages(local_optimization=FindLocalOptimum(), M="Diagonal", middle_steps=25,
                                 doubling_stages=5):
    """reference fixed_stepsize_warmup_stages (src/warmup.jl:383-389)"""
    return (local_optimization,) + tuple(TuningNUTS(middle_steps << d, FixedStepsize(), M) for d in range(doubling_stages))


@dataclass
class GaussianKineticEnergy:
    """reference GaussianKineticEnergy: diagonal M^-1 and W = 1/sqrt(M^-1) (src/hamiltonian.jl:33-74)"""
    minv: np.ndarray

    @staticmethod
    def identity(D, m_inv=1.0):
        return GaussianKineticEnergy(np.full(D, float(m_inv)))

    @property
    def W(self):
        return 1.0 / np.sqrt(self.minv)


# ---- running the stages on the engine -----------------------------------------------------------------
def _options_from(stages, algorithm, eps, eps_mode, metric_mode):
    tun = [s for s in stages if isinstance(s, TuningNUTS)]
    da = next((s.stepsize_adaptation for s in tun if isinstance(s.stepsize_adaptation, DualAveraging)), DualAveraging())
    ss = next((s for s in stages if isinstance(s, InitialStepsizeSearch)), None)
    kw = dict(max_depth=algorithm.max_depth, min_delta=algorithm.min_delta, da_delta=da.delta, da_gamma=da.gamma,
              da_kappa=da.kappa, da_t0=da.t0, eps_mode=eps_mode, metric_mode=metric_mode,
              stepsize_search=int(ss is not None and eps is None), eps_init=1.0 if eps is None else float(eps))
    if ss is not None:
        kw.update(ss_a_min=ss.a_min, ss_a_max=ss.a_max, ss_eps0=ss.eps0, ss_C=ss.C,
                  ss_maxiter_crossing=ss.maxiter_crossing, ss_maxiter_bisect=ss.maxiter_bisect)
    return _e.default_options(**kw)


def num_stored(N, stages):
    """NS = max(N, longest warmup stage) (reference src/mcmc.jl:115-116)"""
    return max([N] + [len(s) for s in stages if s is not None])


def run_stages(eng, N, stages, initialization, store_draws=True, reporter=None):
    """mcmc_with_warmup! (reference src/mcmc.jl:94-105): initialise, run the warmup stages, then mcmc!.
    Returns (chains [nchains, NS, D] or None, tree_statistics [nchains, NS], eps [nchains])."""
    C, D = eng.C, eng.D
    NS = num_stored(N, stages)
    chains = np.zeros((C, NS, D)) if store_draws else None
    stats = np.zeros((C, NS), dtype=_e.TREE_STATS_DTYPE)
    q0 = initialization.get("q")
    kap = initialization.get("kappa", initialization.get("κ"))
    eps0 = initialization.get("eps", initialization.get("ϵ"))
    if q0 is None:
        eng.random_position()                                  # src/warmup.jl:119
    else:
        q0 = np.asarray(q0, dtype=np.float64)
        eng.set_q(np.broadcast_to(q0, (C, D)))
    if kap is not None:
        eng.set_minv(kap.minv if isinstance(kap, GaussianKineticEnergy) else np.asarray(kap, dtype=np.float64))
    if eps0 is not None:
        eng.set_eps(eps0)
    it = 0
    for st in stages:
        if st is None:
            continue
        if isinstance(st, FindLocalOptimum):                   # src/warmup.jl:152-186
            report(reporter, "finding initial optimum")        # :161
            eng.find_local_optimum(st.magnitude_penalty, st.iterations)
            continue
        if isinstance(st, InitialStepsizeSearch):
            if eps0 is None:                                   # src/warmup.jl:188-200
                eng.refresh_momentum(0)
                eng.find_initial_stepsize()
                if reporter is not None and not isinstance(reporter, NoProgressReport):
                    report(reporter, "found initial stepsize", eps=float(np.median(eng.eps)))   # :197
            continue
        if isinstance(st, TuningNUTS):
            if isinstance(st.stepsize_adaptation, FixedStepsize):
                d, s = _fixed_stage(eng, st, it, store_draws)
            else:
                d, s = eng.tuning_stage(st.N, st.M == "Diagonal", it, store_draws=store_draws)
            it += st.N
            if reporter is not None and not isinstance(reporter, NoProgressReport):
                report(reporter, "warmup stage done", steps=st.N, metric=st.M, eps=float(np.median(eng.eps)),
                       mean_depth=float(s[-1]["depth"].mean()))
            if store_draws:
                chains[:, :st.N] = d.transpose(1, 0, 2)        # stage draws restart at column 0 (src/warmup.jl:280)
            stats[:, :st.N] = s.T
            continue
        raise TypeError("unknown warmup stage %r" % (st,))
    d, s = eng.mcmc(N, it, store_draws=store_draws)            # src/warmup.jl:316-332
    report(reporter, "mcmc done", steps=N)
    if store_draws and N:
        chains[:, :N] = d.transpose(1, 0, 2)
    if N:
        stats[:, :N] = s.T
    return chains, stats, eng.eps


def _fixed_stage(eng, st, it, store_draws):
    """TuningNUTS with FixedStepsize: metric adaptation only (src/stepsize.jl:251-259)."""
    draws = np.empty((st.N, eng.C, eng.D)) if store_draws else None
    stats = np.empty((st.N, eng.C), dtype=_e.TREE_STATS_DTYPE)
    adapt = st.M == "Diagonal"
    if adapt:
        eng.metric_begin()
    for n in range(st.N):
        eng.nuts_transition(it + 1 + n, _e.T_ACCUM_METRIC if adapt else 0)
        stats[n] = eng.tree_stats()
        if store_draws:
            draws[n] = eng.q
    if adapt:
        eng.metric_update(5.0 / st.N)
    return draws, stats


def threaded_mcmc(model, N, delta=0.8, initialization=None, warmup_stages=None, algorithm=NUTS(),
                  reporter=None, nchains=1, seed=1, device=0, first_chain=0, eps_mode=_e.EPS_PER_CHAIN,
                  metric_mode=_e.METRIC_PER_CHAIN, store_draws=True):
    """reference threaded_mcmc(l, N; d, initialization, warmup_stages, algorithm, reporter, nchains)
    (src/mcmc.jl:130-159): `nchains` independent chains, here one per wavefront instead of one per thread.
    Returns (chains, tree_statistics) with chains[c] of shape (NS, D) -- the reference's D x NS x nchains
    array read in C order -- and tree_statistics of shape (nchains, NS), NS = max(N, longest stage)."""
    initialization = dict(initialization or {})
    stages = warmup_stages if warmup_stages is not None else default_warmup_stages(
        stepsize_adaptation=DualAveraging(delta=delta))
    eps0 = initialization.get("eps", initialization.get("ϵ"))
    opt = _options_from(stages, algorithm, eps0, eps_mode, metric_mode)
    eng = _e.Engine(model, nchains, opt, seed=seed, first_chain=first_chain, device=device)
    try:
        chains, stats, _ = run_stages(eng, N, stages, initialization, store_draws=store_draws, reporter=reporter)
    finally:
        eng.close()
    return chains, stats


def mcmc_with_warmup(model, N, delta=0.8, initialization=None, warmup_stages=None, algorithm=NUTS(),
                     reporter=None, seed=1, device=0):
    """reference mcmc_with_warmup(l, N; ...) (src/mcmc.jl:109-128): one chain.
    Returns (chain [NS, D], tree_statistics [NS])."""
    chains, stats = threaded_mcmc(model, N, delta=delta, initialization=initialization,
                                  warmup_stages=warmup_stages, algorithm=algorithm, reporter=reporter,
                                  nchains=1, seed=seed, device=device)
    return chains[0], stats[0]
