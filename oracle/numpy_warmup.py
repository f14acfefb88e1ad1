"""TEST INFRASTRUCTURE: the warm-up around the NUTS transition, restated independently in numpy from the Julia -- companion of
oracle/numpy_tree.py (same rules: plain numpy / libm arithmetic, no code shared with oracle/idhmc_oracle.c or the kernels, random
numbers injected by the caller).  Covers find_initial_stepsize (src/stepsize.jl:51-126,150-164), dual averaging (:208-241),
the regularised diagonal metric (src/hamiltonian.jl:77-101,156-158), the tuning stage (src/warmup.jl:269-314), mcmc!
(:316-332) and the default stage sequence (:341-372), i.e. SURVEY rows H13, H14, H18 and f1/f2.  Only tests/ may import it."""
import math

import numpy as np

from . import numpy_tree as NT


# ---- InitialStepsizeSearch, src/stepsize.jl:29-37 --------------------------------------------------------------------
class InitialStepsizeSearch:
    def __init__(self, a_min=0.25, a_max=0.75, eps0=1.0, C=2.0, maxiter_crossing=400, maxiter_bisect=400):
        self.a_min, self.a_max, self.eps0, self.C = a_min, a_max, eps0, C
        self.maxiter_crossing, self.maxiter_bisect = maxiter_crossing, maxiter_bisect


def find_crossing_stepsize(par, A, e0, A0):                 # src/stepsize.jl:51-72
    s, a = (1.0, par.a_max) if A0 > par.a_max else (-1.0, par.a_min)
    C = 1.0 / par.C if s < 0 else par.C
    for _ in range(par.maxiter_crossing):
        e = e0 * C
        Ae = A(e)
        if s * (Ae - a) <= 0:
            return e0, A0, e, Ae
        e0, A0 = e, Ae
    raise RuntimeError("Reached maximum number of iterations searching for eps")


def bisect_stepsize(par, A, e0, e1):                        # src/stepsize.jl:83-102
    for _ in range(par.maxiter_bisect):
        em = 0.5 * (e0 + e1)                                # middle(e0, e1)
        Am = A(em)
        if par.a_min <= Am <= par.a_max:
            return em
        if Am < par.a_min:
            e1 = em
        else:
            e0 = em
    raise RuntimeError("Reached maximum number of iterations while bisecting")


def find_initial_stepsize(par, A):                          # src/stepsize.jl:111-126
    A0 = A(par.eps0)
    if par.a_min <= A0 <= par.a_max:
        return par.eps0
    e0, A0, e1, A1 = find_crossing_stepsize(par, A, par.eps0, A0)
    if par.a_min <= A1 <= par.a_max:
        return e1
    return bisect_stepsize(par, A, e0, e1) if e0 < e1 else bisect_stepsize(par, A, e1, e0)


def local_acceptance_ratio(H, z):                           # src/stepsize.jl:150-164
    target = H.logdensity(z)
    assert math.isfinite(target), "Starting point has non-finite density."
    return lambda eps: math.exp(H.logdensity(H.leapfrog(z, eps)) - target)


# ---- DualAveraging, src/stepsize.jl:173-241 -------------------------------------------------------------------------------
class DualAveraging:
    def __init__(self, delta=0.8, gamma=0.05, kappa=0.75, t0=10):
        self.delta, self.gamma, self.kappa, self.t0 = delta, gamma, kappa, t0

    def initial_state(self, eps):                           # :208-212
        le = math.log(eps)
        return [math.log(10) + le, 0, 0.0, le, 0.0]         # mu, m, Hbar, logeps, logeps_bar

    def adapt(self, st, a):                                 # :220-229
        mu, m, Hbar, le, lb = st
        m += 1
        Hbar += (self.delta - a - Hbar) / (m + self.t0)
        le = mu - math.sqrt(m) / self.gamma * Hbar
        lb += m ** (-self.kappa) * (le - lb)
        return [mu, m, Hbar, le, lb]


# ---- GaussianKineticEnergy!(kappa, chain, lambda), src/hamiltonian.jl:77-101,119-189 --------------------------------------
def regularized_metric(draws, lam):
    """draws: N x D.  Shift by the first draw, one pass (:86-93), then :94-97 with :156-158."""
    N = draws.shape[0]
    d = draws[1:] - draws[0]
    s1, s2 = d.sum(axis=0), (d * d).sum(axis=0)
    s2nm1 = s2 - (s1 * s1) * (1.0 / N)
    mulreg = N / ((N + lam) * (N - 1))
    addreg = 1e-3 * lam / (N + lam)
    minv = s2nm1 * mulreg + addreg
    return minv


# ---- the stages, src/warmup.jl:188-200, 269-332, 341-372; src/mcmc.jl:94-105 ----------------------------------------------
def tuning_stage(density, minv, q, eps, N, adapt_metric, rng, it0, da, max_depth, min_delta=-1000.0):
    """warmup!(TuningNUTS): returns (q, minv, final eps, it, eps used, records).  rng(it) -> (p_unit_normals, directions, randexp iterator)"""
    H = NT.Hamiltonian(density, minv)
    st = da.initial_state(eps)
    chain, used, recs = [], [], []
    it = it0
    for _ in range(N):
        e = math.exp(st[3])                                  # current_eps :235
        assert e >= 1e-10
        used.append(e)
        it += 1
        z01, dirs, rexp = rng(it)
        q, rec = NT.sample_tree(H, q, z01 / np.sqrt(minv), e, dirs, rexp, max_depth=max_depth, min_delta=min_delta)   # rand_p!: W .* randn
        chain.append(q)
        recs.append(rec)
        st = da.adapt(st, rec["acceptance_rate"])
    if adapt_metric:
        minv = regularized_metric(np.stack(chain), 5.0 / N)  # lambda = 5/N, src/warmup.jl:229
    return q, minv, math.exp(st[4]), it, used, recs          # final_eps :241


def mcmc_with_warmup(density, D, q0, rng, N, init_steps=75, middle_steps=25, doubling_stages=5, terminating_steps=50,
                     max_depth=10, search=None, da=None):
    """mcmc_with_warmup! without the local-optimum stage: initial state kappa = I (:102), stepsize search (:188-200),
    default_warmup_stages (:361-372), mcmc! (:316-332).  Returns (draws N x D, records, eps, minv)."""
    search, da = search or InitialStepsizeSearch(), da or DualAveraging()
    minv = np.ones(D)
    q = np.asarray(q0, float)
    H = NT.Hamiltonian(density, minv)
    z01, _, _ = rng(0)                                       # rand_p! of the search stage (transition number 0)
    lq, g = H.evaluate(q)
    eps = find_initial_stepsize(search, local_acceptance_ratio(H, NT.PhasePoint(q, lq, g, z01 / np.sqrt(minv))))
    it = 0
    stages = [(init_steps, False)] + [(middle_steps << d, True) for d in range(doubling_stages)] + [(terminating_steps, False)]
    for n, adapt in stages:
        q, minv, eps, it, _, _ = tuning_stage(density, minv, q, eps, n, adapt, rng, it, da, max_depth)
    H = NT.Hamiltonian(density, minv)
    draws, recs = [], []
    for _ in range(N):
        it += 1
        z01, dirs, rexp = rng(it)
        q, rec = NT.sample_tree(H, q, z01 / np.sqrt(minv), eps, dirs, rexp, max_depth=max_depth)
        draws.append(q)
        recs.append(rec)
    return np.stack(draws), recs, eps, minv
