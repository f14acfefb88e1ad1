"""Oracle independence (the reference ships no test vectors, so parity cannot be pinned by it: test/runtests.jl:4-6).
Two restatements of the same Julia, written separately, are held against each other:
  oracle/idhmc_oracle.c   -- C, flat in-place arithmetic with the engine's own math library (what the HIP kernels match bit for bit)
  oracle/numpy_tree.py    -- numpy, recursive like src/tree.jl:321-366, libm / numpy arithmetic, no shared code
driven the way the reference's own test affordance allows (injected `p` and `directions`, src/NUTS.jl:251-252) with the
same exponential draws.  Tree decisions (depth, steps, termination) must be identical on every transition whose smallest
decision margin, as the C oracle reports it, exceeds 1e-9; draws and statistics agree to 1e-12 relative.
Plus hypothesis properties of a transition record (src/tree.jl:278-300, :382-444)."""
import itertools
import math

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import numpy_tree as NT


def _randexp_stream(O, seed, chain, it):
    L = O.lib()
    return (L.orc_randexp_export(seed, chain, it, k) for k in itertools.count())


def _momentum(O, seed, chain, it, Lp, D, w):
    import ctypes
    z = np.zeros(Lp)
    O.lib().orc_randn_export(seed, chain, it, Lp, z.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return w * z[:D]


@pytest.mark.parametrize("D,eps,max_depth,seed", [(10, 0.35, 6, 5), (24, 0.08, 8, 17), (3, 0.9, 5, 2), (50, 0.2, 7, 91)])
def test_two_restatements_build_the_same_trees(oracle, D, eps, max_depth, seed):
    O = oracle
    mu, sig = np.cos(np.arange(D, dtype=float)), np.logspace(-0.5, 0.5, D)
    tau, minv = 1.0 / sig ** 2, sig ** 2 * np.linspace(0.7, 1.3, D)
    om = O.OracleModel.diag(mu, tau)
    H = NT.Hamiltonian(NT.DiagGaussianDensity(mu, tau), minv)
    checked = skipped = 0
    for chain in range(6):
        ch = O.OracleChain(om, O.default_options(max_depth=max_depth), seed=seed, chain_id=chain)
        ch.set_minv(minv)
        ch.random_position()
        for it in range(1, 13):
            q0 = ch.q[:D].copy()
            dirs = O.lib().orc_rand_directions_export(seed, chain, it)
            p = _momentum(O, seed, chain, it, ch.L, D, 1.0 / np.sqrt(minv))
            st_c = ch.sample_tree(eps, it)                          # the C oracle draws the same p and directions itself
            q_np, st_np = NT.sample_tree(H, q0, p, eps, dirs, _randexp_stream(O, seed, chain, it), max_depth=max_depth)
            if ch.last_margin() < 1e-9:                             # a decision within rounding of flipping: not comparable
                skipped += 1
                ch.set_q(q_np)                                      # keep both on the same path
                continue
            checked += 1
            assert (st_c.depth, st_c.steps, st_c.term_left, st_c.term_right) == \
                   (st_np["depth"], st_np["steps"], st_np["term_left"], st_np["term_right"]), (chain, it)
            assert np.allclose(ch.q[:D], q_np, rtol=1e-12, atol=1e-12)
            assert abs(st_c.pi - st_np["pi"]) <= 1e-10 * max(1.0, abs(st_np["pi"]))
            assert abs(st_c.acceptance_rate - st_np["acceptance_rate"]) <= 1e-10
    assert checked >= 60 and skipped <= 6


@pytest.mark.parametrize("D,eps,max_depth,seed", [(6, 0.12, 6, 3), (17, 0.05, 7, 29)])
def test_two_restatements_dense_density(oracle, D, eps, max_depth, seed):
    """the general (non-separable) density path of the C oracle -- gradient carried, candidates stored -- against the numpy
    recursion with numpy's own matrix-vector product (a different summation order: 1e-10 on draws)"""
    O = oracle
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = (Q / np.logspace(-1.5, 0, D)) @ Q.T
    P = 0.5 * (P + P.T)
    mu = np.cos(np.arange(D, dtype=float))
    minv = np.linspace(0.02, 0.05, D)
    om = O.OracleModel.dense(mu, P)
    H = NT.Hamiltonian(NT.DenseGaussianDensity(mu, P), minv)
    checked = skipped = 0
    for chain in range(4):
        ch = O.OracleChain(om, O.default_options(max_depth=max_depth), seed=seed, chain_id=chain)
        ch.set_minv(minv)
        ch.random_position()
        for it in range(1, 11):
            q0 = ch.q[:D].copy()
            dirs = O.lib().orc_rand_directions_export(seed, chain, it)
            p = _momentum(O, seed, chain, it, ch.L, D, 1.0 / np.sqrt(minv))
            st_c = ch.sample_tree(eps, it)
            q_np, st_np = NT.sample_tree(H, q0, p, eps, dirs, _randexp_stream(O, seed, chain, it), max_depth=max_depth)
            if ch.last_margin() < 1e-8:
                skipped += 1
                ch.set_q(q_np)
                continue
            checked += 1
            assert (st_c.depth, st_c.steps, st_c.term_left, st_c.term_right) == \
                   (st_np["depth"], st_np["steps"], st_np["term_left"], st_np["term_right"]), (chain, it)
            assert np.allclose(ch.q[:D], q_np, rtol=1e-10, atol=1e-10)
            assert abs(st_c.acceptance_rate - st_np["acceptance_rate"]) <= 1e-8
    assert checked >= 30 and skipped <= 5


def test_injected_directions_and_divergence(oracle):
    """reference kwargs: fixed directions steer the doubling; a hopeless stepsize diverges at the first leaf (depth 0, proposal unchanged)"""
    O = oracle
    D = 6
    H = NT.Hamiltonian(NT.DiagGaussianDensity(np.zeros(D), np.ones(D)), np.ones(D))
    q0, p = np.full(D, 0.3), np.linspace(-1, 1, D)
    for dirs in (0b0, 0b111111, 0b101010):
        _, s = NT.sample_tree(H, q0, p, 0.05, dirs, iter(lambda: 1.0, None), max_depth=4)
        assert 1 <= s["depth"] <= 4 and s["steps"] >= 2 ** s["depth"] - 1
        om = O.OracleModel.iso(D)
        ch = O.OracleChain(om, O.default_options(max_depth=4), seed=1, chain_id=0)
        ch.set_q(q0)
        ch.set_p(p)
        sc = ch.sample_tree(0.05, 1, directions=dirs, refresh_p=False)
        assert (sc.depth, sc.steps, sc.term_left, sc.term_right) == (s["depth"], s["steps"], s["term_left"], s["term_right"])
    qd, s = NT.sample_tree(H, q0, p, 1e3, 0b1, iter(lambda: 1.0, None), max_depth=4)
    assert s["depth"] == 0 and s["steps"] == 1 and s["term_left"] == s["term_right"] == 1 and np.array_equal(qd, q0)


@settings(max_examples=150, deadline=None)
@given(seed=st.integers(1, 2 ** 31), D=st.integers(1, 40), eps=st.floats(0.01, 2.5), max_depth=st.integers(1, 8),
       scale=st.floats(0.2, 5.0))
def test_transition_record_properties(seed, D, eps, max_depth, scale):
    """src/tree.jl:278-300, :382-444: what a TreeStatisticsNUTS record can look like"""
    from oracle import oracle as O
    om = O.OracleModel.diag(np.zeros(D), np.full(D, 1.0 / scale ** 2))
    ch = O.OracleChain(om, O.default_options(max_depth=max_depth), seed=seed, chain_id=3)
    ch.random_position()
    for it in (1, 2, 3):
        s = ch.sample_tree(eps, it)
        L, R, d, n = s.term_left, s.term_right, s.depth, s.steps
        assert 0 <= d <= max_depth and 0.0 <= s.acceptance_rate <= 1.0 and np.isfinite(s.pi)
        full = 2 ** d - 1                                   # leapfrogs of the completed tree of this depth
        if (L, R) == (1, 0):                                # REACHED_MAX_DEPTH: every doubling valid, never turning
            assert d == max_depth and n == full
        elif L == R:                                        # divergence at node L of the failed doubling number d
            assert L != 0 and full < n <= 2 * full + 1 and 2 ** d <= abs(L) + (2 ** d - 1)
        elif L <= 0 <= R and R - L == full:                 # the whole tree turned after a successful doubling
            assert d >= 1 and n == full
        else:                                               # a sub-tree of the failed doubling turned: it lies on one side
            size = abs(R - L) + 1
            assert (L > 0) == (R > 0) and L != 0 and size >= 2 and size & (size - 1) == 0
            assert full < n <= 2 * full + 1


def test_whole_warmup_restated_in_numpy(oracle):
    """The adaptation around the tree -- initial stepsize search, dual averaging, regularised diagonal metric, stage sequence,
    sampling -- restated in numpy from src/stepsize.jl and src/warmup.jl (oracle/numpy_warmup.py), fed the same random
    numbers, against the C oracle's whole mcmc_with_warmup on a shortened schedule.  The two arithmetics differ in the last
    bits, and a warm-up is a chain of ~100 chaotic transitions: the comparison is made while the paths still coincide -- the
    stepsize search exactly, then every transition's tree record and draw (1e-9) up to the first one whose decision margin,
    as the C oracle reports it, falls below 1e-7 -- and the run must get through at least the first metric update."""
    import ctypes
    from oracle import numpy_warmup as NW
    O = oracle
    D, seed, chain = 12, 123, 2
    mu, sig = np.cos(np.arange(D, dtype=float)), np.logspace(-0.4, 0.4, D)
    tau = 1.0 / sig ** 2
    short = dict(init_steps=12, middle_steps=8, doubling_stages=2, terminating_steps=6, max_depth=6)
    om = O.OracleModel.diag(mu, tau)
    Lp = om.L

    def rng(it):
        z = np.zeros(Lp)
        O.lib().orc_randn_export(seed, chain, it, Lp, z.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return z[:D], O.lib().orc_rand_directions_export(seed, chain, it), _randexp_stream(O, seed, chain, it)

    # the C oracle, stage by stage through its fine-grained calls so that margins are visible per transition
    ch = O.OracleChain(om, O.default_options(**short), seed=seed, chain_id=chain)
    ch.random_position()
    q0 = ch.q[:D].copy()
    ch.rand_p(0)
    rc, eps_c = ch.find_initial_stepsize()
    assert rc == 0
    dens = NT.DiagGaussianDensity(mu, tau)
    H = NT.Hamiltonian(dens, np.ones(D))
    lq, g = H.evaluate(q0)
    eps_n = NW.find_initial_stepsize(NW.InitialStepsizeSearch(), NW.local_acceptance_ratio(H, NT.PhasePoint(q0, lq, g, rng(0)[0])))
    assert eps_n == eps_c                                           # bracketing / bisection: same decisions, same eps

    da = NW.DualAveraging()
    q_n, minv_n, it = q0.copy(), np.ones(D), 0
    stages = [(12, False), (8, True), (16, True), (6, False)]
    compared, metric_updates, diverged = 0, 0, False
    for n, adapt in stages:
        st_c = O.DAState()
        O.lib().orc_da_init(st_c, eps_c)
        st_n = da.initial_state(eps_n)
        chain_c, chain_n = [], []
        for _ in range(n):
            it += 1
            e_c, e_n = O.lib().orc_da_current_eps(st_c), math.exp(st_n[3])
            assert abs(e_c - e_n) <= 1e-9 * e_c
            rec_c = ch.sample_tree(e_c, it)
            z01, dirs, rexp = rng(it)
            q_n, rec_n = NT.sample_tree(NT.Hamiltonian(dens, minv_n), q_n, z01 / np.sqrt(minv_n), e_n, dirs, rexp, max_depth=6)
            if ch.last_margin() < 1e-7:
                diverged = True
                break
            assert (rec_c.depth, rec_c.steps, rec_c.term_left, rec_c.term_right) == \
                   (rec_n["depth"], rec_n["steps"], rec_n["term_left"], rec_n["term_right"]), it
            assert np.allclose(ch.q[:D], q_n, rtol=1e-9, atol=1e-9) and abs(rec_c.acceptance_rate - rec_n["acceptance_rate"]) < 1e-9
            compared += 1
            chain_c.append(ch.q.copy())
            chain_n.append(q_n.copy())
            O.lib().orc_da_adapt(ch.opt, st_c, rec_c.acceptance_rate)
            st_n = da.adapt(st_n, rec_n["acceptance_rate"])
        if diverged:
            break
        if adapt:
            mc, _ = O.metric_from_draws(np.stack(chain_c), D, 5.0 / n)
            minv_n = NW.regularized_metric(np.stack(chain_n), 5.0 / n)
            assert np.allclose(mc[:D], minv_n, rtol=1e-7, atol=0)
            ch.set_minv(mc[:D])
            metric_updates += 1
        eps_c, eps_n = O.lib().orc_da_final_eps(st_c), math.exp(st_n[4])
        assert abs(eps_c - eps_n) <= 1e-9 * eps_c
    assert compared >= 30 and metric_updates >= 1, (compared, metric_updates)      # this seed: all 42 transitions, both metric updates
