"""Every padded length: one dimension per NCH = 1..16 (L = 128 NCH), alternating densities and metric modes, each against the
oracle bit for bit -- random position, momentum refresh, fused leapfrog (1 and 3 steps, both gradient modes), stepsize search,
NUTS transitions with adaptation flags, metric window.  The per-NCH kernels are separate instantiations; this is the test that
touches all of them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("nch", list(range(1, 17)))
def test_every_padded_length(idhmc, oracle, nch):
    O = oracle
    rng = np.random.default_rng(100 + nch)
    D = 128 * nch - int(rng.integers(0, 127))              # somewhere in (128 (nch-1), 128 nch]
    C, seed = 5, 40 + nch
    iso = nch % 3 == 0
    shared = nch % 2 == 0
    if iso:
        gm, om = idhmc.IsoGaussian(D), O.OracleModel.iso(D)
        minv = np.ones(D)
    else:
        mu, sig = np.cos(np.arange(D, dtype=np.float64)), np.logspace(-0.7, 0.7, D)
        gm, om = idhmc.DiagGaussian(mu, sigma=sig), O.OracleModel.diag(mu, 1.0 / sig ** 2)
        minv = sig ** 2 * rng.uniform(0.8, 1.25, D)
    kw = dict(max_depth=6)
    eng = idhmc.Engine(gm, C, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED if shared else idhmc.METRIC_PER_CHAIN, **kw), seed=seed)
    assert eng.padded_dim() == 128 * nch
    eng.set_minv(minv)
    chains = [O.OracleChain(om, O.default_options(**kw), seed=seed, chain_id=c) for c in range(C)]
    for ch in chains:
        ch.set_minv(minv)
        ch.random_position()
        ch.rand_p(2)
    eng.random_position()
    eng.refresh_momentum(2)
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains])) and same_bits(eng.p, np.stack([c.p[:D] for c in chains]))
    eps = 0.1 if iso else 0.04
    eng.leapfrog(eps, 1)
    eng.set_leapfrog_grad_mode(idhmc.GRAD_RECOMPUTE)
    eng.leapfrog(eps, 1)
    eng.set_leapfrog_grad_mode(idhmc.GRAD_STORE)
    eng.leapfrog(-eps, 3)
    for ch in chains:
        ch.leapfrog(eps); ch.leapfrog(eps)
        for _ in range(3):
            ch.leapfrog(-eps)
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains])) and same_bits(eng.p, np.stack([c.p[:D] for c in chains]))
    assert same_bits(eng.grad, np.stack([c.grad[:D] for c in chains])) and same_bits(eng.logdensity(), [c.logdensity() for c in chains])
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()
    e0 = []
    for ch in chains:
        ch.rand_p(0)
        rc, e = ch.find_initial_stepsize()
        assert rc == 0
        e0.append(e)
    assert same_bits(eng.eps, e0)
    eng.set_eps(eps)
    flags = 0 if shared else idhmc.T_ACCUM_METRIC
    if flags:
        eng.metric_begin()
    window = []
    for it in (1, 2, 3, 4):
        eng.nuts_transition(it, flags)
        window.append(eng.q)
        ost = [ch.sample_tree(eps, it) for ch in chains]
        st = eng.tree_stats()
        for f in ("depth", "steps", "term_left", "term_right"):
            assert st[f].tolist() == [getattr(s, f) for s in ost], (f, it)
        assert same_bits(st["pi"], [s.pi for s in ost]) and same_bits(st["acceptance_rate"], [s.acceptance_rate for s in ost])
        assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    if flags:                                              # the window's metric against the oracle's formula on the four draws
        eng.metric_update(5.0 / 4)
        got = eng.minv
        for c in range(C):
            draws = np.zeros((4, 128 * nch))
            draws[:, :D] = np.stack([w[c] for w in window])
            ref, _ = O.metric_from_draws(draws, D, 5.0 / 4)
            assert same_bits(got[c], ref[:D]), c
    eng.close()
