"""Dimensions off the powers of two and beyond 1024.  A vector is padded to the next multiple of 128 (L = 128 ceil(D / 128),
every NCH = L / 128 in 1..16 has its own kernels; the reference pads to its SIMD width, src/mcmc.jl:117), so D = 1100 runs
at L = 1152, not 2048.  Same bit-exact bar against the oracle as everywhere else; the dense density is limited to 1024."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def pair(idhmc, oracle, kind, D, C, seed, **okw):
    if kind == "iso":
        gm, om = idhmc.IsoGaussian(D), oracle.OracleModel.iso(D)
        minv = np.ones(D)
    else:
        mu, sig = np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)
        gm, om = idhmc.DiagGaussian(mu, sigma=sig), oracle.OracleModel.diag(mu, 1.0 / sig ** 2)
        minv = sig ** 2
    eng = idhmc.Engine(gm, C, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED, **okw), seed=seed)
    eng.set_minv(minv)
    chains = [oracle.OracleChain(om, oracle.default_options(**okw), seed=seed, chain_id=c) for c in range(C)]
    for ch in chains:
        ch.set_minv(minv)
    return eng, chains


@pytest.mark.parametrize("kind,D", [("diag", 2048), ("diag", 1500), ("iso", 1025)])
def test_streaming_kernels(idhmc, oracle, kind, D):
    C = 5
    eng, chains = pair(idhmc, oracle, kind, D, C, seed=9)
    assert eng.lib.idhmc_padded_dim(eng.h) == (D + 127) // 128 * 128
    eng.random_position()
    eng.refresh_momentum(3)
    for ch in chains:
        ch.random_position()
        ch.rand_p(3)
    assert same_bits(eng.p, np.stack([c.p[:D] for c in chains])) and same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))
    eng.leapfrog(0.05, 1)
    eng.leapfrog(0.05, 3)
    eng.set_leapfrog_grad_mode(idhmc.GRAD_RECOMPUTE)
    eng.leapfrog(0.05, 1)
    eng.leapfrog(-0.05, 1)
    for ch in chains:
        for _ in range(5):
            ch.leapfrog(0.05)
        ch.leapfrog(-0.05)
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains])) and same_bits(eng.p, np.stack([c.p[:D] for c in chains]))
    assert same_bits(eng.lq, [c.lq for c in chains]) and same_bits(eng.logdensity(), [c.logdensity() for c in chains])
    assert same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()
    ref = []
    for ch in chains:
        ch.rand_p(0)
        rc, e = ch.find_initial_stepsize()
        assert rc == 0
        ref.append(e)
    assert same_bits(eng.eps, ref)


@pytest.mark.parametrize("kind,D,eps,L", [("diag", 2048, 0.05, 2048), ("iso", 1300, 0.2, 1408), ("diag", 1100, 0.05, 1152),
                                          ("diag", 700, 0.05, 768), ("iso", 600, 0.3, 640), ("diag", 300, 0.05, 384)])
def test_nuts_transitions(idhmc, oracle, kind, D, eps, L):
    C, T = 6, 8
    eng, chains = pair(idhmc, oracle, kind, D, C, seed=4, max_depth=7)
    assert eng.padded_dim() == L
    eng.random_position()
    eng.set_eps(eps)
    for ch in chains:
        ch.random_position()
    for it in range(1, T + 1):
        eng.nuts_transition(it)
        st = eng.tree_stats()
        ost = [ch.sample_tree(eps, it) for ch in chains]
        for f in ("depth", "steps", "term_left", "term_right"):
            np.testing.assert_array_equal(st[f], [getattr(s, f) for s in ost], err_msg="%s @%d" % (f, it))
        assert same_bits(st["pi"], [s.pi for s in ost]) and same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))


def test_warmup_with_the_pooled_metric_and_the_optimum_stage(idhmc):
    D, C = 2048, 32
    mu, sig = np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)
    short = dict(init_steps=25, middle_steps=15, doubling_stages=2, terminating_steps=15, local_opt_iterations=30)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C,
                       idhmc.default_options(metric_mode=idhmc.METRIC_POOLED, eps_mode=idhmc.EPS_GLOBAL, **short), seed=6)
    draws, stats = eng.mcmc_with_warmup(60)
    x = draws.reshape(-1, D)
    assert np.isfinite(x).all() and 0.55 < stats["acceptance_rate"][-30:].mean() < 0.97
    assert np.median(np.abs(x.mean(axis=0) - mu) / sig) < 0.15 and np.median(x.var(axis=0) / sig ** 2) > 0.5
    eng.close()


def test_limits_are_errors(idhmc):
    with pytest.raises(idhmc.IdhmcError):
        idhmc.Engine(idhmc.IsoGaussian(2049), 4, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED))
    with pytest.raises(idhmc.IdhmcError) as e:
        idhmc.Engine(idhmc.DenseMVN(np.zeros(1100), np.eye(1100)), 2, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED))
    assert "limited to D <= 1024" in str(e.value)


def test_custom_density_beyond_1024(idhmc, oracle, tmp_path):
    """a hipRTC density at D = 1500 (NCH = 16 instantiations of the general kernels), shared metric: evaluation,
    leapfrog, stepsize search and NUTS transitions against the same density given to the oracle as C"""
    from test_gpu_custom import HIP_SRC, C_SRC, PARAMS
    D, C = 1500, 5
    opt = idhmc.default_options(max_depth=6, metric_mode=idhmc.METRIC_SHARED)
    eng = idhmc.Engine(idhmc.CustomDensity(D, HIP_SRC, PARAMS), C, opt, seed=31)
    om = oracle.OracleModel.custom(D, C_SRC, PARAMS, str(tmp_path))
    chains = [oracle.OracleChain(om, oracle.default_options(max_depth=6), seed=31, chain_id=c) for c in range(C)]
    eng.random_position()
    for ch in chains:
        ch.random_position()
    assert same_bits(eng.lq, [c.lq for c in chains]) and same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))
    eng.refresh_momentum(1)
    eng.leapfrog(0.01, 2)
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()
    eps = []
    for ch in chains:
        ch.rand_p(1)
        ch.leapfrog(0.01); ch.leapfrog(0.01)
        ch.rand_p(0)
        rc, e = ch.find_initial_stepsize()
        eps.append(e)
    assert same_bits(eng.eps, eps)
    eng.set_eps(0.02)
    for it in (1, 2, 3):
        eng.nuts_transition(it)
        ost = [ch.sample_tree(0.02, it) for ch in chains]
        np.testing.assert_array_equal(eng.tree_stats()["steps"], [s.steps for s in ost])
        assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))


def test_per_chain_metric_beyond_1024(idhmc, oracle):
    """reference semantics (every chain adapts its own stepsize and metric) at D = 1100: the NUTS kernel runs three
    wavefronts per workgroup there (LDS); the whole shortened warm-up + draws equal the oracle bit for bit"""
    D, C, N = 1100, 4, 8
    mu, sig = np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)
    short = dict(init_steps=10, middle_steps=8, doubling_stages=2, terminating_steps=6, max_depth=7)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(**short), seed=21)
    draws, stats = eng.mcmc_with_warmup(N)
    rc, och, ost, oeps = oracle.threaded_mcmc(oracle.OracleModel.diag(mu, 1.0 / sig ** 2), N, C,
                                              oracle.default_options(**short), seed=21)
    assert rc == 0 and same_bits(eng.eps, oeps)
    for n in range(N):
        assert same_bits(draws[n], och[:, n, :D])
    assert np.array_equal(stats.T, ost[:, :N])
