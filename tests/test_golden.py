"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them bit for bit.  GPU (-m gpu): the HIP path reproduces them through
the C ABI -- bit-exact fp64, no oracle involved on the GPU box."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def diag_params(D):
    return np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


# ---- CPU: oracle vs golden -------------------------------------------------------------------------
def test_oracle_rng_golden(oracle):
    g = load("rng.npz")
    L = oracle.lib()
    z = np.zeros(256)
    L.orc_randn_export(int(g["seed"]), int(g["chain"]), int(g["iter"]), 256, oracle._dp(z))
    assert same_bits(z, g["randn"])
    assert same_bits([L.orc_randexp_export(int(g["seed"]), 3, 5, i) for i in range(16)], g["randexp"])
    assert [L.orc_rand_directions_export(int(g["seed"]), c, 5) for c in range(16)] == g["directions"].tolist()


@pytest.mark.parametrize("name,kind,D", [("transitions_iso32.npz", "iso", 32), ("transitions_diag100.npz", "diag", 100),
                                         ("transitions_diag40_long.npz", "diag_identity_metric", 40)])
def test_oracle_transitions_golden(oracle, name, kind, D):
    g = load(name)
    T, C = g["q"].shape[:2]
    if kind == "iso":
        m, minv = oracle.OracleModel.iso(D), None
    else:
        mu, sig = diag_params(D)
        m, minv = oracle.OracleModel.diag(mu, 1 / sig ** 2), (sig ** 2 if kind == "diag" else None)
    opt = oracle.default_options(max_depth=int(g["max_depth"]))
    for c in range(C):
        ch = oracle.OracleChain(m, opt, seed=int(g["seed"]), chain_id=c)
        if minv is not None:
            ch.set_minv(minv)
        ch.random_position()
        for t in range(T):
            st = ch.sample_tree(float(g["eps"]), t + 1)
            assert same_bits(ch.q[:D], g["q"][t, c])
            assert (st.depth, st.steps, st.term_left, st.term_right) == tuple(int(g["stats"][t, c][f]) for f in ("depth", "steps", "term_left", "term_right"))
            assert st.pi == g["stats"][t, c]["pi"] and st.acceptance_rate == g["stats"][t, c]["acceptance_rate"]


def test_oracle_cfg1_golden(oracle):
    g = load("cfg1_iso32.npz")
    rc, chains, stats, eps = oracle.threaded_mcmc(oracle.OracleModel.iso(32), int(g["N"]), 4,
                                                  oracle.default_options(max_depth=int(g["max_depth"])), seed=int(g["seed"]))
    assert rc == 0 and same_bits(eps, g["eps"]) and same_bits(chains[:, 99, :32], g["last_draw"])
    assert np.array_equal(stats[:, :100], g["stats"])


# ---- GPU: HIP path vs golden -----------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_leapfrog_golden(idhmc):
    g = load("leapfrog_diag1024.npz")
    mu, sig = diag_params(1024)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), 3, seed=int(g["seed"]))
    eng.set_minv(sig ** 2)
    eng.random_position()
    eng.refresh_momentum(1)
    eng.leapfrog(float(g["eps"]), 2)
    eng.leapfrog(float(g["eps"]), 1)
    assert same_bits(eng.q, g["q"]) and same_bits(eng.p, g["p"]) and same_bits(eng.lq, g["lq"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,D", [("transitions_iso32.npz", "iso", 32), ("transitions_diag100.npz", "diag", 100),
                                         ("transitions_diag40_long.npz", "diag_identity_metric", 40)])
def test_gpu_transitions_golden(idhmc, name, kind, D):
    g = load(name)
    T, C = g["q"].shape[:2]
    opt = idhmc.default_options(max_depth=int(g["max_depth"]))
    if kind == "iso":
        eng = idhmc.Engine(idhmc.IsoGaussian(D), C, opt, seed=int(g["seed"]))
    else:
        mu, sig = diag_params(D)
        eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, opt, seed=int(g["seed"]))
        if kind == "diag":
            eng.set_minv(sig ** 2)
    eng.random_position()
    eng.set_eps(float(g["eps"]))
    for t in range(T):
        eng.nuts_transition(t + 1)
        assert same_bits(eng.q, g["q"][t]), "draw %d" % t
        assert np.array_equal(eng.tree_stats(), g["stats"][t]), "stats %d" % t


@pytest.mark.gpu
def test_gpu_cfg1_golden(idhmc):
    """BASELINE.json configs[0] end to end on the device: 900 warmup transitions with per-chain dual
    averaging and five metric updates, then 100 draws -- still bit-identical to the golden run."""
    g = load("cfg1_iso32.npz")
    opt = idhmc.default_options(max_depth=int(g["max_depth"]))
    eng = idhmc.Engine(idhmc.IsoGaussian(32), 4, opt, seed=int(g["seed"]))
    draws, stats = eng.mcmc_with_warmup(int(g["N"]))
    assert same_bits(eng.eps, g["eps"])
    assert same_bits(draws[99], g["last_draw"])
    assert np.allclose(draws.sum(axis=0), g["draw_sum"], rtol=1e-12, atol=1e-12)   # numpy summation order differs by layout
    assert np.array_equal(stats.T, g["stats"])


def dense_problem(D, seed=7):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    lam = np.logspace(-2, 0, D)
    P = (Q / lam) @ Q.T
    return np.cos(np.arange(D, dtype=np.float64)), 0.5 * (P + P.T)


def test_oracle_optimum_golden(oracle):
    g = load("optimum_diag100.npz")
    mu, sig = diag_params(100)
    m = oracle.OracleModel.diag(mu, 1.0 / sig ** 2)
    for c in range(g["q"].shape[0]):
        ch = oracle.OracleChain(m, seed=int(g["seed"]), chain_id=c)
        ch.random_position()
        assert ch.find_local_optimum(float(g["penalty"]), int(g["iterations"])) == 0
        assert same_bits(ch.q[:100], g["q"][c]) and same_bits(ch.grad[:100], g["grad"][c]) and ch.lq == g["lq"][c]


@pytest.mark.gpu
def test_gpu_optimum_golden(idhmc):
    g = load("optimum_diag100.npz")
    mu, sig = diag_params(100)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), g["q"].shape[0], seed=int(g["seed"]))
    eng.random_position()
    eng.find_local_optimum(float(g["penalty"]), int(g["iterations"]))
    assert same_bits(eng.q, g["q"]) and same_bits(eng.grad, g["grad"]) and same_bits(eng.lq, g["lq"])


@pytest.mark.gpu
def test_gpu_dense_transitions_golden(idhmc):
    """dense MVN NUTS (workgroup-cooperative matrix-core gradient) against the committed oracle vectors"""
    g = load("transitions_dense64.npz")
    mu, P = dense_problem(64)
    T, C, D = g["q"].shape
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, idhmc.default_options(max_depth=int(g["max_depth"])), seed=int(g["seed"]))
    eng.random_position()
    eng.set_eps(float(g["eps"]))
    for t in range(T):
        eng.nuts_transition(t + 1)
        st = eng.tree_stats()
        assert same_bits(eng.q, g["q"][t])
        for f in ("depth", "steps", "term_left", "term_right"):
            assert np.array_equal(st[f], g["stats"][f][t])
        assert same_bits(st["pi"], g["stats"]["pi"][t])
