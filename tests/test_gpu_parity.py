"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerance: BIT-EXACT fp64.  Both sides use the same counter-based RNG, the same +,-,*,/,sqrt,fma-only
elementary functions and the same reduction order (oracle/orc_math.h, csrc/idhmc_math.hpp), so every
comparison below is `==` on the raw doubles unless stated otherwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def diag_model(pkg, O, D, seed=0):
    sig = np.logspace(-1, 1, D)
    mu = np.sin(np.arange(D, dtype=np.float64))
    return pkg.DiagGaussian(mu, sigma=sig), O.OracleModel.diag(mu, 1.0 / sig ** 2), mu, sig


def make_pair(pkg, O, kind, D, C, seed, **optkw):
    if kind == "iso":
        gm, om = pkg.IsoGaussian(D), O.OracleModel.iso(D)
    else:
        gm, om, _, _ = diag_model(pkg, O, D)
    gopt = pkg.default_options(**optkw)
    oopt = O.default_options(**{k: v for k, v in optkw.items() if k not in ("eps_mode", "metric_mode")})
    eng = pkg.Engine(gm, C, gopt, seed=seed)
    chains = [O.OracleChain(om, oopt, seed=seed, chain_id=c) for c in range(C)]
    return eng, chains


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bits_equal(a, b, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)[:5]
        raise AssertionError("%s: %d of %d values differ, first at %s: gpu=%r oracle=%r" % (
            what, (~same).sum(), same.size, idx.tolist(), a[tuple(idx[0])], b[tuple(idx[0])]))


@pytest.mark.parametrize("kind,D", [("iso", 32), ("diag", 100), ("diag", 256), ("diag", 1024), ("iso", 1000)])
def test_random_position_and_eval(idhmc, oracle, kind, D):
    C = 7
    eng, chains = make_pair(idhmc, oracle, kind, D, C, seed=11)
    eng.random_position()
    for ch in chains:
        ch.random_position()
    assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q")
    assert_bits_equal(eng.grad, np.stack([c.grad[:D] for c in chains]), "grad")
    assert_bits_equal(eng.lq, np.array([c.lq for c in chains]), "lq")
    q = eng.q
    assert q.min() >= -2.0 and q.max() < 2.0


@pytest.mark.parametrize("kind,D", [("iso", 32), ("diag", 200), ("diag", 1024)])
def test_momentum_refresh(idhmc, oracle, kind, D):
    C = 5
    eng, chains = make_pair(idhmc, oracle, kind, D, C, seed=3)
    minv = np.linspace(0.5, 2.0, D)
    eng.set_minv(minv)
    eng.random_position()
    eng.refresh_momentum(17)
    for ch in chains:
        ch.set_minv(minv)
        ch.random_position()
        ch.rand_p(17)
    assert_bits_equal(eng.p, np.stack([c.p[:D] for c in chains]), "p")
    assert_bits_equal(eng.logdensity(), np.array([c.logdensity() for c in chains]), "pi")


@pytest.mark.parametrize("kind,D,nsteps", [("iso", 32, 1), ("diag", 100, 1), ("diag", 1024, 1), ("diag", 1024, 5),
                                           ("diag", 512, 3), ("iso", 1024, 1)])
def test_leapfrog(idhmc, oracle, kind, D, nsteps):
    C = 9
    eng, chains = make_pair(idhmc, oracle, kind, D, C, seed=5)
    minv = np.logspace(-1, 1, D) ** 2 if kind == "diag" else np.ones(D)
    eng.set_minv(minv)
    eng.random_position()
    eng.refresh_momentum(1)
    eps = 0.05
    eng.leapfrog(eps, nsteps)
    eng.leapfrog(-eps, 1)
    for ch in chains:
        ch.set_minv(minv)
        ch.random_position()
        ch.rand_p(1)
        for _ in range(nsteps):
            ch.leapfrog(eps)
        ch.leapfrog(-eps)
    assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q")
    assert_bits_equal(eng.p, np.stack([c.p[:D] for c in chains]), "p")
    assert_bits_equal(eng.grad, np.stack([c.grad[:D] for c in chains]), "grad")
    assert_bits_equal(eng.lq, np.array([c.lq for c in chains]), "lq")
    assert_bits_equal(eng.logdensity(), np.array([c.logdensity() for c in chains]), "pi")


@pytest.mark.parametrize("kind,D,eps,md", [("iso", 32, 0.3, 5), ("diag", 100, 0.05, 10), ("diag", 1024, 0.02, 8),
                                           ("iso", 256, 0.2, 10), ("iso", 32, 1.9, 6), ("diag", 500, 0.03, 9)])
def test_nuts_transitions(idhmc, oracle, kind, D, eps, md):
    """Every transition of every chain: identical tree (depth, steps, termination), identical draw."""
    C, T = 12, 25
    eng, chains = make_pair(idhmc, oracle, kind, D, C, seed=2026, max_depth=md)
    eng.random_position()
    eng.set_eps(eps)
    for ch in chains:
        ch.random_position()
    for it in range(1, T + 1):
        eng.nuts_transition(it)
        gst = eng.tree_stats()
        ost = [ch.sample_tree(eps, it) for ch in chains]
        for f in ("depth", "steps", "term_left", "term_right"):
            np.testing.assert_array_equal(gst[f], np.array([getattr(s, f) for s in ost]), err_msg="%s at transition %d" % (f, it))
        assert_bits_equal(gst["pi"], np.array([s.pi for s in ost]), "stats.pi @%d" % it)
        assert_bits_equal(gst["acceptance_rate"], np.array([s.acceptance_rate for s in ost]), "stats.a @%d" % it)
        assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q @%d" % it)
        assert_bits_equal(eng.lq, np.array([c.lq for c in chains]), "lq @%d" % it)
    assert_bits_equal(eng.grad, np.stack([c.grad[:D] for c in chains]), "grad")
    assert eng.total_steps() == 0 or eng.total_steps() > 0


def test_max_depth_13_trees(idhmc, oracle):
    """doublings of up to 4096 leaves (64 blocks of 64 leaves under six levels of the reference's own cascade), max_depth = 13"""
    D, C, T, eps = 12, 6, 3, 0.0007
    eng, chains = make_pair(idhmc, oracle, "iso", D, C, seed=5, max_depth=13)
    eng.random_position()
    eng.set_eps(eps)
    for ch in chains:
        ch.random_position()
    for it in range(1, T + 1):
        eng.nuts_transition(it)
        gst = eng.tree_stats()
        ost = [ch.sample_tree(eps, it) for ch in chains]
        for f in ("depth", "steps", "term_left", "term_right"):
            np.testing.assert_array_equal(gst[f], np.array([getattr(s, f) for s in ost]), err_msg="%s at transition %d" % (f, it))
        assert_bits_equal(gst["acceptance_rate"], np.array([s.acceptance_rate for s in ost]), "stats.a @%d" % it)
        assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q @%d" % it)
    assert gst["depth"].max() >= 11


@pytest.mark.parametrize("kind,D,eps", [("diag", 40, 0.004), ("iso", 24, 0.012), ("diag", 130, 0.006)])
def test_long_doublings_stop_at_every_level(idhmc, oracle, kind, D, eps):
    """Round 3 evaluates the tree's bookkeeping after the tree (nuts_replay): the doublings of up to 32 leaves in one set of passes,
    longer ones in 64-leaf blocks with the reference's own merge above them.  Small stepsizes and max_depth = 10 give trees of 128 to
    1023 leaves that end in every way the reference knows -- whole-tree turn, a turning sub-tree at a low level (inside a block), at a high
    one (between blocks), max depth -- and multinomial picks that draw across block boundaries: records and draws must equal the
    oracle's recursion bit for bit (src/tree.jl:321-444, src/NUTS.jl:32-45, 68-84)."""
    C, T = 24, 8
    eng, chains = make_pair(idhmc, oracle, kind, D, C, seed=77, max_depth=10)
    eng.random_position()
    eng.set_eps(eps)
    for ch in chains:
        ch.random_position()
    seen_depth, seen_sub_turn_leaves = set(), set()
    for it in range(1, T + 1):
        eng.nuts_transition(it)
        gst = eng.tree_stats()
        ost = [ch.sample_tree(eps, it) for ch in chains]
        for f in ("depth", "steps", "term_left", "term_right"):
            np.testing.assert_array_equal(gst[f], np.array([getattr(s, f) for s in ost]), err_msg="%s at transition %d" % (f, it))
        assert_bits_equal(gst["pi"], np.array([s.pi for s in ost]), "stats.pi @%d" % it)
        assert_bits_equal(gst["acceptance_rate"], np.array([s.acceptance_rate for s in ost]), "stats.a @%d" % it)
        assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q @%d" % it)
        assert_bits_equal(eng.lq, np.array([c.lq for c in chains]), "lq @%d" % it)
        seen_depth.update(int(d) for d in gst["depth"])
        full = gst["steps"] == (1 << gst["depth"]) - 1
        seen_sub_turn_leaves.update(int(n) for n in (gst["steps"][~full] - ((1 << gst["depth"][~full]) - 1)))   # leaves of the doubling that stopped
    assert max(seen_depth) >= 8                                     # doublings of 64 and more leaves were merged
    assert any(n >= 64 for n in seen_sub_turn_leaves)               # ... and one stopped between its 64-leaf blocks


@pytest.mark.parametrize("D", [1024, 700])
@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("wide", ["0", "1"])
def test_both_forms_of_the_l1024_kernel(idhmc, oracle, monkeypatch, wide, shared, D):
    """512 < L <= 1024 has two forms of k_nuts: two wavefronts per SIMD (the default: level-1 rho in LDS with a shared
    metric, whole-tree rho parked in that slot between doublings) and one per SIMD (level-2 summary and whole-tree rho on
    chip); IDHMC_NUTS_WIDE pins one.  Same arithmetic: both equal the oracle bit for bit, with a per-chain and a shared
    metric, deep enough trees (depth 7) for level >= 3 merges, the regeneration checkpoints and every park path."""
    monkeypatch.setenv("IDHMC_NUTS_WIDE", wide)
    C, T = 9, 6
    kw = dict(metric_mode=idhmc.METRIC_SHARED) if shared else {}
    eng, chains = make_pair(idhmc, oracle, "diag", D, C, seed=11, max_depth=7, **kw)
    eng.random_position()
    eng.set_eps(0.04)
    for ch in chains:
        ch.random_position()
    depths = []
    for it in range(1, T + 1):
        eng.nuts_transition(it)
        st = eng.tree_stats()
        ost = [ch.sample_tree(0.04, it) for ch in chains]
        np.testing.assert_array_equal(st["steps"], np.array([s.steps for s in ost]))
        assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q @%d" % it)
        depths.append(st["depth"].max())
    assert max(depths) >= 6
    assert_bits_equal(eng.grad, np.stack([c.grad[:D] for c in chains]), "grad")


@pytest.mark.parametrize("kind,D", [("iso", 32), ("diag", 100), ("diag", 1024)])
def test_leapfrog_gradient_recompute_mode(idhmc, oracle, kind, D):
    """IDHMC_GRAD_RECOMPUTE: the single-step kernel neither reads nor writes the gradient array (4 of the 6 streams);
    q, p, l, pi after any number of steps, and the gradient whenever it is asked for, are bit-identical to the
    store mode and the oracle; the calls that need the array (stepsize search, n-step kernel, get_grad) see it fresh"""
    C = 7
    eng, chains = make_pair(idhmc, oracle, kind, D, C, seed=42)
    eng.set_leapfrog_grad_mode(idhmc.GRAD_RECOMPUTE)
    eng.random_position()
    eng.refresh_momentum(1)
    for ch in chains:
        ch.random_position()
        ch.rand_p(1)
    for _ in range(4):
        eng.leapfrog(0.05, 1)
        for ch in chains:
            ch.leapfrog(0.05)
    assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q")
    assert_bits_equal(eng.p, np.stack([c.p[:D] for c in chains]), "p")
    assert_bits_equal(eng.lq, np.array([c.lq for c in chains]), "lq")
    assert_bits_equal(eng.logdensity(), np.array([c.logdensity() for c in chains]), "pi")
    assert_bits_equal(eng.grad, np.stack([c.grad[:D] for c in chains]), "grad on demand")
    eng.leapfrog(0.05, 1)                       # stale again ...
    eng.leapfrog(0.05, 3)                       # ... and the n-step kernel (which reads the array) still agrees
    for ch in chains:
        for _ in range(4):
            ch.leapfrog(0.05)
    assert_bits_equal(eng.q, np.stack([c.q[:D] for c in chains]), "q after the n-step kernel")
    eng.leapfrog(0.05, 1)
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()
    ref = []
    for ch in chains:
        ch.leapfrog(0.05)
        ch.rand_p(0)
        rc, e = ch.find_initial_stepsize()
        ref.append(e)
    assert_bits_equal(eng.eps, np.array(ref), "stepsize search after stale gradient")
