"""Known-answer tests that pin the CPU oracle (SURVEY.md 8c list): the reference has no tests
(test/runtests.jl:4-6), so the oracle is checked against analytic truth and independent numpy
restatements of the reference formulas, each citing the reference line it covers."""
import numpy as np
import pytest


def make_diag(O, D, seed=0):
    sig = np.logspace(-1, 1, D)
    mu = np.sin(np.arange(D, dtype=np.float64))
    return O.OracleModel.diag(mu, 1.0 / sig ** 2), mu, sig


def test_density_and_gradient(oracle):
    """logdensity_and_gradient! contract (src/kinetic_energy.jl:73) for the three built-ins"""
    rng = np.random.default_rng(0)
    D = 200
    q = rng.standard_normal(D)
    lq, g = oracle.OracleModel.iso(D).logdensity_and_gradient(q)
    assert abs(lq + 0.5 * q @ q) < 1e-12 and np.allclose(g, -q, rtol=0, atol=0)
    m, mu, sig = make_diag(oracle, D)
    lq, g = m.logdensity_and_gradient(q)
    assert abs(lq + 0.5 * np.sum((q - mu) ** 2 / sig ** 2)) < 1e-9 * abs(lq)
    assert np.allclose(g, -(q - mu) / sig ** 2, rtol=1e-15)
    A = rng.standard_normal((D, D))
    P = A @ A.T / D + np.eye(D)
    lq, g = oracle.OracleModel.dense(mu, P).logdensity_and_gradient(q)
    assert abs(lq + 0.5 * (q - mu) @ P @ (q - mu)) < 1e-10 * abs(lq)
    assert np.allclose(g, -P @ (q - mu), rtol=1e-12, atol=1e-12)


def test_leapfrog_closed_form(oracle):
    """leapfrog (src/kinetic_energy.jl:144-161) on N(mu, diag sigma^2): per coordinate the linear map
    q' = q + e m (p + e/2 g(q)),  p' = p + e/2 (g(q) + g(q'))"""
    D = 300
    m, mu, sig = make_diag(oracle, D)
    ch = oracle.OracleChain(m, seed=4)
    minv = np.linspace(0.5, 3.0, D)
    ch.set_minv(minv)
    ch.random_position()
    ch.rand_p(1)
    q, p = ch.q[:D].copy(), ch.p[:D].copy()
    eps = 0.037
    g = -(q - mu) / sig ** 2
    pm = p + 0.5 * eps * g
    q1 = q + eps * minv * pm
    g1 = -(q1 - mu) / sig ** 2
    p1 = pm + 0.5 * eps * g1
    ch.leapfrog(eps)
    assert np.allclose(ch.q[:D], q1, rtol=1e-14, atol=1e-14)
    assert np.allclose(ch.p[:D], p1, rtol=1e-13, atol=1e-13)
    assert np.allclose(ch.grad[:D], g1, rtol=1e-13, atol=1e-13)
    assert abs(ch.lq + 0.5 * np.sum((q1 - mu) ** 2 / sig ** 2)) < 1e-9 * abs(ch.lq)
    K = 0.5 * np.sum(p1 * minv * p1)
    assert abs(ch.logdensity() - (ch.lq - K)) < 1e-9 * abs(ch.lq)      # kinetic_energy :14-24, logdensity :107-112
    assert np.all(ch.q[D:] == 0) and np.all(ch.p[D:] == 0)


def test_leapfrog_reversible_and_second_order(oracle):
    D = 64
    ch = oracle.OracleChain(oracle.OracleModel.iso(D), seed=9)
    ch.random_position()
    ch.rand_p(1)
    q0, p0 = ch.q.copy(), ch.p.copy()
    ch.leapfrog(0.1)
    ch.leapfrog(-0.1)                         # backward step = negative eps, src/NUTS.jl:20
    assert np.abs(ch.q - q0).max() <= 1e-12 and np.abs(ch.p - p0).max() <= 1e-12
    errs = []
    for n in (16, 32, 64):                    # fixed integration time 1.6: energy error ~ eps^2
        ch.set_q(q0[:D]); ch.set_p(p0[:D])
        h0 = ch.logdensity()
        worst = 0.0
        for _ in range(n):
            ch.leapfrog(1.6 / n)
            worst = max(worst, abs(ch.logdensity() - h0))
        errs.append(worst)
    assert 3.0 < errs[0] / errs[1] < 5.0 and 3.0 < errs[1] / errs[2] < 5.0


def test_nonfinite_handling(oracle):
    """evaluate_l! (src/kinetic_energy.jl:80-84) and logdensity(H,z) (:107-112): never an error"""
    D = 32
    ch = oracle.OracleChain(oracle.OracleModel.iso(D), seed=1)
    base = np.linspace(-1, 1, D)
    for bad in (np.inf, -np.inf, np.nan):
        q = base.copy(); q[3] = bad
        ch.set_q(q)
        assert ch.lq == -np.inf                      # l(q) in {-Inf, NaN} -> -Inf
        ch.set_p(np.ones(D))
        assert ch.logdensity() == -np.inf            # K is skipped
    ch.set_q(base)
    for bad in (np.inf, np.nan):
        p = np.ones(D); p[5] = bad
        ch.set_p(p)
        assert ch.logdensity() == -np.inf            # K in {Inf, NaN} -> pi = -Inf
    ch.set_p(np.ones(D))
    assert np.isfinite(ch.logdensity())


def test_logprob2_and_acceptance(oracle):
    L = oracle.lib()
    w1, w2 = -1.25, -0.5
    w = np.logaddexp(w1, w2)
    assert abs(L.orc_logprob2_export(0, w1, w2) - (w2 - w)) < 1e-15        # src/tree.jl:261-263, bias = false
    assert L.orc_logprob2_export(1, w1, w2) == w2 - w1                     # bias = true (doubling)
    assert L.orc_logprob2_export(1, -3.0, -1.0) > 0                        # > 0 is allowed and means "certain"
    assert L.orc_acceptance_rate_export(np.log(3.0), 4) == pytest.approx(0.75, abs=1e-15)   # src/NUTS.jl:84
    assert L.orc_acceptance_rate_export(np.log(9.0), 4) == 1.0
    assert L.orc_acceptance_rate_export(-np.inf, 1) == 0.0


def test_is_turning_signs(oracle):
    """src/NUTS.jl:148-170: turning iff either dot product is strictly negative"""
    L, n = oracle.lib(), 128
    e = lambda i, v=1.0: np.eye(n)[i] * v
    turning = lambda r, a, b: bool(L.orc_is_turning_export(oracle._dp(r), oracle._dp(a), oracle._dp(b), n))
    assert not turning(e(0), e(0), e(0))
    assert turning(e(0), e(0, -1.0), e(0))
    assert turning(e(0), e(0), e(0, -1.0))
    assert not turning(e(0), e(1), e(2))                 # both dots exactly 0: not turning
    assert turning(e(0) + e(1), e(1, -1e-300), e(0))


def hoffman_gelman(eps0, accepts, delta=0.8, gamma=0.05, kappa=0.75, t0=10):
    mu, m, Hbar, le, lb = np.log(10.0) + np.log(eps0), 0, 0.0, np.log(eps0), 0.0
    out = []
    for a in accepts:
        m += 1
        Hbar += (delta - a - Hbar) / (m + t0)
        le = mu - np.sqrt(m) / gamma * Hbar
        lb += m ** (-kappa) * (le - lb)
        out.append((Hbar, le, lb))
    return out


def test_dual_averaging_trace(oracle):
    """adapt_stepsize (src/stepsize.jl:220-229) against an independent numpy statement of Algorithm 6"""
    for eps0, accepts in ((0.7, [0.5] * 10), (0.1, [1.0, 0.9, 0.2, 0.85, 0.0, 0.6, 0.99, 0.75, 0.8, 0.81])):
        got = oracle.da_trace(eps0, accepts)
        ref = hoffman_gelman(eps0, accepts)
        for (mu, m, Hbar, le, lb, cur, fin), (rH, rle, rlb) in zip(got, ref):
            assert abs(Hbar - rH) < 1e-14 and abs(le - rle) < 1e-12 and abs(lb - rlb) < 1e-12
            assert abs(cur - np.exp(rle)) < 1e-12 * np.exp(rle) and abs(fin - np.exp(rlb)) < 1e-12 * np.exp(rlb)
        assert got[-1][1] == len(accepts) and abs(got[0][0] - (np.log(10) + np.log(eps0))) < 1e-14
    # constant acceptance above target drives eps up, below target down
    assert oracle.da_trace(0.5, [1.0] * 20)[-1][5] > 0.5 > oracle.da_trace(0.5, [0.0] * 20)[-1][5]


def test_regularised_metric(oracle):
    """GaussianKineticEnergy! (src/hamiltonian.jl:119-189): N/(N+l) s^2 + 1e-3 l/(N+l), W = 1/sqrt"""
    rng = np.random.default_rng(3)
    for N in (25, 50, 400):
        D, L = 70, 128
        draws = np.zeros((N, L))
        draws[:, :D] = rng.standard_normal((N, D)) * np.linspace(0.1, 5, D) + 100.0
        lam = 5.0 / N
        minv, w = oracle.metric_from_draws(draws, D, lam)
        ref = np.var(draws[:, :D], axis=0, ddof=1) * N / (N + lam) + 1e-3 * lam / (N + lam)
        assert np.allclose(minv[:D], ref, rtol=1e-10)
        assert np.allclose(w[:D], 1 / np.sqrt(ref), rtol=1e-10)
        assert np.all(minv[D:] == 1.0) and np.all(w[D:] == 1.0)


def test_directions_lsb_first_and_complete_tree(oracle):
    """Directions are consumed LSB first (src/tree.jl:152-155); a completed valid tree has
    2^depth - 1 leapfrog steps; the termination code is InvalidTree(i-, i+) (src/tree.jl:438)"""
    D = 16
    for dirs, check in ((0xffffffff, lambda l, r: l == 0 and r > 0), (0x0, lambda l, r: r == 0 and l < 0),
                        (0b0101, None)):
        ch = oracle.OracleChain(oracle.OracleModel.iso(D), oracle.default_options(max_depth=8), seed=3)
        ch.random_position()
        st = ch.sample_tree(0.2, 1, directions=dirs)
        full_turn = (st.term_left < st.term_right) and (st.term_right - st.term_left == (1 << st.depth) - 1)
        if full_turn or (st.term_left, st.term_right) == (1, 0):
            assert st.steps == (1 << st.depth) - 1
        if check is not None and full_turn:
            assert check(st.term_left, st.term_right)
        if dirs == 0b0101 and full_turn and st.depth >= 3:
            # fwd(1), bwd(2), fwd(4): i+ = 1 + 4 = 5, i- = -2 after three doublings
            assert (st.term_left, st.term_right)[0] <= -2


def test_divergence_at_first_leaf(oracle):
    """Delta < min_Delta at the first leaf: termination = (+-1, +-1), depth 0, proposal unchanged
    (src/NUTS.jl:179-186, src/tree.jl:332, 417)"""
    m, mu, sig = make_diag(oracle, 64)
    ch = oracle.OracleChain(m, seed=8)
    ch.random_position()
    q0, lq0 = ch.q.copy(), ch.lq
    st = ch.sample_tree(1e3, 1)
    assert st.depth == 0 and st.steps == 1 and st.term_left == st.term_right and abs(st.term_left) == 1
    assert np.array_equal(ch.q, q0) and ch.lq == lq0 and st.acceptance_rate == 0.0
    fwd = oracle.lib().orc_rand_directions_export(8, 0, 1) & 1
    assert st.term_left == (1 if fwd else -1)


def test_harmonic_oscillator_turns_after_half_period(oracle):
    """1-D standard normal: the trajectory is a rotation, U-turn after ~pi/eps leapfrogs"""
    ch = oracle.OracleChain(oracle.OracleModel.iso(1), oracle.default_options(max_depth=12), seed=1)
    eps = 0.01
    ch.set_q([0.0]); ch.set_p([1.0])
    st = ch.sample_tree(eps, 1, directions=0xffffffff, refresh_p=False)
    assert st.term_left < st.term_right                       # turning, not divergence / max depth
    assert np.pi / eps / 2 <= st.steps <= 2 * np.pi / eps * 1.05
    assert st.acceptance_rate > 0.99


def test_no_draw_when_logprob_nonnegative(oracle):
    """rand_bool_logprob (src/NUTS.jl:32-34): with the first doubling's w' >= w the new point is taken
    without consuming a random number -- the draw counter address of later draws is unchanged."""
    D = 8
    ch = oracle.OracleChain(oracle.OracleModel.iso(D), oracle.default_options(max_depth=1), seed=77)
    taken = 0
    for it in range(1, 200):
        ch.set_q(np.full(D, 1.5))
        st = ch.sample_tree(0.3, it)
        moved = not np.allclose(ch.q[:D], 1.5)
        # pi(z') >= pi(z) means logprob2 >= 0 under biased progressive sampling: must always move
        if st.depth == 1 and st.pi >= -0.5 * D * 1.5 ** 2 - 1e300 and moved:
            taken += 1
    assert taken > 0


def test_cfg1_plumbing(oracle):
    """BASELINE.json configs[0]: 32-dim isotropic Gaussian, 4 chains, NUTS max_depth=5, default warmup
    (75/25/50/100/200/400/50 = 900 transitions) + 1000 draws; output shapes as src/mcmc.jl:115-119"""
    m = oracle.OracleModel.iso(32)
    opt = oracle.default_options(max_depth=5)
    rc, chains, stats, eps = oracle.threaded_mcmc(m, 1000, 4, opt, seed=20261004)
    assert rc == 0 and chains.shape == (4, 1000, 128) and stats.shape == (4, 1000)
    assert np.all((eps > 0.3) & (eps < 1.5))
    x = chains[:, :, :32]
    assert np.abs(x.mean(axis=(0, 1))).max() < 0.1 and np.abs(x.var(axis=(0, 1)) - 1).max() < 0.15
    assert 0.7 < stats["acceptance_rate"].mean() < 0.95
    assert stats["depth"].max() <= 5 and (stats["steps"] <= 31).all()
    rc, chains, stats, eps = oracle.threaded_mcmc(m, 100, 2, opt, seed=1)
    assert chains.shape == (2, 400, 128) and stats.shape == (2, 400)      # NS = max(N, 400)
    assert np.all(chains[:, :, 32:] == 0)


def test_threaded_is_deterministic_and_sharding_invariant(oracle):
    m, _, _ = make_diag(oracle, 40)
    opt = oracle.default_options(max_depth=6, init_steps=20, middle_steps=10, doubling_stages=2, terminating_steps=10)
    _, a, sa, ea = oracle.threaded_mcmc(m, 30, 6, opt, seed=5, nthreads=3)
    _, b, sb, eb = oracle.threaded_mcmc(m, 30, 3, opt, seed=5, first_chain=3, nthreads=1)
    assert np.array_equal(a[3:], b) and np.array_equal(ea[3:], eb) and np.array_equal(sa[3:], sb)


def test_local_optimum_stage(oracle):
    """FindLocalOptimum contract (src/warmup.jl:137-187) with the engine's own L-BFGS: reaches the mode of a
    well-conditioned Gaussian, climbs on an ill-conditioned one, leaves (q, lq, grad) consistent, and is a
    no-op at iterations = 0"""
    D = 32
    ch = oracle.OracleChain(oracle.OracleModel.iso(D), seed=5, chain_id=2)
    ch.random_position()
    q0, lq0 = ch.q.copy(), ch.lq
    assert ch.find_local_optimum(1e-4, 0) == 0 and np.array_equal(ch.q, q0) and ch.lq == lq0
    assert ch.find_local_optimum(1e-4, 50) == 0
    assert np.abs(ch.q[:D]).max() < 1e-6 and abs(ch.lq) < 1e-10
    assert np.allclose(ch.grad[:D], -ch.q[:D], atol=1e-15)
    mu, sig = np.sin(np.arange(100.0)), np.logspace(-1, 1, 100)
    c2 = oracle.OracleChain(oracle.OracleModel.diag(mu, 1.0 / sig ** 2), seed=5, chain_id=0)
    c2.random_position()
    l0 = c2.lq
    assert c2.find_local_optimum(1e-4, 50) == 0 and c2.lq > l0 + 100.0
    assert np.isclose(c2.lq, -0.5 * np.sum(((c2.q[:100] - mu) / sig) ** 2), rtol=1e-12)
