"""BASELINE.json's full sizes (65 536 chains x 1024 dims) checked through size-independent properties:
time reversibility, energy conservation, independence of a chain's result from how many other chains
run beside it (sharding invariance), step accounting, and spot parity of scattered chains against the
oracle.  Everything goes through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, C = 1024, 65536


def workload():
    return np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)


@pytest.fixture(scope="module")
def big(idhmc):
    mu, sig = workload()
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED), seed=1)
    eng.set_minv(sig ** 2)
    yield eng
    eng.close()


def test_leapfrog_reversible_and_conserves_energy_at_full_size(big):
    big.random_position()
    big.refresh_momentum(1)
    q0, p0, h0 = big.q, big.p, big.logdensity()
    big.leapfrog(0.1, 1)
    for _ in range(9):
        big.leapfrog(0.1, 1)
    h1 = big.logdensity()
    assert np.isfinite(h1).all()
    # the start is U[-2,2]^D, ~20 sigma out in the stiff directions (|H| ~ 1e4): ten leapfrogs at eps = 0.1
    # change the energy by O(eps^2 |H|); 0.5 % is the tolerance
    assert np.abs(h1 - h0).max() < 0.005 * np.abs(h0).max()
    big.leapfrog(-0.1, 10)
    assert np.abs(big.q - q0).max() <= 1e-10 and np.abs(big.p - p0).max() <= 1e-10
    assert np.abs(big.logdensity() - h0).max() <= 1e-8 * np.abs(h0).max()


def test_fused_steps_equal_single_steps_at_full_size(big):
    big.random_position()
    big.refresh_momentum(2)
    big.leapfrog(0.05, 4)
    qa, la = big.q[:4096], big.lq
    big.random_position()
    big.refresh_momentum(2)
    for _ in range(4):
        big.leapfrog(0.05, 1)
    assert np.array_equal(qa, big.q[:4096]) and np.array_equal(la, big.lq)


def test_nuts_is_invariant_to_the_number_of_chains(idhmc, big):
    """chain c's transition does not depend on the other 65 535 (RNG keyed by global id, no shared state)"""
    mu, sig = workload()
    big.random_position()
    big.set_eps(0.2)
    stats = []
    for it in (1, 2, 3):
        big.nuts_transition(it)
        stats.append(big.tree_stats())
    qbig = big.q
    assert big.total_steps() >= sum(int(s["steps"].sum()) for s in stats)
    for first in (0, 30000, C - 64):
        small = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), 64, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED),
                             seed=1, first_chain=first)
        small.set_minv(sig ** 2)
        small.random_position()
        small.set_eps(0.2)
        for it in (1, 2, 3):
            small.nuts_transition(it)
            assert np.array_equal(small.tree_stats(), stats[it - 1][first:first + 64])
        assert np.array_equal(small.q, qbig[first:first + 64])
        small.close()
    s = stats[-1]
    assert (s["steps"] >= 1).all() and (s["depth"] <= 10).all()
    assert ((s["steps"] <= (1 << s["depth"]) - 1 + (1 << s["depth"]))).all()


def test_several_transitions_per_launch_at_full_size(idhmc, oracle, big):
    """idhmc_nuts_transitions at 65 536 x 1024 (every XCD's range has 8192 chains and 32 workgroups handing chains to one another):
    the same bits as single launches, and scattered chains against the oracle after the fused launch"""
    mu, sig = workload()
    big.random_position()
    big.set_eps(0.2)
    s0 = big.total_steps()
    for it in (1, 2, 3, 4):
        big.nuts_transition(it)
    q1, st1, n1 = big.q, big.tree_stats(), big.total_steps() - s0
    big.random_position()
    s0 = big.total_steps()
    big.nuts_transitions(1, 4)
    assert big.poll_abort(0) == 0
    assert np.array_equal(big.q, q1) and np.array_equal(big.tree_stats(), st1) and big.total_steps() - s0 == n1
    om = oracle.OracleModel.diag(mu, 1.0 / sig ** 2)
    for c in (0, 8191, 8192, 40000, 65535):
        ch = oracle.OracleChain(om, seed=1, chain_id=c)
        ch.set_minv(sig ** 2)
        ch.random_position()
        for it in (1, 2, 3, 4):
            s = ch.sample_tree(0.2, it)
        assert np.array_equal(q1[c], ch.q[:D]) and (st1[c]["depth"], st1[c]["steps"], st1[c]["pi"]) == (s.depth, s.steps, s.pi)


def test_scattered_chains_match_the_oracle_at_full_size(idhmc, oracle, big):
    mu, sig = workload()
    big.random_position()
    big.set_eps(0.15)
    big.nuts_transition(1)
    big.nuts_transition(2)
    q, st = big.q, big.tree_stats()
    om = oracle.OracleModel.diag(mu, 1.0 / sig ** 2)
    for c in (0, 1, 4097, 32768, 65535):
        ch = oracle.OracleChain(om, seed=1, chain_id=c)
        ch.set_minv(sig ** 2)
        ch.random_position()
        ch.sample_tree(0.15, 1)
        s = ch.sample_tree(0.15, 2)
        assert np.array_equal(q[c], ch.q[:D])
        assert (st[c]["depth"], st[c]["steps"], st[c]["pi"]) == (s.depth, s.steps, s.pi)


def test_headline_sweep_on_the_placed_arrays_matches_the_oracle(idhmc, oracle, big):
    """the bench's own instantiation at the bench's own size: three `leapfrog(0.1, 1)` sweeps of k_leapfrog1<8, DiagGaussian> over the
    512 MiB-per-array state that place_state laid out (the only size at which the placement search is live), scattered chains
    compared with the oracle with ==: q, p, grad l, l(q) and pi (src/kinetic_energy.jl:126-163)"""
    mu, sig = workload()
    probe_GBps, candidates = big.placement_info()
    cost = big.placement_cost()
    assert candidates >= 1 and probe_GBps > 1000.0, (probe_GBps, candidates)          # the search ran and measured something
    # (the pair walk may hold untouched spacers of up to 64 GiB for a moment, the walk over whole sets 16 GiB; both are given back)
    assert cost["peak_transient_bytes"] <= (64 << 30) + (2 << 30) and 0.0 < cost["create_ms"] < 1000.0 and cost["single_array_GBps"] > 1000.0, cost
    big.random_position()
    big.refresh_momentum(5)
    for _ in range(3):
        big.leapfrog(0.1, 1)
    q, p, g, lq, pi = big.q, big.p, big.grad, big.lq, big.logdensity()
    om = oracle.OracleModel.diag(mu, 1.0 / sig ** 2)
    for c in (0, 1, 4097, 32768, 65535):
        ch = oracle.OracleChain(om, seed=1, chain_id=c)
        ch.set_minv(sig ** 2)
        ch.random_position()
        ch.rand_p(5)
        for _ in range(3):
            ch.leapfrog(0.1)
        assert np.array_equal(q[c], ch.q[:D]) and np.array_equal(p[c], ch.p[:D]) and np.array_equal(g[c], ch.grad[:D]), c
        assert lq[c] == ch.lq and pi[c] == ch.logdensity(), c


# ---- BASELINE.json configs[3] at full size: 256-dim dense MVN, 16 384 chains ---------------------------------------
def dense_workload(Dd=256, seed=7):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((Dd, Dd)))
    lam = np.logspace(-2, 0, Dd)
    P = (Q / lam) @ Q.T
    return np.cos(np.arange(Dd, dtype=np.float64)), 0.5 * (P + P.T), Q, lam


def test_dense_full_size_properties(idhmc, oracle):
    """matrix-core leapfrog: n steps in one launch == n single-step launches (bit for bit), time reversal returns to the
    start, the gradient is Sigma^-1 (q - mu) to rounding; cooperative NUTS: a chain's draw does not depend on which
    15 other chains share its workgroup (sharding invariance), and scattered chains equal the oracle"""
    Dd, Cd = 256, 16384
    mu, P, Q, lam = dense_workload(Dd)
    opt = idhmc.default_options(metric_mode=idhmc.METRIC_SHARED, max_depth=8)
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), Cd, opt, seed=5)
    eng.random_position()
    g = eng.grad
    assert np.abs(g + (eng.q - mu) @ P).max() < 1e-9 * np.abs(g).max()
    eng.refresh_momentum(1)
    q0, p0 = eng.q, eng.p
    eng.leapfrog(0.01, 6)
    qa, pa, la = eng.q, eng.p, eng.lq
    eng.set_q(q0); eng.set_p(p0)
    for _ in range(6):
        eng.leapfrog(0.01, 1)
    assert np.array_equal(eng.q, qa) and np.array_equal(eng.p, pa) and np.array_equal(eng.lq, la)
    eng.leapfrog(-0.01, 6)
    assert np.abs(eng.q - q0).max() < 1e-9 and np.abs(eng.p - p0).max() < 1e-9
    # NUTS: full context vs a 40-chain context holding chains 1000..1039 (different workgroup mates, same chain ids)
    eng.set_q(q0)
    eng.set_eps(0.05)
    for it in (1, 2, 3):
        eng.nuts_transition(it)
    sub = idhmc.Engine(idhmc.DenseMVN(mu, P), 40, opt, seed=5, first_chain=1000)
    sub.set_q(q0[1000:1040])
    sub.set_eps(0.05)
    for it in (1, 2, 3):
        sub.nuts_transition(it)
    assert np.array_equal(eng.q[1000:1040], sub.q)
    assert np.array_equal(eng.tree_stats()["steps"][1000:1040], sub.tree_stats()["steps"])
    om = oracle.OracleModel.dense(mu, P)
    for c in (0, 1017, Cd - 1):
        ch = oracle.OracleChain(om, oracle.default_options(max_depth=8), seed=5, chain_id=c)
        ch.set_q(q0[c])
        for it in (1, 2, 3):
            ch.sample_tree(0.05, it)
        assert np.array_equal(eng.q[c], ch.q[:Dd])
    sub.close()
    eng.close()


def test_configs2_warmup_at_full_size(idhmc):
    """BASELINE.json configs[2] at its full size -- 65 536 chains x 1024 dims, every chain adapting its OWN stepsize and
    diagonal metric (reference semantics, src/warmup.jl:284-309) -- on a shortened schedule (same stage structure:
    stepsize search, init, doubling windows with metric updates, terminating stage), then 20 draws reduced on the device.
    Size-independent checks: no chain raised a status, mean acceptance of the DRAWS in [0.78, 0.92] (the draws run at the
    averaged iterate exp(log eps bar), src/stepsize.jl:241, which sits below the last adaptive stepsize: acceptance ends
    above the target 0.8 -- 0.85 on the full default schedule, profiles/r01_cfg3_full_run.log, 0.86 on this short one), the
    adapted metrics' median M^-1 / sigma^2 within 10 %, the pooled posterior mean within 4 sigma / sqrt(C n) per
    coordinate (median z <= 1, max z <= 5: 1024 coordinates), chains' eps identical to what one of them gets alone
    (sharding invariance of the per-chain path at full size)."""
    mu, sig = workload()
    short = dict(init_steps=30, middle_steps=15, doubling_stages=3, terminating_steps=20)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(**short), seed=5)
    n = 20
    eng.moments_reset()
    _, stats = eng.mcmc_with_warmup(n, store_draws=False)          # raises on any chain's status (eps underflow, search failure)
    assert eng.poll_abort(0) == 0
    acc = stats["acceptance_rate"]
    assert 0.78 < acc.mean() < 0.92, acc.mean()
    assert (stats["term_left"] == stats["term_right"]).mean() < 1e-3          # divergences: none to speak of
    eps = eng.eps
    assert np.isfinite(eps).all() and 0.05 < np.median(eps) < 1.5
    # adapted metric against the truth (sample 512 chains; the device array is 512 MiB)
    minv = eng.minv[::128]
    ratio = np.median(minv / sig ** 2)
    assert abs(ratio - 1.0) < 0.10, ratio
    mean, var, cnt = eng.moments()
    assert (cnt == n).all()
    pooled = mean.mean(axis=0)
    z = np.abs(pooled - mu) / (sig / np.sqrt(C * n))
    assert np.median(z) <= 1.0 and z.max() <= 5.0, (np.median(z), z.max())
    # variance: pooled within-chain variance + between-chain variance of the means ~ sigma^2
    tot = var.mean(axis=0) * (n - 1) / n + mean.var(axis=0)
    assert np.all(np.abs(tot / sig ** 2 - 1.0) < 0.05), np.abs(tot / sig ** 2 - 1.0).max()
    eng.close()
    # one chain of the 65 536 alone: the same adapted stepsize and draw
    c0 = 4242
    one = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), 1, idhmc.default_options(**short), seed=5, first_chain=c0)
    one.mcmc_with_warmup(n, store_draws=False)
    assert one.eps[0] == eps[c0]
    one.close()


def test_configs4_shape_global_eps_pooled_metric_at_full_size(idhmc):
    """configs[4]'s shard shape on one GPU: 65 536 chains x 1024 dims with ONE dual-averaging stepsize (the exact exchange
    record, here through a single-rank RCCL communicator) and the pooled metric, shortened schedule, 20 draws.  Every chain
    holds the same eps and metric; the pooled metric of 65 536 x 60 draws matches sigma^2 to 1 % (median) / 5 % (every
    coordinate); mean acceptance of the draws within 0.05 of the target (one eps for everybody averages the per-chain
    scatter away); pooled posterior mean within 5 sigma / sqrt(C n) per coordinate; all-reduces = 1 + transitions + 1 per stage + 3 per window."""
    mu, sig = workload()
    short = dict(init_steps=30, middle_steps=15, doubling_stages=3, terminating_steps=20)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C,
                       idhmc.default_options(eps_mode=idhmc.EPS_GLOBAL, metric_mode=idhmc.METRIC_POOLED, **short), seed=6)
    idhmc.distributed.attach_global_eps_native(eng, rank=0, world=1)
    n = 20
    eng.moments_reset()
    _, stats = eng.mcmc_with_warmup(n, store_draws=False)
    eps, minv = eng.eps, eng.minv[:4]
    assert np.all(eps == eps[0]) and np.array_equal(minv, np.broadcast_to(minv[0], minv.shape))
    ratio = minv[0] / sig ** 2
    assert abs(np.median(ratio) - 1.0) < 0.01 and np.all(np.abs(ratio - 1.0) < 0.05), (np.median(ratio), ratio.min(), ratio.max())
    assert abs(stats["acceptance_rate"].mean() - 0.8) < 0.05, stats["acceptance_rate"].mean()
    mean, var, cnt = eng.moments()
    z = np.abs(mean.mean(axis=0) - mu) / (sig / np.sqrt(C * n))
    assert np.median(z) <= 1.0 and z.max() <= 5.0, (np.median(z), z.max())
    ranks, _, allreduces = eng.comm_info()
    transitions = 30 + 15 + 30 + 60 + 20
    # search + one per transition + one status agreement per stage (5 stages) + (count, pass 0, pass 1) per window
    assert ranks == 1 and allreduces == 1 + transitions + 5 + 3 * 3
    eng.close()
