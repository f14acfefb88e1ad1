"""Several transitions per launch (idhmc_nuts_transitions): the device hands out (transition, chain) pairs from one queue and a chain's
state passes between workgroups inside the launch.  Every random number is addressed by (seed, chain, transition), so the result must be
bit for bit that of single-transition launches -- which the other suites hold against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def dense_problem(D, seed=2):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((D, D)) / np.sqrt(D)
    P = A @ A.T + np.eye(D)
    return rng.standard_normal(D), 0.5 * (P + P.T)


def make(idhmc, kind, D, C, shared, seed=9, max_depth=7):
    opt = idhmc.default_options(max_depth=max_depth, metric_mode=idhmc.METRIC_SHARED) if shared else idhmc.default_options(max_depth=max_depth)
    if kind == "diag":
        rng = np.random.default_rng(4)
        model = idhmc.DiagGaussian(rng.standard_normal(D), np.exp(rng.standard_normal(D)))
    elif kind == "iso":
        model = idhmc.IsoGaussian(D)
    else:
        model = idhmc.DenseMVN(*dense_problem(D))
    eng = idhmc.Engine(model, C, opt, seed=seed)
    eng.random_position()
    return eng


def snapshot(eng):
    st = eng.tree_stats()
    return dict(q=eng.q.copy(), grad=eng.grad.copy(), eps=eng.eps.copy(), steps=int(eng.total_steps()),
                **{"st_" + k: np.array(st[k]) for k in ("depth", "steps", "term_left", "term_right", "pi", "acceptance_rate")})


def assert_same(a, b):
    assert a["steps"] == b["steps"]
    for k in ("st_depth", "st_steps", "st_term_left", "st_term_right"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in ("q", "grad", "eps", "st_pi", "st_acceptance_rate"):
        assert same_bits(a[k], b[k]), k


CASES = [("diag", 40, 37, False, 0.3), ("diag", 1024, 300, False, 0.25), ("diag", 1024, 300, True, 0.25), ("iso", 300, 64, False, 0.4),
         ("dense", 256, 37, False, 0.03), ("dense", 256, 100, True, 0.03), ("dense", 100, 16, False, 0.05), ("dense", 512, 24, False, 0.02)]


@pytest.mark.parametrize("kind,D,C,shared,eps", CASES)
def test_fused_launch_equals_single_launches(idhmc, kind, D, C, shared, eps):
    """plain transitions, then transitions that adapt the stepsize (and, with a per-chain metric, fill the metric window)"""
    T = 9
    single, fused = make(idhmc, kind, D, C, shared), make(idhmc, kind, D, C, shared)
    for e in (single, fused):
        e.set_eps(eps)
    for it in range(1, T + 1):
        single.nuts_transition(it)
    fused.nuts_transitions(1, T)
    assert_same(snapshot(single), snapshot(fused))
    flags = idhmc.T_ADAPT_EPS | (0 if shared else idhmc.T_ACCUM_METRIC)
    for e in (single, fused):
        e.da_init()
        if not shared:
            e.metric_begin()
    for it in range(T + 1, 2 * T + 1):
        single.nuts_transition(it, flags)
    fused.nuts_transitions(T + 1, T, flags)
    assert_same(snapshot(single), snapshot(fused))
    if not shared:
        for e in (single, fused):
            e.metric_update(0.1)
        assert same_bits(single.minv, fused.minv)


@pytest.mark.parametrize("kind,D,C", [("diag", 200, 3), ("dense", 256, 3), ("dense", 64, 1), ("diag", 1024, 5)])
def test_few_chains_many_transitions(idhmc, kind, D, C):
    """fewer chains than resident wavefronts: a chain's next transition is handed out while its previous one still runs, so the taker
    really waits (the cooperative dense kernel: while serving its workgroup's rounds)"""
    T = 25
    single, fused = make(idhmc, kind, D, C, False, max_depth=5), make(idhmc, kind, D, C, False, max_depth=5)
    for e in (single, fused):
        e.set_eps(0.1 if kind == "diag" else 0.04)
    for it in range(1, T + 1):
        single.nuts_transition(it)
    fused.nuts_transitions(1, T)
    assert_same(snapshot(single), snapshot(fused))


@pytest.mark.parametrize("kind,D,C,seed", [("diag", 64, 300, 1), ("diag", 64, 300, 2), ("dense", 64, 200, 3), ("dense", 128, 130, 4)])
def test_many_hand_overs(idhmc, kind, D, C, seed):
    """60 short transitions of a few hundred chains in one launch: every chain changes workgroups (hence CUs) dozens of times, with the
    stepsize adapting (the state the next transition reads is written by the previous one's epilogue)"""
    T = 60
    single, fused = make(idhmc, kind, D, C, False, seed=seed, max_depth=4), make(idhmc, kind, D, C, False, seed=seed, max_depth=4)
    for e in (single, fused):
        e.set_eps(0.3 if kind == "diag" else 0.1)
        e.da_init()
        e.metric_begin()
    fl = idhmc.T_ADAPT_EPS | idhmc.T_ACCUM_METRIC
    for it in range(1, T + 1):
        single.nuts_transition(it, fl)
    fused.nuts_transitions(1, T, fl)
    assert_same(snapshot(single), snapshot(fused))
    for e in (single, fused):
        e.metric_update(0.05)
    assert same_bits(single.minv, fused.minv)


def test_bad_arguments(idhmc):
    eng = make(idhmc, "diag", 40, 8, False)
    assert eng.fused_launch_info() == (True, True)          # MI355X: workgroups b and b + 8 share an XCD (probed at creation)
    with pytest.raises(idhmc.IdhmcError):
        eng.nuts_transitions(1, 0)
    with pytest.raises(idhmc.IdhmcError):
        eng.nuts_transitions(1, 4, idhmc.T_USE_DIRECTIONS)


def test_a_range_served_from_two_xcds_is_refused(idhmc):
    """the hand-over inside a launch relies on workgroups b and b + 8 sharing an XCD; the kernel checks it (HW_REG_XCC_ID) and raises the
    abort code otherwise.  Bit 30 of the flags (test suite only) makes the workgroups of a range report different ids"""
    eng = make(idhmc, "diag", 40, 600, False)       # 38 workgroups: every range has workgroups b and b + 8
    eng.set_eps(0.3)
    eng.nuts_transitions(1, 3)
    assert eng.poll_abort(0) == 0
    eng.nuts_transitions(4, 3, 1 << 30)
    assert eng.poll_abort(0) == idhmc.ERR_HIP
    with pytest.raises(idhmc.IdhmcError) as e:      # the drivers report it (and clear it) like every device-side error
        eng.tuning_stage(3, False, 10, store_stats=False)
    assert e.value.code == idhmc.ERR_HIP
    eng.tuning_stage(3, False, 20, store_stats=False)
    assert eng.poll_abort(0) == 0


@pytest.mark.parametrize("kind,D,C,N", [("diag", 40, 37, 70), ("dense", 256, 20, 9), ("diag", 1024, 64, 130)])
def test_draws_and_records_of_fused_sampling(idhmc, kind, D, C, N, monkeypatch):
    """idhmc_mcmc with host arrays: the kernel writes every transition's draw and record into a staging block of K <= 64 transitions,
    blocks alternate between two buffers and are copied out while the next one computes (N = 70 and 130: a ragged last block, three
    blocks).  Same draws and records as one launch per transition (IDHMC_FUSE=0); also with only one of the two arrays wanted"""
    monkeypatch.setenv("IDHMC_FUSE", "0")
    plain = make(idhmc, kind, D, C, False)
    monkeypatch.delenv("IDHMC_FUSE")
    fused = make(idhmc, kind, D, C, False)
    assert plain.fused_launch_info() == (True, False) and fused.fused_launch_info() == (True, True)
    for e in (plain, fused):
        e.set_eps(0.2 if kind == "diag" else 0.04)
    d0, s0 = plain.mcmc(N, 0)
    d1, s1 = fused.mcmc(N, 0)
    assert same_bits(d0, d1) and np.array_equal(s0, s1) and same_bits(plain.q, fused.q)
    d0, _ = plain.mcmc(5, N, store_stats=False)
    d1, _ = fused.mcmc(5, N, store_stats=False)
    assert same_bits(d0, d1)
    _, s0 = plain.mcmc(7, N + 5, store_draws=False)
    _, s1 = fused.mcmc(7, N + 5, store_draws=False)
    assert np.array_equal(s0, s1) and same_bits(plain.q, fused.q)


def test_drivers_fuse_and_still_match_the_oracle(idhmc, oracle, monkeypatch):
    """IDHMC_FUSE=1: every warm-up stage is one launch (no per-transition record leaves the device there); the draws that follow are
    fetched per transition.  Same bits as the oracle's chains, i.e. as the unfused drivers"""
    monkeypatch.setenv("IDHMC_FUSE", "1")
    D, C, N = 64, 6, 12
    short = dict(init_steps=15, middle_steps=10, doubling_stages=2, terminating_steps=10, max_depth=8)
    mu, P = dense_problem(D)
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, idhmc.default_options(**short), seed=77)
    draws, stats = eng.mcmc_with_warmup(N)
    rc, och, ost, oeps = oracle.threaded_mcmc(oracle.OracleModel.dense(mu, P), N, C, oracle.default_options(**short), seed=77)
    assert rc == 0 and same_bits(eng.eps, oeps)
    for n in range(N):
        assert same_bits(draws[n], och[:, n, :D])
    assert np.array_equal(stats.T, ost[:, :N])
