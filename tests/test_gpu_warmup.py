"""GPU parity of the adaptation path and the caller loops (8f rank 1-2)
against the CPU oracle: initial stepsize search, per-chain dual averaging, metric windows, the full
mcmc_with_warmup schedule, the reference-surface API, and the global-eps mode.  Bit-exact fp64 unless a
tolerance is written in the test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def diag(D):
    return np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


SHORT = dict(init_steps=15, middle_steps=10, doubling_stages=3, terminating_steps=10)


@pytest.mark.parametrize("kind,D", [("iso", 32), ("diag", 100), ("diag", 1024)])
def test_initial_stepsize_search(idhmc, oracle, kind, D):
    C = 10
    if kind == "iso":
        gm, om = idhmc.IsoGaussian(D), oracle.OracleModel.iso(D)
    else:
        mu, sig = diag(D)
        gm, om = idhmc.DiagGaussian(mu, sigma=sig), oracle.OracleModel.diag(mu, 1 / sig ** 2)
    eng = idhmc.Engine(gm, C, seed=42)
    eng.random_position()
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()
    ref = []
    for c in range(C):
        ch = oracle.OracleChain(om, seed=42, chain_id=c)
        ch.random_position()
        ch.rand_p(0)
        rc, e = ch.find_initial_stepsize()
        assert rc == 0
        ref.append(e)
    assert same_bits(eng.eps, ref)
    assert np.all(eng.eps > 0)


def test_stepsize_search_nonfinite_start_is_an_error(idhmc):
    eng = idhmc.Engine(idhmc.IsoGaussian(16), 3, seed=1)
    q = np.zeros((3, 16)); q[1, 2] = np.inf
    eng.set_q(q)
    eng.refresh_momentum(0)
    with pytest.raises(idhmc.IdhmcError) as e:
        eng.find_initial_stepsize()
    assert e.value.code == 5 and "non-finite" in str(e.value)      # reference src/stepsize.jl:152-153


def test_tuning_stage_and_metric(idhmc, oracle):
    """one TuningNUTS{Diagonal} stage: eps trace, draws, regularised metric, final eps"""
    D, C, N = 100, 6, 30
    mu, sig = diag(D)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(max_depth=7), seed=9)
    eng.random_position()
    eng.set_eps(0.05)
    draws, stats = eng.tuning_stage(N, True, 0, store_draws=True)
    om = oracle.OracleModel.diag(mu, 1 / sig ** 2)
    oopt = oracle.default_options(max_depth=7)
    for c in range(C):
        ch = oracle.OracleChain(om, oopt, seed=9, chain_id=c)
        ch.random_position()
        da = oracle.DAState()
        L = oracle.lib()
        import ctypes as Cc
        L.orc_da_init(Cc.byref(da), 0.05)
        ch_draws = np.zeros((N, ch.L))
        for n in range(N):
            e = L.orc_da_current_eps(Cc.byref(da))
            st = ch.sample_tree(e, n + 1)
            ch_draws[n] = ch.q
            assert same_bits(draws[n, c], ch.q[:D]), "draw %d chain %d" % (n, c)
            assert stats[n, c]["steps"] == st.steps and stats[n, c]["acceptance_rate"] == st.acceptance_rate
            L.orc_da_adapt(Cc.byref(oopt), Cc.byref(da), st.acceptance_rate)
        minv, w = oracle.metric_from_draws(ch_draws, D, 5.0 / N)
        assert same_bits(eng.minv[c], minv[:D])
        assert eng.eps[c] == L.orc_da_final_eps(Cc.byref(da))


@pytest.mark.parametrize("kind,D,C", [("iso", 32, 4), ("diag", 100, 5), ("diag", 1024, 3), ("diag", 400, 9)])
def test_mcmc_with_warmup_matches_oracle(idhmc, oracle, kind, D, C):
    """the whole schedule (search, 3+2 tuning stages, sampling) on the device vs one oracle thread per chain"""
    if kind == "iso":
        gm, om = idhmc.IsoGaussian(D), oracle.OracleModel.iso(D)
    else:
        mu, sig = diag(D)
        gm, om = idhmc.DiagGaussian(mu, sigma=sig), oracle.OracleModel.diag(mu, 1 / sig ** 2)
    N = 20
    eng = idhmc.Engine(gm, C, idhmc.default_options(max_depth=8, **SHORT), seed=314)
    draws, stats = eng.mcmc_with_warmup(N)
    rc, ochains, ostats, oeps = oracle.threaded_mcmc(om, N, C, oracle.default_options(max_depth=8, **SHORT), seed=314)
    assert rc == 0
    assert same_bits(eng.eps, oeps)
    for n in range(N):
        assert same_bits(draws[n], ochains[:, n, :D]), "draw %d" % n
    assert np.array_equal(stats.T, ostats[:, :N])
    assert same_bits(eng.lq, [oracle.OracleModel.logdensity_and_gradient(om, ochains[c, N - 1, :D])[0] for c in range(C)])
    assert eng.total_steps() >= int(ostats[:, :N]["steps"].sum())


def test_api_threaded_mcmc_shapes_and_parity(idhmc, oracle):
    """reference surface: threaded_mcmc(l, N; nchains) -> (chains, tree_statistics) with NS = max(N, longest stage)"""
    D, C, N = 32, 4, 12
    stages = idhmc.default_warmup_stages(init_steps=15, middle_steps=10, doubling_stages=3, terminating_steps=10)
    chains, stats = idhmc.threaded_mcmc(idhmc.IsoGaussian(D), N, warmup_stages=stages, algorithm=idhmc.NUTS(max_depth=6),
                                        nchains=C, seed=99)
    NS = idhmc.num_stored(N, stages)
    assert NS == 40 and chains.shape == (C, NS, D) and stats.shape == (C, NS) and stats.dtype == idhmc.TreeStatisticsNUTS
    # the reference's default stages begin with FindLocalOptimum() (src/warmup.jl:362): the oracle runs it too
    rc, och, ost, _ = oracle.threaded_mcmc(oracle.OracleModel.iso(D), N, C,
                                           oracle.default_options(max_depth=6, local_opt_iterations=50, **SHORT), seed=99)
    assert same_bits(chains, och[:, :, :D])                   # including the warmup leftovers beyond column N
    assert np.array_equal(stats, ost)
    chain, st = idhmc.mcmc_with_warmup(idhmc.IsoGaussian(D), N, warmup_stages=stages, algorithm=idhmc.NUTS(max_depth=6), seed=99)
    assert chain.shape == (NS, D) and same_bits(chain, chains[0])
    e = idhmc.EBFMI(stats[:, :N])
    assert e.shape == (C,) and np.all(e > 0)


def test_api_initialization_and_fixed_stepsize(idhmc):
    D, C = 16, 3
    q0 = np.linspace(-1, 1, D)
    stages = idhmc.fixed_stepsize_warmup_stages(middle_steps=10, doubling_stages=2)
    chains, stats = idhmc.threaded_mcmc(idhmc.IsoGaussian(D), 15, nchains=C, warmup_stages=stages,
                                        initialization={"q": q0, "eps": 0.4, "kappa": idhmc.GaussianKineticEnergy.identity(D, 0.5)})
    assert chains.shape == (C, 20, D) and np.isfinite(chains).all()
    assert (stats["steps"][:, :15] >= 1).all()


def test_running_moments(idhmc):
    D, C, N = 64, 8, 40
    eng = idhmc.Engine(idhmc.IsoGaussian(D), C, idhmc.default_options(max_depth=6), seed=5)
    eng.random_position()
    eng.set_eps(0.4)
    eng.moments_reset()
    draws, _ = eng.mcmc(N, 0)
    mean, var, cnt = eng.moments()
    assert (cnt == N).all()
    assert np.allclose(mean, draws.mean(axis=0), rtol=1e-12, atol=1e-13)      # Welford vs two-pass: 1e-12
    assert np.allclose(var, draws.var(axis=0, ddof=1), rtol=1e-10, atol=1e-12)


def test_global_eps_mode(idhmc):
    """north_star's global dual averaging: every chain uses one eps driven by the pooled mean acceptance.
    Checked against the Hoffman-Gelman recursion evaluated in numpy on the device's own acceptance rates
    (tolerance 1e-12: numpy's log/exp vs the engine's deterministic ones)."""
    D, C, N = 64, 32, 25
    mu, sig = diag(D)
    opt = idhmc.default_options(max_depth=7, eps_mode=idhmc.EPS_GLOBAL)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, opt, seed=12)
    eng.random_position()
    eng.set_eps(0.03)
    _, stats = eng.tuning_stage(N, False, 0)
    mu_da, m, Hbar, le, lb = np.log(10) + np.log(0.03), 0, 0.0, np.log(0.03), 0.0
    for n in range(N):
        a = stats[n]["acceptance_rate"].mean()
        m += 1
        Hbar += (0.8 - a - Hbar) / (m + 10)
        le = mu_da - np.sqrt(m) / 0.05 * Hbar
        lb += m ** -0.75 * (le - lb)
    eps = eng.eps
    assert np.all(eps == eps[0]) and abs(eps[0] - np.exp(lb)) < 1e-12 * np.exp(lb)


def test_global_eps_through_the_allreduce_hook(idhmc):
    """the exchange path the multi-GPU run uses: the library reduces into a torch CUDA tensor on torch's
    stream, calls back into Python for the all-reduce (a no-op at world size 1), and continues on the device.
    Must equal the hook-less run bit for bit; the hook must have been called once per warm-up transition and once
    more for the stage's status agreement (errors are agreed on across ranks before they are returned)."""
    import torch
    D, C, N = 64, 16, 12
    mu, sig = diag(D)
    opt = idhmc.default_options(max_depth=6, eps_mode=idhmc.EPS_GLOBAL)
    res = []
    calls = []
    for use_hook in (False, True):
        eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, opt, seed=3)
        if use_hook:
            buf = torch.zeros(idhmc.XCHG_DOUBLES, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            st = torch.cuda.Stream()             # a real stream: the default stream's NULL handle means "library stream"
            eng.set_stream(st.cuda_stream)

            def hook(_ptr, buf=buf, st=st):
                calls.append(1)
                with torch.cuda.stream(st):
                    idhmc.distributed.allreduce_xchg(buf)
            eng.set_allreduce_hook(hook, buf.data_ptr())
        eng.random_position()
        eng.set_eps(0.05)
        draws, stats = eng.tuning_stage(N, False, 0, store_draws=True)
        res.append((draws, eng.eps))
        if use_hook:
            eng.synchronize()
            # the stage's last exchange is its status agreement: sums zero, every chain counted, none with a pending error
            assert np.array_equal(buf.cpu().numpy(), [0.0, 0.0, C, 0.0])
            eng.accept_sum(buf.data_ptr())
            eng.synchronize()
            rec = buf.cpu().numpy()                                  # the fixed-point record of the last transition
            assert np.array_equal(rec, idhmc.xchg_accumulate(idhmc.XCHG_ACCEPT, stats[-1]["acceptance_rate"]))
            assert rec[2] == C and abs(idhmc.xchg_mean(idhmc.XCHG_ACCEPT, rec) - stats[-1]["acceptance_rate"].mean()) < 1e-15
            keep = idhmc.distributed.attach_global_eps(eng)          # the packaged form of the same wiring
            assert keep[0].shape == (idhmc.XCHG_DOUBLES,)
        eng.close()
    assert len(calls) == N + 1                   # one per warm-up transition + the stage's status agreement
    assert same_bits(res[0][0], res[1][0]) and same_bits(res[0][1], res[1][1])


def test_global_eps_through_the_native_rccl_communicator(idhmc):
    """idhmc_comm_*: the library's own RCCL communicator (single-rank here: one GPU per box) carries the
    4-double all-reduce on the context's stream.  Must equal the communicator-less run bit for bit, the
    explicit all-reduce entry point must leave a buffer unchanged at one rank, and idhmc_comm_info counts the calls
    (one for the pooled initial stepsize, one per warm-up transition)."""
    import torch
    D, C, N = 64, 16, 12
    mu, sig = diag(D)
    opt = idhmc.default_options(max_depth=6, eps_mode=idhmc.EPS_GLOBAL)
    res = []
    for native in (False, True):
        eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, opt, seed=3)
        if native:
            idhmc.distributed.attach_global_eps_native(eng, rank=0, world=1)
            with pytest.raises(idhmc.IdhmcError):
                eng.comm_init(1, 0, eng.comm_unique_id())       # a context holds one communicator
            buf = torch.tensor([3.5, 16.0, 2.0, 0.0], dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            eng.comm_allreduce(buf.data_ptr())
            eng.synchronize()
            assert buf.tolist() == [3.5, 16.0, 2.0, 0.0]
            assert eng.comm_info() == (1, 0, 1)
        else:
            assert eng.comm_info() == (0, 0, 0)
        eng.random_position()
        eng.refresh_momentum(0)
        eng.find_initial_stepsize()
        e0 = eng.eps
        assert np.all(e0 == e0[0])
        draws, stats = eng.tuning_stage(N, False, 0, store_draws=True)
        res.append((draws, eng.eps))
        if native:
            assert eng.comm_info() == (1, 0, 1 + 1 + N + 1)     # + the stage's status agreement
            eng.comm_destroy()
        eng.close()
    assert same_bits(res[0][0], res[1][0]) and same_bits(res[0][1], res[1][1])


def test_posterior_moments_cfg_small(idhmc):
    """statistical parity with analytic truth: mean within 4 sigma/sqrt(ESS),
    variance within 4 sigma^2 sqrt(2/ESS), mean acceptance within +-0.05 of a plausible band"""
    D, C, N = 100, 64, 150
    mu, sig = diag(D)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(), seed=2)
    draws, stats = eng.mcmc_with_warmup(N)
    x = draws.transpose(1, 0, 2)                                # chains, draws, D
    ess = np.array([idhmc.ess(x[c]) for c in range(C)]).sum(axis=0)
    assert np.all(np.abs(x.mean(axis=(0, 1)) - mu) < 4 * sig / np.sqrt(ess))
    assert np.all(np.abs(x.var(axis=(0, 1)) - sig ** 2) < 4 * sig ** 2 * np.sqrt(2 / np.minimum(ess, C * N)) + 0.05 * sig ** 2)
    assert 0.7 < stats["acceptance_rate"].mean() < 0.95
    s = idhmc.summarize_tree_statistics(stats)
    assert s.termination_counts["divergence"] == 0


def test_log_progress_report(idhmc, caplog):
    """reference LogProgressReport: messages at the points the reference reports (src/warmup.jl:161,197) plus stage ends"""
    import logging
    stages = idhmc.default_warmup_stages(init_steps=5, middle_steps=5, doubling_stages=1, terminating_steps=5)
    with caplog.at_level(logging.INFO, logger="InplaceDHMC"):
        idhmc.threaded_mcmc(idhmc.IsoGaussian(8), 5, nchains=3, warmup_stages=stages,
                            reporter=idhmc.LogProgressReport(chain_id="gpu0"))
    text = "\n".join(r.getMessage() for r in caplog.records)
    assert "finding initial optimum" in text and "found initial stepsize" in text and "chain_id = gpu0" in text
    assert text.count("warmup stage done") == 3 and "mcmc done" in text
    with caplog.at_level(logging.INFO, logger="InplaceDHMC"):
        caplog.clear()
        idhmc.threaded_mcmc(idhmc.IsoGaussian(8), 5, nchains=3, warmup_stages=stages, reporter=idhmc.NoProgressReport())
    assert not caplog.records


def test_pooled_metric(idhmc):
    """IDHMC_METRIC_POOLED (an addition for the many-chain regime, not reference semantics): one M^-1 for all chains from
    the pooled windows.  The device estimate equals the formula applied to the stored draws of the stage (numpy as the
    comparator: 1e-10 relative), every chain reads the same metric, the native communicator (one rank) changes nothing,
    and the posterior comes out right."""
    D, C, N = 100, 48, 40
    mu, sig = diag(D)
    opt = idhmc.default_options(metric_mode=idhmc.METRIC_POOLED, eps_mode=idhmc.EPS_GLOBAL, max_depth=8)
    res = []
    for native in (False, True):
        eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, opt, seed=12)
        if native:
            idhmc.distributed.attach_global_eps_native(eng, rank=0, world=1)
        eng.random_position()
        eng.set_eps(0.05)
        draws, stats = eng.tuning_stage(N, True, 0, store_draws=True)          # [N][C][D]
        minv = eng.minv
        assert np.array_equal(minv, np.broadcast_to(minv[0], minv.shape))       # shared by all chains
        x = draws.reshape(N * C, D)
        Nt, lam = float(N * C), 5.0 / N
        S = ((x - x.mean(axis=0)) ** 2).sum(axis=0)
        ref = S * Nt / ((Nt + lam) * (Nt - 1.0)) + 1e-3 * lam / (Nt + lam)      # src/hamiltonian.jl:156-158 with the pooled count
        assert np.allclose(minv[0], ref, rtol=1e-10, atol=0)
        res.append((draws, minv))
        eng.close()
    assert same_bits(res[0][0], res[1][0]) and same_bits(res[0][1], res[1][1])
    # end to end: default schedule (shortened), pooled metric + global stepsize
    short = dict(init_steps=30, middle_steps=15, doubling_stages=3, terminating_steps=20)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), 64,
                       idhmc.default_options(metric_mode=idhmc.METRIC_POOLED, eps_mode=idhmc.EPS_GLOBAL, **short), seed=2)
    d, st = eng.mcmc_with_warmup(120)
    xs = d.reshape(-1, D)
    assert np.all(np.abs(xs.mean(axis=0) - mu) < 0.1 * sig) and np.all(np.abs(xs.var(axis=0) / sig ** 2 - 1.0) < 0.15)
    assert np.all(np.abs(eng.minv[0] / sig ** 2 - 1.0) < 0.25)                  # 64 chains x 60 draws in the last window
    assert 0.6 < st["acceptance_rate"][-50:].mean() < 0.95
    eng.close()
