"""Diagnostics: the reference's summarize_tree_statistics / EBFMI (src/diagnostics.jl:28-32, 61-101) computed from stored
records (numpy, host) against the same quantities reduced on the device (IDHMC_T_ACCUM_DIAG: per-chain running sums and
integer counters), and the moment-based ESS.  The counter -> summary step is host code of libidhmc.so and runs without
a GPU; the device accumulation is checked on the GPU against numpy on the very records it summarised."""
import numpy as np
import pytest


def _counters_from_records(idhmc, ts):
    """what the kernel epilogue accumulates, restated in numpy (integers only)"""
    from inplacedhmc_jl_amd import _lib
    ts = np.asarray(ts).ravel()
    cn = np.zeros(_lib.DIAG_COUNTERS, dtype=np.uint64)
    rec = idhmc.xchg_accumulate(idhmc.XCHG_ACCEPT, ts["acceptance_rate"])
    cn[0], cn[1], cn[2] = len(ts), int(rec[0]), int(rec[1])
    maxd = (ts["term_left"] == 1) & (ts["term_right"] == 0)
    div = (ts["term_left"] == ts["term_right"])
    cn[3], cn[4], cn[5] = maxd.sum(), div.sum(), len(ts) - maxd.sum() - div.sum()
    cn[6:39] = np.bincount(np.minimum(ts["depth"], 32), minlength=33)
    bins = np.clip((ts["acceptance_rate"] * 1024).astype(np.int64), 0, 1023)
    cn[39:] = np.bincount(bins, minlength=1024)
    return cn


def test_summary_from_counters_matches_the_record_summary():
    import inplacedhmc_jl_amd as idhmc
    rng = np.random.default_rng(3)
    n = 5000
    ts = np.zeros(n, dtype=idhmc.TREE_STATS_DTYPE)
    ts["acceptance_rate"] = np.clip(rng.beta(5, 1.5, n), 0, 1)
    ts["acceptance_rate"][:7] = [0.0, 1.0, 1.0, 0.5, 2.0 ** -30, 1 - 2.0 ** -53, 0.25]
    ts["depth"] = rng.integers(0, 9, n)
    kind = rng.integers(0, 20, n)
    ts["term_left"] = np.where(kind == 0, 1, np.where(kind == 1, 5, -3))
    ts["term_right"] = np.where(kind == 0, 0, np.where(kind == 1, 5, 12))
    ref = idhmc.summarize_tree_statistics(ts)
    got = idhmc.summary_from_counters(_counters_from_records(idhmc, ts))
    assert got.N == ref.N and got.termination_counts == ref.termination_counts
    assert np.array_equal(got.depth_counts, ref.depth_counts)
    assert abs(got.a_mean - ref.a_mean) < 1e-15
    assert np.all(np.abs(got.a_quantiles - ref.a_quantiles) <= 1.0 / 1024 + 1e-12)
    assert str(got).splitlines()[0] == str(ref).splitlines()[0] and "termination:" in str(got)
    # counters of two shards add to the counters of the whole (what makes multi-rank summaries exact)
    a, b = _counters_from_records(idhmc, ts[:1234]), _counters_from_records(idhmc, ts[1234:])
    assert np.array_equal(a + b, _counters_from_records(idhmc, ts))
    assert idhmc.summary_from_counters(np.zeros_like(a)).N == 0
    with pytest.raises(ValueError):
        idhmc.summary_from_counters(a[:10])


def test_ess_from_moments_is_informative():
    """replicated batch means: recovers ESS/n = (1 - r) / (1 + r) of AR(1) chains, > 1 for antithetic ones (round 1 capped at 1)"""
    import inplacedhmc_jl_amd as idhmc
    rng = np.random.default_rng(11)
    Cn, n = 4000, 200
    for r in (0.6, 0.0, -0.5):
        x = np.empty((Cn, n))
        x[:, 0] = rng.standard_normal(Cn)
        for t in range(1, n):
            x[:, t] = r * x[:, t - 1] + np.sqrt(1 - r * r) * rng.standard_normal(Cn)
        mean, var = x.mean(axis=1)[:, None], x.var(axis=1, ddof=1)[:, None]
        e = idhmc.ess_from_moments(mean, var, n)[0] / (Cn * n)
        expect = (1 - r) / (1 + r)
        assert abs(e / expect - 1.0) < 0.08, (r, e, expect)
        assert idhmc.ess_from_moments(mean, var, n, cap=True)[0] <= Cn * n


@pytest.mark.gpu
def test_device_diagnostics_match_numpy_on_the_stored_records(idhmc):
    D, C, N = 40, 96, 60
    mu, sig = np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(max_depth=5), seed=8)
    eng.random_position()
    eng.set_eps(0.09)                      # some trees reach max_depth = 5, most turn
    eng.diag_reset()
    _, stats = eng.mcmc(N, 0, store_draws=False)            # [N][C] records
    cn = eng.diag_counters()
    assert np.array_equal(cn, _counters_from_records(idhmc, stats))          # integer for integer
    dev, ref = eng.tree_summary(), idhmc.summarize_tree_statistics(stats)
    assert dev.N == N * C and dev.termination_counts == ref.termination_counts and ref.termination_counts["max_depth"] > 0
    assert np.array_equal(dev.depth_counts, ref.depth_counts)
    assert abs(dev.a_mean - ref.a_mean) < 1e-14 and np.all(np.abs(dev.a_quantiles - ref.a_quantiles) <= 1 / 1024 + 1e-12)
    e_dev, e_ref = eng.ebfmi(), idhmc.EBFMI(stats.T)        # per chain
    assert np.allclose(e_dev, e_ref, rtol=1e-10, atol=0)    # shifted running sums vs numpy two-pass: 1e-10
    # a second window after a reset; and the flag form for callers that drive transitions themselves
    eng.diag_reset()
    for it in range(N + 1, N + 6):
        eng.nuts_transition(it, idhmc.T_ACCUM_DIAG)
    assert eng.diag_counters()[0] == 5 * C
    eng.close()
