"""CPU-side checks of the drop-in boundary: libidhmc.so loads, exports every symbol include/idhmc.h
declares, struct layouts and defaults match the reference, and the product fails loudly (no CPU
fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "idhmc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(idhmc_[a-z0-9_]+)\s*\(", text)) - {"idhmc_allreduce_fn"})


def test_library_exports_every_declared_symbol(idhmc):
    lib = idhmc.load_library()
    declared = header_symbols()
    assert len(declared) >= 45
    from inplacedhmc_jl_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared, "ctypes table and header disagree"
    for name in declared:
        assert hasattr(lib, name), "libidhmc.so does not export %s" % name


def test_library_version_is_the_headers(idhmc):
    """what __graft_entry__.build() asserts: the shared library was built from this header"""
    text = open(os.path.join(ROOT, "include", "idhmc.h")).read()
    assert idhmc.load_library().idhmc_version() == int(re.search(r"#define\s+IDHMC_VERSION\s+(\d+)", text).group(1)) == 4


def test_struct_layouts(idhmc):
    from inplacedhmc_jl_amd import _lib
    assert idhmc.TREE_STATS_DTYPE.itemsize == 32            # TreeStatisticsNUTS is 32 bytes, src/NUTS.jl:229
    assert [idhmc.TREE_STATS_DTYPE.fields[f][1] for f in ("pi", "acceptance_rate", "term_left", "term_right", "depth", "steps")] == \
        [0, 8, 16, 20, 24, 28]
    assert C.sizeof(_lib.ModelDesc) == 56
    assert C.sizeof(_lib.Options) == 144


def test_default_options_match_reference(idhmc):
    o = idhmc.default_options()
    assert (o.max_depth, o.min_delta) == (10, -1000.0)                               # src/NUTS.jl:214
    assert (o.da_delta, o.da_gamma, o.da_kappa, o.da_t0) == (0.8, 0.05, 0.75, 10)    # src/stepsize.jl:191
    assert (o.ss_a_min, o.ss_a_max, o.ss_eps0, o.ss_C) == (0.25, 0.75, 1.0, 2.0)     # src/stepsize.jl:29
    assert (o.ss_maxiter_crossing, o.ss_maxiter_bisect) == (400, 400)
    assert (o.init_steps, o.middle_steps, o.doubling_stages, o.terminating_steps) == (75, 25, 5, 50)  # src/warmup.jl:366
    assert (o.eps_mode, o.metric_mode) == (idhmc.EPS_PER_CHAIN, idhmc.METRIC_PER_CHAIN)
    with pytest.raises(AttributeError):
        idhmc.default_options(no_such_field=1)


def test_api_surface_names_and_defaults(idhmc):
    """exports of src/InplaceDHMC.jl:3-11 and the option structs' defaults"""
    for name in ("GaussianKineticEnergy", "NoProgressReport", "LogProgressReport", "TuningNUTS", "mcmc_with_warmup",
                 "threaded_mcmc", "default_warmup_stages", "NUTS", "DualAveraging", "InitialStepsizeSearch",
                 "FixedStepsize", "FindLocalOptimum", "fixed_stepsize_warmup_stages", "EBFMI",
                 "summarize_tree_statistics", "TreeStatisticsNUTS"):
        assert hasattr(idhmc, name), name
    st = idhmc.default_warmup_stages()
    assert [len(s) for s in st] == [0, 0, 75, 25, 50, 100, 200, 400, 50]             # src/warmup.jl:361-372
    assert [getattr(s, "M", None) for s in st[2:]] == ["Nothing"] + ["Diagonal"] * 5 + ["Nothing"]
    assert idhmc.num_stored(100, st) == 400 and idhmc.num_stored(1000, st) == 1000   # src/mcmc.jl:115-116
    fs = idhmc.fixed_stepsize_warmup_stages()
    assert [len(s) for s in fs] == [0, 25, 50, 100, 200, 400]
    assert st[4].lam is None and idhmc.TuningNUTS(25).N == 25
    with pytest.raises(ValueError):
        idhmc.NUTS(max_depth=0)
    with pytest.raises(ValueError):
        idhmc.NUTS(min_delta=1.0)
    with pytest.raises(ValueError):
        idhmc.TuningNUTS(25, M="Symmetric")
    k = idhmc.GaussianKineticEnergy.identity(4, 0.25)
    assert np.all(k.minv == 0.25) and np.all(k.W == 2.0)                              # src/hamiltonian.jl:63-74


def test_model_validation(idhmc):
    with pytest.raises(ValueError):
        idhmc.Model(1, 8, mu=np.zeros(7), tau=np.ones(8))
    m = idhmc.DiagGaussian(np.zeros(5), sigma=np.full(5, 2.0))
    assert np.all(m.tau == 0.25) and m.D == 5
    assert idhmc.DenseMVN(np.zeros(3), np.eye(3)).prec.shape == (3, 3)


def test_fails_loudly_without_gpu(idhmc):
    """No CPU path: with no device the engine refuses to exist and says why."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(idhmc.IdhmcError) as e:
        idhmc.Engine(idhmc.IsoGaussian(32), 4)
    assert e.value.code == 6 and "no HIP device" in str(e.value)
    with pytest.raises(idhmc.IdhmcError):
        idhmc.threaded_mcmc(idhmc.IsoGaussian(8), 10, nchains=2)


def test_missing_library_is_an_import_error(idhmc, tmp_path, monkeypatch):
    from inplacedhmc_jl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libidhmc.so"))
    with pytest.raises(ImportError) as e:
        _lib.load()
    assert "no CPU fallback" in str(e.value)


def test_argument_errors_do_not_need_a_gpu(idhmc):
    lib = idhmc.load_library()
    from inplacedhmc_jl_amd import _lib
    h = C.c_void_p()
    desc = idhmc.IsoGaussian(3000).desc()
    rc = lib.idhmc_create(C.byref(h), 0, 4, 0, C.byref(desc), None, 1)
    assert rc == 1 and b"unsupported" in lib.idhmc_last_error()            # D > 2048: IDHMC_ERR_BAD_ARG
    desc = idhmc.IsoGaussian(8).desc()
    assert lib.idhmc_create(C.byref(h), 0, 0, 0, C.byref(desc), None, 1) == 1
    o = idhmc.default_options(max_depth=40)
    assert lib.idhmc_create(C.byref(h), 0, 4, 0, C.byref(desc), C.byref(o), 1) == 1
    assert lib.idhmc_set_eps(None, 0.1) == 1 and b"null context" in lib.idhmc_last_error()
    assert lib.idhmc_destroy(None) == 0


def test_diagnostics_match_reference_formulas(idhmc):
    rng = np.random.default_rng(0)
    ts = np.zeros(500, dtype=idhmc.TREE_STATS_DTYPE)
    ts["pi"] = rng.standard_normal(500).cumsum() * 0.1 + rng.standard_normal(500)
    ts["acceptance_rate"] = rng.uniform(0.5, 1, 500)
    ts["depth"] = rng.integers(1, 5, 500)
    ts["term_left"], ts["term_right"] = -3, 4
    ts["term_left"][:10], ts["term_right"][:10] = 1, 0          # REACHED_MAX_DEPTH
    ts["term_left"][10:15], ts["term_right"][10:15] = 2, 2      # divergence
    pis = ts["pi"]
    assert idhmc.EBFMI(ts) == pytest.approx(np.mean(np.diff(pis) ** 2) / np.var(pis, ddof=1))   # src/diagnostics.jl:28-32
    s = idhmc.summarize_tree_statistics(ts)
    assert s.termination_counts == {"max_depth": 10, "divergence": 5, "turning": 485}
    assert s.depth_counts.sum() == 500 and s.depth_counts[0] == 0 and len(s.a_quantiles) == 5
    assert "acceptance rate mean" in str(s)
    x = rng.standard_normal((4000, 3))
    e = idhmc.ess(x)
    assert np.all((e > 3000) & (e < 5000))
    ar = np.zeros(4000)
    for i in range(1, 4000):
        ar[i] = 0.9 * ar[i - 1] + rng.standard_normal()
    assert 100 < idhmc.ess(ar)[0] < 400                          # tau = (1+r)/(1-r) = 19


def test_shard_range(idhmc):
    sr = idhmc.distributed.shard_range
    for total, world in ((524288, 8), (10, 3), (7, 8), (65536, 1)):
        parts = [sr(total, r, world) for r in range(world)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == total
        for (f0, c0), (f1, _) in zip(parts, parts[1:]):
            assert f0 + c0 == f1
    assert sr(524288, 3, 8) == (196608, 65536)                   # cfg5: chain id -> GPU = id // 65536
    with pytest.raises(ValueError):
        sr(8, 8, 8)


def test_moment_diagnostics(idhmc):
    """R-hat / ESS from per-chain running moments (what a 65 536-chain run can afford to bring to the host)"""
    rng = np.random.default_rng(0)
    C, n, D = 400, 50, 3
    rho = 0.6                                                        # AR(1) chains: ESS per draw = (1 - rho) / (1 + rho)
    x = np.empty((C, n, D))
    x[:, 0] = rng.standard_normal((C, D))
    for t in range(1, n):
        x[:, t] = rho * x[:, t - 1] + np.sqrt(1 - rho ** 2) * rng.standard_normal((C, D))
    mean, var = x.mean(axis=1), x.var(axis=1, ddof=1)
    r = idhmc.rhat_from_moments(mean, var, n)
    assert r.shape == (D,) and np.all(np.abs(r - 1.0) < 0.1)
    e = idhmc.ess_from_moments(mean, var, n)
    assert np.all(e > 0.5 * C * n * (1 - rho) / (1 + rho)) and np.all(e < 2.0 * C * n * (1 - rho) / (1 + rho))
    r_bad = idhmc.rhat_from_moments(mean + np.arange(C)[:, None] % 2 * 2.0, var, n)     # two clusters of chains
    assert np.all(r_bad > 1.3)
