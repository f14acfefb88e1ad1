"""Edge cases on the GPU path (through the C ABI), each against the oracle bit for bit: smallest and
largest dimension, a single chain, chain counts that do not fill a workgroup, non-finite inputs
(-Inf sentinel semantics, src/kinetic_energy.jl:80-84,107-112), divergence at the first leaf,
max_depth reached, injected directions / kept momentum (reference kwargs, src/NUTS.jl:251-258),
per-chain eps and metric, argument errors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64)) or (np.isnan(a) == np.isnan(b)).all() and \
        np.array_equal(np.nan_to_num(a, nan=0.0).view(np.uint64), np.nan_to_num(b, nan=0.0).view(np.uint64))


def pair(idhmc, oracle, D, C, seed, **kw):
    eng = idhmc.Engine(idhmc.IsoGaussian(D), C, idhmc.default_options(**kw), seed=seed)
    chains = [oracle.OracleChain(oracle.OracleModel.iso(D), oracle.default_options(**kw), seed=seed, chain_id=c) for c in range(C)]
    return eng, chains


@pytest.mark.parametrize("D,C", [(1, 1), (1, 5), (2, 3), (127, 2), (128, 9), (129, 7), (1024, 1), (1023, 3)])
def test_shapes_from_one_dim_one_chain_to_the_maximum(idhmc, oracle, D, C):
    eng, chains = pair(idhmc, oracle, D, C, seed=4, max_depth=6)
    eng.random_position()
    eng.set_eps(0.3)
    for ch in chains:
        ch.random_position()
    for it in (1, 2, 3):
        eng.nuts_transition(it)
        st = eng.tree_stats()
        ost = [ch.sample_tree(0.3, it) for ch in chains]
        assert st["steps"].tolist() == [s.steps for s in ost] and st["depth"].tolist() == [s.depth for s in ost]
        assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    eng.refresh_momentum(9)          # the momentum is unspecified after a transition until it is refreshed
    eng.leapfrog(0.1, 1)
    for ch in chains:
        ch.rand_p(9)
        ch.leapfrog(0.1)
    assert same_bits(eng.p, np.stack([c.p[:D] for c in chains])) and same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert eng.q.shape == (C, D) and eng.padded_dim() == (D + 127) // 128 * 128


def test_dimension_limit_is_an_argument_error(idhmc):
    with pytest.raises(idhmc.IdhmcError) as e:
        idhmc.Engine(idhmc.IsoGaussian(2049), 2)
    assert e.value.code == 1
    with pytest.raises(idhmc.IdhmcError):
        idhmc.Engine(idhmc.IsoGaussian(8), 2, idhmc.default_options(max_depth=16))
    eng = idhmc.Engine(idhmc.IsoGaussian(8), 2)
    with pytest.raises(ValueError):
        eng.set_q(np.zeros((3, 8)))
    with pytest.raises(idhmc.IdhmcError):
        eng.set_eps(-1.0)
    with pytest.raises(idhmc.IdhmcError):
        eng.set_minv(np.zeros(8))
    with pytest.raises(idhmc.IdhmcError):
        eng.leapfrog(0.1, 0)


def test_nonfinite_inputs_are_data_not_errors(idhmc, oracle):
    D, C = 16, 6
    eng, chains = pair(idhmc, oracle, D, C, seed=2)
    q = np.tile(np.linspace(-1, 1, D), (C, 1))
    q[1, 3] = np.inf
    q[2, 0] = np.nan
    q[3, 5] = -np.inf
    eng.set_q(q)
    p = np.ones((C, D))
    p[4, 2] = np.inf
    p[5, 7] = np.nan
    eng.set_p(p)
    for c, ch in enumerate(chains):
        ch.set_q(q[c])
        ch.set_p(p[c])
    lq = eng.lq
    assert lq[1] == -np.inf and lq[2] == -np.inf and lq[3] == -np.inf and np.isfinite(lq[0])
    assert same_bits(lq, [c.lq for c in chains])
    pi = eng.logdensity()
    assert (pi[1:] == -np.inf).all() and np.isfinite(pi[0])                    # K in {Inf, NaN} -> -Inf too
    assert same_bits(pi, [c.logdensity() for c in chains])
    # a transition from a -Inf start: every leaf has Delta = NaN or -Inf; no error, termination records it
    eng.set_eps(0.2)
    eng.nuts_transition(1)
    st = eng.tree_stats()
    ost = [ch.sample_tree(0.2, 1) for ch in chains]
    assert st["depth"].tolist() == [s.depth for s in ost] and st["steps"].tolist() == [s.steps for s in ost]
    assert st["term_left"].tolist() == [s.term_left for s in ost]


def test_divergence_and_max_depth_terminations(idhmc, oracle):
    D, C = 64, 8
    mu, sig = np.sin(np.arange(D, dtype=float)), np.logspace(-1, 1, D)
    eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(max_depth=3), seed=8)
    chains = [oracle.OracleChain(oracle.OracleModel.diag(mu, 1 / sig ** 2), oracle.default_options(max_depth=3), seed=8, chain_id=c)
              for c in range(C)]
    eng.random_position()
    for ch in chains:
        ch.random_position()
    q0 = eng.q
    eng.set_eps(1e3)                                              # every first leaf diverges
    eng.nuts_transition(1)
    st = eng.tree_stats()
    ost = [ch.sample_tree(1e3, 1) for ch in chains]
    assert (st["depth"] == 0).all() and (st["steps"] == 1).all() and (st["term_left"] == st["term_right"]).all()
    assert (np.abs(st["term_left"]) == 1).all() and (st["acceptance_rate"] == 0).all()
    assert same_bits(eng.q, q0) and st["term_left"].tolist() == [s.term_left for s in ost]
    eng.set_eps(1e-4)                                             # never turns within depth 3
    eng.nuts_transition(2)
    st = eng.tree_stats()
    ost = [ch.sample_tree(1e-4, 2) for ch in chains]
    assert (st["depth"] == 3).all() and (st["steps"] == 7).all()
    assert (st["term_left"] == 1).all() and (st["term_right"] == 0).all()          # REACHED_MAX_DEPTH, src/tree.jl:300
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert same_bits(st["pi"], [s.pi for s in ost])


def test_injected_directions_and_kept_momentum(idhmc, oracle):
    """reference test affordances: sample_tree(...; p = ..., directions = ...) (src/NUTS.jl:251-258)"""
    D, C = 32, 5
    eng, chains = pair(idhmc, oracle, D, C, seed=6, max_depth=7)
    eng.random_position()
    p = np.random.default_rng(0).standard_normal((C, D))
    eng.set_p(p)
    eng.set_eps(0.25)
    dirs = np.array([0xffffffff, 0x0, 0b0101, 0b1010, 0x12345678], dtype=np.uint32)
    eng.nuts_transition(1, flags=idhmc.T_KEEP_P, directions=dirs)
    st = eng.tree_stats()
    for c, ch in enumerate(chains):
        ch.random_position()
        ch.set_p(p[c])
        s = ch.sample_tree(0.25, 1, directions=int(dirs[c]), refresh_p=False)
        assert (st[c]["depth"], st[c]["steps"], st[c]["term_left"], st[c]["term_right"]) == (s.depth, s.steps, s.term_left, s.term_right)
        assert st[c]["pi"] == s.pi and st[c]["acceptance_rate"] == s.acceptance_rate
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert st[0]["term_left"] == 0 and st[1]["term_right"] == 0          # all-forward / all-backward trees


def test_per_chain_eps_and_metric(idhmc, oracle):
    D, C = 48, 6
    eng, chains = pair(idhmc, oracle, D, C, seed=10, max_depth=7)
    rng = np.random.default_rng(1)
    minv = rng.uniform(0.2, 3.0, (C, D))
    eps = rng.uniform(0.05, 0.4, C)
    eng.set_minv(minv)
    eng.set_eps(eps)
    eng.random_position()
    for c, ch in enumerate(chains):
        ch.set_minv(minv[c])
        ch.random_position()
    for it in (1, 2):
        eng.nuts_transition(it)
        ost = [ch.sample_tree(eps[c], it) for c, ch in enumerate(chains)]
        assert eng.tree_stats()["steps"].tolist() == [s.steps for s in ost]
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert same_bits(eng.minv, minv) and same_bits(eng.eps, eps)
    eng.refresh_momentum(3)
    eng.leapfrog(None, 2)
    for c, ch in enumerate(chains):
        ch.rand_p(3)
        ch.leapfrog(eps[c]); ch.leapfrog(eps[c])
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))


def test_eps_underflow_is_reported(idhmc):
    """dual averaging driving eps below 1e-10 raises the reference's assertion (src/warmup.jl:291-296) -- and, like the
    reference, promptly: the driver polls the device's abort word with a lag of 8 launches instead of finishing the stage"""
    D, C = 8, 2

    def fresh():
        eng = idhmc.Engine(idhmc.IsoGaussian(D), C, idhmc.default_options(max_depth=2, da_gamma=1e-9), seed=1)
        eng.set_q(np.full((C, D), 1e6))
        eng.set_eps(1.0)
        return eng
    # by hand: the transition after which the abort word is set
    eng = fresh()
    eng.da_init()
    k = None
    for it in range(1, 60):
        eng.nuts_transition(it, idhmc.T_ADAPT_EPS)
        if eng.poll_abort(0) != 0:
            k = it
            break
    assert k is not None and eng.poll_abort(0) == 3
    eng.close()
    # the driver: N = 200 would take >= 200 * C leapfrogs; it must stop within 8 transitions of transition k
    eng = fresh()
    with pytest.raises(idhmc.IdhmcError) as e:
        eng.tuning_stage(200, False, 0)
    assert e.value.code == 3 and "1e-10" in str(e.value)
    assert eng.total_steps() <= (k + 9) * C * 3 < 200 * C
    assert eng.poll_abort(0) == 0                    # reported once, then cleared
    eng.close()


def test_placement_probe_does_not_change_results(idhmc, monkeypatch):
    """contexts with large state arrays try several placements of q, p, grad at creation and keep the one whose probe is
    fastest: same bits as a context that takes the first placement, same footprint, nothing left allocated"""
    import numpy as np
    D, C = 1024, 12288                      # 96 MiB per array: above the 64 MiB threshold
    sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=np.float64))
    out = {}
    for tries in ("1", "32"):
        monkeypatch.setenv("IDHMC_PLACEMENT_TRIES", tries)
        eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), C, idhmc.default_options(metric_mode=idhmc.METRIC_PER_CHAIN), seed=3)
        eng.random_position(); eng.refresh_momentum(1)
        eng.leapfrog(0.05, 1); eng.leapfrog(0.05, 3)
        eng.set_eps(0.2); eng.nuts_transition(2)
        gbps, ncand = eng.placement_info()
        assert (ncand == 1 and gbps == 0.0) if tries == "1" else (1 <= ncand <= 48 and gbps > 1000.0)
        out[tries] = (eng.q, eng.p, eng.grad, eng.lq, eng.minv, eng.device_bytes())
        eng.close()
    for a, b in zip(out["1"][:5], out["32"][:5]):
        assert np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))
    assert 0 <= out["32"][5] - out["1"][5] <= 3 * (2050 << 20)      # the spread-out placement, when it is the one kept, holds its gaps too


def test_create_destroy_returns_all_device_memory(idhmc):
    """contexts own their device memory (include/idhmc.h): after idhmc_destroy nothing stays allocated -- including the placement
    candidates that were not kept and the lanes' streams of the dense leapfrog"""
    import numpy as np
    import torch
    D = 256
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = (Q / np.logspace(-1, 0, D)) @ Q.T
    P = 0.5 * (P + P.T)
    sig = np.logspace(-1, 1, 1024)

    def cycle():
        e1 = idhmc.Engine(idhmc.DiagGaussian(np.zeros(1024), sigma=sig), 12288, idhmc.default_options(metric_mode=idhmc.METRIC_PER_CHAIN), seed=1)
        e1.random_position(); e1.refresh_momentum(1); e1.leapfrog(0.05, 1); e1.set_eps(0.2); e1.nuts_transition(1)
        e1.close()
        e2 = idhmc.Engine(idhmc.DenseMVN(np.zeros(D), P), 12288 + 5, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED), seed=1)
        e2.random_position(); e2.refresh_momentum(1)
        for _ in range(3):
            e2.leapfrog(0.01, 1)               # lanes open when the context goes away
        e2.close()
    cycle()                                    # first use: the runtime's own pools (streams, events, code objects) are up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "device memory grew by %.1f MiB over 5 create/destroy cycles" % ((free0 - free1) / 2**20)
