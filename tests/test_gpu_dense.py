"""BASELINE.json configs[3] family: dense multivariate normal (general, non-separable density).
GPU path (per-wave GEMV from L2 inside every kernel) against the CPU oracle, bit-exact fp64: the engine
defines the gradient's summation order (ascending column, one fma chain per row) and both sides follow it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def dense_problem(D, seed=7):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    lam = np.logspace(-2, 0, D)                      # Sigma eigenvalues
    P = (Q / lam) @ Q.T
    P = 0.5 * (P + P.T)
    mu = np.cos(np.arange(D, dtype=np.float64))
    return mu, P


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("D", [40, 128, 256])
def test_dense_eval_leapfrog_and_search(idhmc, oracle, D):
    C = 6
    mu, P = dense_problem(D)
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, seed=21)
    om = oracle.OracleModel.dense(mu, P)
    chains = [oracle.OracleChain(om, seed=21, chain_id=c) for c in range(C)]
    eng.random_position()
    for ch in chains:
        ch.random_position()
    assert same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))
    assert same_bits(eng.lq, [c.lq for c in chains])
    q = eng.q[0]
    assert abs(eng.lq[0] + 0.5 * (q - mu) @ P @ (q - mu)) < 1e-9 * abs(eng.lq[0])
    eng.refresh_momentum(1)
    eng.leapfrog(0.01, 3)
    eng.leapfrog(-0.01, 1)
    for ch in chains:
        ch.rand_p(1)
        for _ in range(3):
            ch.leapfrog(0.01)
        ch.leapfrog(-0.01)
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert same_bits(eng.p, np.stack([c.p[:D] for c in chains]))
    assert same_bits(eng.logdensity(), [c.logdensity() for c in chains])
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()
    ref = []
    for ch in chains:
        ch.rand_p(0)
        rc, e = ch.find_initial_stepsize()
        assert rc == 0
        ref.append(e)
    assert same_bits(eng.eps, ref)


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("D,eps,C", [(40, 0.05, 6), (256, 0.02, 6), (256, 0.02, 37), (128, 0.03, 16), (256, 0.02, 100)])
def test_dense_nuts_transitions(idhmc, oracle, D, eps, C, shared):
    """L <= 256 runs the workgroup-cooperative matrix-core gradient (16 chains per workgroup, ragged last group), with a per-chain and
    with a shared metric (two instantiations of the kernel).  The unit metric is the same metric shared or per chain, so the oracle's
    chains are the reference for both"""
    T = 12
    mu, P = dense_problem(D)
    opt = idhmc.default_options(max_depth=8, metric_mode=idhmc.METRIC_SHARED) if shared else idhmc.default_options(max_depth=8)
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, opt, seed=5)
    om = oracle.OracleModel.dense(mu, P)
    chains = [oracle.OracleChain(om, oracle.default_options(max_depth=8), seed=5, chain_id=c) for c in range(C)]
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


def test_dense_full_warmup_matches_oracle(idhmc, oracle):
    D, C, N = 64, 4, 15
    mu, P = dense_problem(D)
    short = dict(init_steps=15, middle_steps=10, doubling_stages=2, terminating_steps=10, max_depth=8)
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, idhmc.default_options(**short), seed=77)
    draws, stats = eng.mcmc_with_warmup(N)
    rc, och, ost, oeps = oracle.threaded_mcmc(oracle.OracleModel.dense(mu, P), N, C, oracle.default_options(**short), seed=77)
    assert rc == 0 and same_bits(eng.eps, oeps)
    for n in range(N):
        assert same_bits(draws[n], och[:, n, :D])
    assert np.array_equal(stats.T, ost[:, :N])


@pytest.mark.parametrize("D,C", [(256, 70), (100, 33), (256, 256), (500, 21)])
def test_dense_mfma_single_step_kernel(idhmc, oracle, D, C, monkeypatch):
    """the matrix-core kernel (16-chain tiles, columns split over 4 wavefronts, n steps per launch): ragged chain
    counts, per-chain eps and M^-1, against the oracle AND against the per-wave GEMV kernel -- all bit-identical"""
    mu, P = dense_problem(D, seed=3)
    rng = np.random.default_rng(0)
    minv = rng.uniform(0.5, 2.0, (C, D))
    eps = rng.uniform(0.005, 0.02, C)
    out = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("IDHMC_DENSE_MFMA", mode)
        eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, seed=9)
        eng.set_minv(minv)
        eng.random_position()
        eng.refresh_momentum(4)
        eng.set_eps(eps)
        eng.leapfrog(None, 1)
        eng.leapfrog(None, 1)
        eng.leapfrog(None, 3)        # one launch, state stays on chip between the steps
        out[mode] = (eng.q, eng.p, eng.grad, eng.lq, eng.logdensity())
        eng.close()
    for other in ("0",):
        for a, b in zip(out["2"], out[other]):
            assert same_bits(a, b)
    om = oracle.OracleModel.dense(mu, P)
    for c in (0, C // 2, C - 1):
        ch = oracle.OracleChain(om, seed=9, chain_id=c)
        ch.set_minv(minv[c])
        ch.random_position()
        ch.rand_p(4)
        for _ in range(5):
            ch.leapfrog(eps[c])
        assert same_bits(out["2"][0][c], ch.q[:D]) and same_bits(out["2"][1][c], ch.p[:D])
        assert out["2"][3][c] == ch.lq and out["2"][4][c] == ch.logdensity()


@pytest.mark.parametrize("D,C", [(256, 16 * 256 * 3 + 5), (100, 16 * 256 * 4 + 16 * 7 + 1)])
def test_dense_single_step_lanes(idhmc, oracle, D, C, monkeypatch):
    """from 768 tiles on, a single-step sweep of the dense density is cut into 3-4 lanes (contiguous ranges of 16-chain tiles,
    one stream each) that fork from the context's stream and join it at the next call of any other kind: the same bits as one
    kernel on one stream and as the oracle, whatever is interleaved with the sweeps (getters, n-step launches, a refresh)"""
    mu, P = dense_problem(D, seed=5)
    rng = np.random.default_rng(1)
    eps = rng.uniform(0.005, 0.02, C)
    out = {}
    for lanes in ("1", "0", "caller's stream"):
        monkeypatch.setenv("IDHMC_DENSE_LANES", "0" if lanes == "0" else "4")
        eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED), seed=13)
        if lanes == "caller's stream":             # one kernel per sweep on that stream (include/idhmc.h, idhmc_leapfrog)
            import torch
            ext = torch.cuda.Stream()
            eng.set_stream(ext.cuda_stream)
        eng.random_position()
        eng.refresh_momentum(2)
        eng.set_eps(eps)
        for _ in range(3):
            eng.leapfrog(None, 1)                  # back to back: lanes stay open
        mid = eng.q[C - 1].copy()                  # a getter joins them
        eng.leapfrog(0.01, 1)
        eng.leapfrog(0.01, 2)                      # the n-step kernel runs on the context's stream
        eng.leapfrog(-0.01, 1)
        eng.refresh_momentum(3)                    # another kernel on the context's stream, straight after open lanes
        eng.leapfrog(None, 1)
        out[lanes] = (eng.q, eng.p, eng.grad, eng.lq, eng.logdensity(), mid)
        eng.close()
    for other in ("0", "caller's stream"):
        for a, b in zip(out["1"], out[other]):
            assert same_bits(a, b)
    om = oracle.OracleModel.dense(mu, P)
    ntiles = (C + 15) // 16
    for c in (0, 16 * (ntiles // 3) - 1, 16 * (ntiles // 3), 16 * (ntiles // 2), C - 1):     # around the lane boundaries too
        ch = oracle.OracleChain(om, seed=13, chain_id=c)
        ch.random_position()
        ch.rand_p(2)
        for _ in range(3):
            ch.leapfrog(eps[c])
        assert same_bits(out["1"][5], ch.q[:D]) if c == C - 1 else True
        for e in (0.01, 0.01, 0.01, -0.01):
            ch.leapfrog(e)
        ch.rand_p(3)
        ch.leapfrog(eps[c])
        assert same_bits(out["1"][0][c], ch.q[:D]) and same_bits(out["1"][1][c], ch.p[:D])
        assert out["1"][3][c] == ch.lq and out["1"][4][c] == ch.logdensity()


def test_dense_requires_symmetric_precision(idhmc):
    P = np.eye(8); P[0, 1] = 0.1
    with pytest.raises(idhmc.IdhmcError) as e:
        idhmc.Engine(idhmc.DenseMVN(np.zeros(8), P), 2)
    assert e.value.code == 1 and "symmetric" in str(e.value)


@pytest.mark.parametrize("D,C,md,eps", [(1, 1, 3, 0.5), (3, 15, 1, 0.2), (17, 17, 2, 0.1), (200, 33, 4, 0.03), (129, 16, 6, 0.05),
                                        (256, 1, 5, 0.02)])
@pytest.mark.parametrize("shared", [False, True])
def test_dense_nuts_edge_shapes(idhmc, oracle, D, C, md, eps, shared):
    """cooperative gradient at the edges: a single chain, ragged groups (15, 17, 33 chains), tiny dimensions, shallow
    trees, a divergent-prone stepsize -- every wavefront of a workgroup must still meet at every barrier (both forms of the kernel)"""
    mu, P = dense_problem(D, seed=11)
    opt = idhmc.default_options(max_depth=md, metric_mode=idhmc.METRIC_SHARED) if shared else idhmc.default_options(max_depth=md)
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, opt, seed=3)
    om = oracle.OracleModel.dense(mu, P)
    chains = [oracle.OracleChain(om, oracle.default_options(max_depth=md), seed=3, chain_id=c) for c in range(C)]
    eng.random_position()
    eng.set_eps(eps)
    for ch in chains:
        ch.random_position()
    for it in range(1, 5):
        eng.nuts_transition(it)
        st = eng.tree_stats()
        ost = [ch.sample_tree(eps, it) for ch in chains]
        for f in ("depth", "steps", "term_left", "term_right"):
            np.testing.assert_array_equal(st[f], [getattr(s, f) for s in ost], err_msg="%s @%d" % (f, it))
        assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))


def test_lanes_sit_on_distinct_hardware_queues_whatever_came_before(idhmc):
    """Round 3: lanes overlap only on different hardware queues, and which queue a new stream gets depends on every stream the process made
    before (the runtime multiplexes streams onto four queues, least-used first).  The library therefore picks its lane streams with an
    idle-kernel test (idhmc_lanes_info).  Here the process first makes and keeps an odd number of other streams -- the situation in which
    round 3's bench line read 63 instead of 48 us per sweep -- and the context must still report four lanes on four queues."""
    import torch
    D, C = 256, 16384
    mu, P = dense_problem(D, seed=5)
    others = [torch.cuda.Stream() for _ in range(3)]            # they stay alive: their queues stay referenced
    for st in others:
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()
    eng = idhmc.Engine(idhmc.DenseMVN(mu, P), C, idhmc.default_options(metric_mode=idhmc.METRIC_SHARED), seed=13)
    assert eng.lanes_info() == (0, 0)                            # nothing chosen before the first single-step sweep
    eng.random_position()
    eng.refresh_momentum(1)
    eng.time_leapfrog(0.02, 20)
    lanes, distinct = eng.lanes_info()
    assert lanes == 4 and distinct == 4, (lanes, distinct)
    ms = min(eng.time_leapfrog(0.02, 200) for _ in range(3))
    assert ms < 0.058, ms                                        # four queues: 0.048-0.050 ms; three: 0.062-0.064 ms
    eng.close()
    del others
