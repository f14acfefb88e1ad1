"""The N>1 path on a real device: two processes (ranks) share the box's one GPU, each owning half of the chains, and run
the global-eps warm-up -- per-chain stepsize searches pooled into one eps, then a tuning stage -- through the library's
own drivers with the exchange wired to torch.distributed (gloo here: RCCL refuses two ranks on one device; on a
multi-GPU node the same call sequence runs over "nccl" = RCCL, or over the library's communicator, idhmc_comm_init).
The exchange record is integer-valued (include/idhmc.h), so the checks are `==`, no tolerance: every rank holds the same
eps after every step; the two-rank run equals the one-rank run of all chains, eps and draws, bit for bit; and both
equal the oracle's chains driven through the same protocol restated with Python integers."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, TOTAL, N, SEED = 64, 25, 10, 11          # 25 chains: ranks own 13 and 12


def _problem():
    return np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)


def _stage(pkg, first, count, attach):
    mu, sig = _problem()
    opt = pkg.default_options(max_depth=6, eps_mode=pkg.EPS_GLOBAL)
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), count, opt, seed=SEED, first_chain=first)
    keep = attach(eng)
    eng.random_position()
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()              # global mode: exp(pooled mean log eps), exchanged
    eps0 = eng.eps.copy()
    draws, stats = eng.tuning_stage(N, False, 0, store_draws=True)
    eps = eng.eps.copy()
    eng.close()
    del keep
    return eps0, eps, draws


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inplacedhmc_jl_amd as pkg
    first, count = pkg.distributed.shard_range(TOTAL, rank, world)
    eps0, eps, draws = _stage(pkg, first, count, lambda eng: pkg.distributed.attach_global_eps(eng))
    np.savez(os.path.join(out, "rank%d.npz" % rank), eps0=eps0, eps=eps, draws=draws, first=first)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_one_rank_bit_for_bit(idhmc, oracle, tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert len(r0["eps"]) == 13 and len(r1["eps"]) == 12
    for k in ("eps0", "eps"):                                       # one global eps, on both ranks
        assert np.all(r0[k] == r0[k][0]) and np.all(r1[k] == r0[k][0])
    eps0, eps, draws = _stage(idhmc, 0, TOTAL, lambda eng: None)
    assert eps0[0] == r0["eps0"][0] and eps[0] == r0["eps"][0]       # ==, not allclose
    both = np.concatenate([r0["draws"], r1["draws"]], axis=1)        # [N][chains][D]
    assert np.array_equal(both, draws)

    # the oracle's chains through the same protocol
    O = oracle
    mu, sig = _problem()
    om = O.OracleModel.diag(mu, 1.0 / sig ** 2)
    oopt = O.default_options(max_depth=6)
    chains = [O.OracleChain(om, oopt, seed=SEED, chain_id=c) for c in range(TOTAL)]
    for ch in chains:
        ch.random_position()
        ch.rand_p(0)
    e0 = O.global_initial_eps(chains)
    assert e0 == eps0[0]
    used, efin = O.global_eps_stage(chains, N, 0, e0, oopt)
    assert efin == eps[0]
    assert np.array_equal(np.stack([ch.q[:D] for ch in chains]), draws[-1])


def _failing_worker(rank, world, port, out):
    """rank 1 owns a chain whose start has a non-finite density: its stepsize search fails (src/stepsize.jl:152-153)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inplacedhmc_jl_amd as pkg
    mu, sig = _problem()
    first, count = pkg.distributed.shard_range(TOTAL, rank, world)
    res = []
    for mode in ("search", "stage"):
        if mode == "search":
            opt = pkg.default_options(max_depth=6, eps_mode=pkg.EPS_GLOBAL)
        else:       # per-chain dual averaging; rank 1's is set up to collapse (as tests/test_gpu_edges.py::test_eps_underflow_is_reported)
            opt = pkg.default_options(max_depth=3, eps_mode=pkg.EPS_PER_CHAIN, da_gamma=1e-9 if rank == 1 else 0.05)
        eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), count, opt, seed=SEED, first_chain=first)
        keep = pkg.distributed.attach_global_eps(eng)
        eng.random_position()
        eng.refresh_momentum(0)
        code = 0
        try:
            if mode == "search":
                if rank == 1:
                    q = eng.q
                    q[3, 0] = np.inf
                    eng.set_q(q)
                eng.find_initial_stepsize()
            else:           # rank 1's chains start far out with a dual averaging that drives eps below 1e-10 (src/warmup.jl:291-296)
                if rank == 1:
                    eng.set_q(np.full((count, D), 1e6))
                eng.set_eps(0.05)
                eng.tuning_stage(60, False, 0, store_stats=False)
        except pkg.IdhmcError as e:
            code = e.code
        res.append(code)
        eng.close()
        del keep
    np.save(os.path.join(out, "codes%d.npy" % rank), np.array(res))
    dist.barrier()
    dist.destroy_process_group()


def test_an_error_on_one_rank_fails_every_rank_together(idhmc, tmp_path):
    """the exchange is enqueued before any error is looked at, and the all-reduced slot [3] of the record tells every
    rank -- the owner returns its own code, the peers IDHMC_ERR_PEER; nobody is left waiting in a collective (the spawn would hang)"""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + 7
    mp.spawn(_failing_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "codes0.npy"), np.load(tmp_path / "codes1.npy")
    assert list(c0) == [idhmc.ERR_PEER, idhmc.ERR_PEER], c0
    assert list(c1) == [idhmc.ERR_NONFINITE_START, idhmc.ERR_EPS_UNDERFLOW], c1


def test_manual_exchange_between_two_contexts_in_one_process(idhmc):
    """The fine-grained exchange API (idhmc_find_initial_stepsize_per_chain, idhmc_logeps_sum, idhmc_set_eps_from_logeps,
    idhmc_accept_sum, idhmc_da_adapt_global) driven by hand: two contexts hold the two shards like two ranks would, the
    host adds their records (integers in doubles: torch on the device) and hands the total to both.  Must equal the
    library's own driver on one context holding all chains, bit for bit."""
    import torch
    mu, sig = _problem()
    opt = idhmc.default_options(max_depth=6, eps_mode=idhmc.EPS_GLOBAL)
    shards = [idhmc.distributed.shard_range(TOTAL, r, 2) for r in range(2)]
    engs = [idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), cnt, opt, seed=SEED, first_chain=first) for first, cnt in shards]
    recs = [torch.zeros(idhmc.XCHG_DOUBLES, dtype=torch.float64, device="cuda") for _ in engs]
    torch.cuda.synchronize()          # (torch's fill is on torch's stream; the engines write from their own)

    def exchange(fill):
        for e, r in zip(engs, recs):
            fill(e, r.data_ptr())
            e.synchronize()
        total = recs[0] + recs[1]                      # any order: the record is integer-valued
        torch.cuda.synchronize()
        for r in recs:
            r.copy_(total)
        torch.cuda.synchronize()

    for e in engs:
        e.random_position()
        e.refresh_momentum(0)
        e.find_initial_stepsize(per_chain_only=True)
    exchange(lambda e, ptr: e.logeps_sum(ptr))
    for e, r in zip(engs, recs):
        e.set_eps_from_logeps(r.data_ptr())
        e.da_init()
    eps0 = [e.eps[0] for e in engs]
    draws = []
    for it in range(1, N + 1):
        for e in engs:
            e.nuts_transition(it)
        exchange(lambda e, ptr: e.accept_sum(ptr))
        assert float(recs[0][2]) == TOTAL
        for e, r in zip(engs, recs):
            e.da_adapt_global(r.data_ptr())
        draws.append(np.concatenate([e.q for e in engs]))
    for e in engs:
        e.da_finalize()
    eps = [e.eps[0] for e in engs]
    for e in engs:
        e.close()
    ref0, ref, ref_draws = _stage(idhmc, 0, TOTAL, lambda eng: None)
    assert eps0[0] == eps0[1] == ref0[0] and eps[0] == eps[1] == ref[0]
    assert np.array_equal(np.stack(draws), ref_draws)


@pytest.mark.parametrize("split,exact", [(2048, True), (1024, True), (2000, False)])
def test_pooled_metric_does_not_depend_on_the_sharding(idhmc, split, exact):
    """IDHMC_METRIC_POOLED across shards: partial sums per segment of 1024 GLOBAL chain ids, added in segment order.  Two
    contexts (two "ranks" in one process) fill their rows of the table, the host adds the tables (every entry is non-zero on
    one side only when the shard boundary is a multiple of 1024), both consume the total: the metric equals the one a
    single context of all chains computes -- bit for bit for aligned shards, to rounding (1e-13) for an unaligned one."""
    import torch
    D, TOT, T = 16, 4096, 6
    mu, sig = np.sin(np.arange(D, dtype=np.float64)), np.logspace(-0.5, 0.5, D)
    opt = idhmc.default_options(max_depth=5, metric_mode=idhmc.METRIC_POOLED)

    def run(first, count):
        eng = idhmc.Engine(idhmc.DiagGaussian(mu, sigma=sig), count, opt, seed=3, first_chain=first)
        eng.random_position()
        eng.set_eps(0.3)
        eng.metric_begin()
        for it in range(1, T + 1):
            eng.nuts_transition(it, idhmc.T_ACCUM_METRIC)
        return eng

    whole = run(0, TOT)
    whole.metric_update(5.0 / T)
    ref = whole.minv[0].copy()
    whole.close()
    parts = [run(0, split), run(split, TOT - split)]
    L = parts[0].padded_dim()
    nseg = TOT // idhmc.POOL_SEGMENT
    tabs = [torch.zeros(nseg * (L + 1), dtype=torch.float64, device="cuda") for _ in parts]
    torch.cuda.synchronize()          # torch's fill runs on ITS stream: it must have landed before the engines' (non-blocking) streams write the tables
    for p_ in (0, 1):
        for e, t in zip(parts, tabs):
            e.pool_partials(p_, t.data_ptr(), 0, nseg)
            e.synchronize()
        if exact:
            assert not bool(((tabs[0] != 0) & (tabs[1] != 0)).any())      # disjoint support: the sum is exact in any order
        total = tabs[1] + tabs[0]
        torch.cuda.synchronize()
        for e in parts:
            e.pool_consume(p_, total.data_ptr(), nseg, 5.0 / T)
            e.synchronize()
    got = [e.minv[0].copy() for e in parts]
    for e in parts:
        e.close()
    assert np.array_equal(got[0], got[1])
    if exact:
        assert np.array_equal(got[0], ref)
    else:
        assert np.allclose(got[0], ref, rtol=1e-13, atol=0)
    assert np.all(np.isfinite(ref)) and np.all(ref > 0)
