"""N>1 path on CPU (gloo, world_size 2).  What can run without a GPU is the HOST side of the product's exchange:
shard_range (contiguous blocks, RNG keyed by the global chain id), the fixed-point record of the global-stepsize
exchange (idhmc_xchg_accumulate / idhmc_xchg_mean of libidhmc.so, include/idhmc.h) and the all-reduce wrapper the GPU
hook uses (distributed.allreduce_xchg).  The oracle's chains stand in for the device (tests may use it); the property
checked is that two ranks reproduce the one-rank run EXACTLY -- the record is integer-valued, so the association of the
all-reduce cannot change a bit (tests/test_gpu_multirank.py checks the same on the device)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _global_da_run(pkg, first, count, N, seed, reduce_fn):
    """Global-eps warm-up stage over oracle chains [first, first+count) with the PRODUCT's host-side exchange:
    returns (eps used per transition, last draws)."""
    from oracle import oracle as O
    D = 24
    m = O.OracleModel.iso(D)
    opt = O.default_options(max_depth=6)
    chains = [O.OracleChain(m, opt, seed=seed, chain_id=first + c) for c in range(count)]
    for ch in chains:
        ch.random_position()
        ch.rand_p(0)
    # initial stepsize: exp(pooled mean of log eps) -- IDHMC_XCHG_LOGEPS record, all-reduced
    L = O.lib()
    logs = [L.orc_log_export(ch.find_initial_stepsize()[1]) for ch in chains]
    rec = torch.from_numpy(pkg.xchg_accumulate(pkg.XCHG_LOGEPS, logs))
    reduce_fn(rec)
    eps0 = L.orc_exp_export(pkg.xchg_mean(pkg.XCHG_LOGEPS, rec.numpy()))
    s = O.DAState()
    L.orc_da_init(s, eps0)
    trace = []
    for n in range(N):
        eps = L.orc_da_current_eps(s)
        acc = [ch.sample_tree(eps, n + 1).acceptance_rate for ch in chains]
        rec = torch.from_numpy(pkg.xchg_accumulate(pkg.XCHG_ACCEPT, acc))
        reduce_fn(rec)
        L.orc_da_adapt(opt, s, pkg.xchg_mean(pkg.XCHG_ACCEPT, rec.numpy()))
        trace.append(eps)
    return np.array(trace), np.stack([ch.q[:D].copy() for ch in chains]), int(rec[2])


def _worker(rank, world, port, total, N, seed, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inplacedhmc_jl_amd as pkg
    first, count = pkg.distributed.shard_range(total, rank, world)
    trace, q, n = _global_da_run(pkg, first, count, N, seed, pkg.distributed.allreduce_xchg)
    assert n == total
    np.savez(os.path.join(out, "rank%d.npz" % rank), trace=trace, q=q, first=first)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 6), (2, 7), (4, 9)])
def test_ranks_reproduce_one_rank_exactly(tmp_path, world, total):
    import inplacedhmc_jl_amd as pkg
    N, seed = 12, 77
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, total, N, seed, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    for k in range(1, world):
        assert np.array_equal(r[0]["trace"], r[k]["trace"])         # every rank holds the same global eps
    trace, q, _ = _global_da_run(pkg, 0, total, N, seed, lambda b: b)
    assert np.array_equal(trace, r[0]["trace"])                     # == : no tolerance, the exchange is exact
    assert np.array_equal(q, np.concatenate([x["q"] for x in r]))   # sharding-invariant chains
    assert [int(x["first"]) for x in r] == [pkg.distributed.shard_range(total, k, world)[0] for k in range(world)]


def test_exchange_record_matches_the_integer_restatement():
    """idhmc_xchg_* (host side of libidhmc.so) against the oracle's restatement with Python integers, edge cases
    included; and the record of a set equals the sum of the records of any split of it (what makes ranks agree)."""
    import inplacedhmc_jl_amd as pkg
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    acc = np.concatenate([rng.random(1000), [0.0, 1.0, 0.5, np.nan, -0.25, 1.5, 2.0 ** -60, 1 - 2.0 ** -53]])
    logs = np.concatenate([rng.normal(0, 3, 1000), [0.0, -745.0, 709.0, np.nan, -np.inf, np.inf, 1e-300, -2.0 ** -41]])
    for kind, vals in ((pkg.XCHG_ACCEPT, acc), (pkg.XCHG_LOGEPS, logs)):
        rec = pkg.xchg_accumulate(kind, vals)
        ref = O.xchg_record(kind, vals)
        assert [int(rec[0]), int(rec[1]), int(rec[2])] == ref and rec[3] == 0.0
        assert pkg.xchg_mean(kind, rec) == O.xchg_mean(kind, ref)
        for cut in (1, 17, 500, len(vals) - 1):
            parts = pkg.xchg_accumulate(kind, vals[:cut]) + pkg.xchg_accumulate(kind, vals[cut:])
            assert np.array_equal(parts, rec)
        rr = vals.copy()
        rng.shuffle(rr)
        assert np.array_equal(pkg.xchg_accumulate(kind, rr), rec)    # order-invariant
    fin = acc[np.isfinite(acc)].clip(0, 1)
    assert abs(pkg.xchg_mean(pkg.XCHG_ACCEPT, pkg.xchg_accumulate(pkg.XCHG_ACCEPT, fin)) - fin.mean()) < 1e-15
    with pytest.raises(pkg.IdhmcError):
        pkg.xchg_accumulate(7, [0.5])


def test_allreduce_is_a_noop_without_a_group():
    import inplacedhmc_jl_amd as pkg
    t = torch.tensor([1.5, 2.0, 3.0, 0.0], dtype=torch.float64)
    assert pkg.distributed.allreduce_xchg(t) is t and t.tolist() == [1.5, 2.0, 3.0, 0.0]
    assert pkg.distributed.env_rank() == (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
                                          int(os.environ.get("LOCAL_RANK", 0)))
    assert pkg.distributed.shard_range(10, 0, 3) == (0, 4) and pkg.distributed.shard_range(10, 2, 3) == (7, 3)


class _FakeEngine:
    """stands in for Engine in attach_global_eps_native: records which communicator calls a rank made"""
    def __init__(self, rank, fail_id_on=None, fail_init_on=None):
        self.rank, self.fail_id_on, self.fail_init_on, self.calls = rank, fail_id_on, fail_init_on, []

    def comm_unique_id(self):
        self.calls.append("id")
        if self.rank == self.fail_id_on:
            raise RuntimeError("librccl.so not found (simulated)")
        return bytes(128)

    def comm_init(self, world, rank, uid):
        self.calls.append("init")
        assert len(uid) == 128
        if self.rank == self.fail_init_on:
            raise RuntimeError("ncclCommInitRank refused (simulated)")

    def comm_destroy(self):
        self.calls.append("destroy")


def _agree_worker(rank, world, port, case, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inplacedhmc_jl_amd as pkg
    eng = _FakeEngine(rank, fail_id_on=1 if case == "no_rccl_on_rank1" else None, fail_init_on=1 if case == "init_fails_on_rank1" else None)
    raised = ""
    try:
        pkg.distributed.attach_global_eps_native(eng)
    except RuntimeError as e:
        raised = str(e)
    with open(os.path.join(out, "rank%d.txt" % rank), "w") as f:
        f.write("%s|%s" % (",".join(eng.calls), raised))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ok", "no_rccl_on_rank1", "init_fails_on_rank1"])
def test_communicator_is_agreed_on_before_the_collective(tmp_path, case):
    """ncclCommInitRank is collective: a rank that cannot load RCCL must keep the OTHERS out of it, and a rank whose communicator
    was created while a peer's was refused must give it back -- every rank raises, none is left inside comm_init or with a
    communicator of its own (inplacedhmc.jl_amd/distributed.py::attach_global_eps_native)"""
    port = 29500 + (os.getpid() % 2000) + 11 + ["ok", "no_rccl_on_rank1", "init_fails_on_rank1"].index(case)
    mp.spawn(_agree_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    r = [open(tmp_path / ("rank%d.txt" % k)).read().split("|") for k in range(2)]
    if case == "ok":
        assert [x[0] for x in r] == ["id,init", "id,init"] and r[0][1] == "" and r[1][1] == ""
    elif case == "no_rccl_on_rank1":
        assert [x[0] for x in r] == ["id", "id"]                    # NOBODY entered the collective
        assert "not attempted" in r[0][1] and "this rank: ok" in r[0][1] and "simulated" in r[1][1]
    else:
        assert r[0][0] == "id,init,destroy" and r[1][0] == "id,init"     # the healthy rank gave its communicator back
        assert "destroyed again" in r[0][1] and "simulated" in r[1][1]
