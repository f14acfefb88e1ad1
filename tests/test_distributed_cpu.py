"""N>1 path on CPU (gloo, world_size 2): chains sharded in contiguous blocks with RNG keyed by the global
chain id, and the only exchange -- the 2-double {sum a, count} all-reduce of the global dual-averaging
stepsize -- through the same hook function the GPU path uses.  The oracle stands in for the device here
(tests may use it); the property checked is that 2 ranks reproduce the 1-rank run exactly."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _global_da_run(first, count, total, N, seed, reduce_fn):
    """Global-eps warmup stage with oracle chains [first, first+count): returns (eps trace, last draws)."""
    from oracle import oracle as O
    D = 24
    m = O.OracleModel.iso(D)
    opt = O.default_options(max_depth=6)
    chains = [O.OracleChain(m, opt, seed=seed, chain_id=first + c) for c in range(count)]
    for ch in chains:
        ch.random_position()
    eps0 = 0.5
    mu, mm, Hbar, le, lb = np.log(10.0) + np.log(eps0), 0, 0.0, np.log(eps0), 0.0
    trace = []
    for n in range(N):
        eps = float(np.exp(le))
        acc = [ch.sample_tree(eps, n + 1).acceptance_rate for ch in chains]
        buf = torch.tensor([float(np.sum(np.array(acc)[::-1][::-1])), float(count)], dtype=torch.float64)
        reduce_fn(buf)
        a = float(buf[0] / buf[1])
        assert int(buf[1]) == total
        mm += 1
        Hbar += (0.8 - a - Hbar) / (mm + 10)
        le = mu - np.sqrt(mm) / 0.05 * Hbar
        lb += mm ** -0.75 * (le - lb)
        trace.append(eps)
    return np.array(trace), np.stack([ch.q[:D].copy() for ch in chains])


def _worker(rank, world, port, total, N, seed, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inplacedhmc_jl_amd as pkg
    first, count = pkg.distributed.shard_range(total, rank, world)
    trace, q = _global_da_run(first, count, total, N, seed, pkg.distributed.allreduce_sum2)
    np.savez(os.path.join(out, "rank%d.npz" % rank), trace=trace, q=q, first=first)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_one_rank(tmp_path):
    total, N, seed = 6, 12, 77
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, total, N, seed, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["trace"], r1["trace"])                 # every rank holds the same global eps
    sys.path.insert(0, ROOT)
    trace, q = _global_da_run(0, total, total, N, seed, lambda b: b)
    # the pooled sum is associated differently (3+3 vs 6): identical up to summation rounding
    assert np.allclose(trace, r0["trace"], rtol=1e-13, atol=0)
    q2 = np.concatenate([r0["q"], r1["q"]])
    if np.array_equal(trace, r0["trace"]):
        assert np.array_equal(q, q2)                                # sharding-invariant chains
    assert int(r1["first"]) == 3


def test_allreduce_is_a_noop_without_a_group():
    sys.path.insert(0, ROOT)
    import inplacedhmc_jl_amd as pkg
    t = torch.tensor([1.5, 2.0], dtype=torch.float64)
    assert pkg.distributed.allreduce_sum2(t) is t and t.tolist() == [1.5, 2.0]
    assert pkg.distributed.env_rank() == (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
                                          int(os.environ.get("LOCAL_RANK", 0)))
