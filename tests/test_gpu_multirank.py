"""The N>1 path on a real device: two processes (ranks) share the box's one GPU, each owning half of the
chains, and run a global-eps warm-up stage through the library's own driver with the exchange wired to
torch.distributed (gloo here: RCCL refuses two ranks on one device; on a multi-GPU node the same call
sequence runs over "nccl" = RCCL, or over the library's communicator, idhmc_comm_init).
Checked: every rank ends every transition with the same global stepsize; the two-rank run reproduces the
one-rank run of all chains (chains are keyed by global id; the pooled sum is associated differently, so
eps agrees to summation rounding -- tolerance written below -- and, when it agrees exactly, so do the draws)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, TOTAL, N, SEED = 64, 24, 10, 11


def _problem():
    return np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)


def _stage(pkg, first, count, attach):
    mu, sig = _problem()
    opt = pkg.default_options(max_depth=6, eps_mode=pkg.EPS_GLOBAL)
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), count, opt, seed=SEED, first_chain=first)
    keep = attach(eng)
    eng.random_position()
    eng.set_eps(0.05)
    draws, stats = eng.tuning_stage(N, False, 0, store_draws=True)
    eps = eng.eps.copy()
    eng.close()
    del keep
    return eps, draws


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inplacedhmc_jl_amd as pkg
    first, count = pkg.distributed.shard_range(TOTAL, rank, world)
    eps, draws = _stage(pkg, first, count, lambda eng: pkg.distributed.attach_global_eps(eng))
    np.savez(os.path.join(out, "rank%d.npz" % rank), eps=eps, draws=draws, first=first)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_one_rank(idhmc, tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.all(r0["eps"] == r0["eps"][0]) and np.array_equal(r0["eps"][:1], r1["eps"][:1])   # one global eps
    eps, draws = _stage(idhmc, 0, TOTAL, lambda eng: None)
    # pooled acceptance sum: (12 chains) + (12 chains) vs 24 chains in one fixed-order reduction
    assert np.allclose(eps[0], r0["eps"][0], rtol=1e-12, atol=0)
    both = np.concatenate([r0["draws"], r1["draws"]], axis=1)        # [N][chains][D]
    assert both.shape == draws.shape
    if eps[0] == r0["eps"][0]:
        assert np.array_equal(both, draws)
    else:
        assert np.allclose(both[0], draws[0], rtol=0, atol=0)         # the first transition uses the common eps0
