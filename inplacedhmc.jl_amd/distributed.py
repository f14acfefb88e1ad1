"""Multi-GPU: one process per GPU, chains sharded in contiguous blocks, no data-path collective.
The only exchange is the 4-double fixed-point record {hi-limb sum, lo-limb sum, count, errors} of the global
dual-averaging stepsize during warm-up (integers carried in doubles: exact under any all-reduce order, so eps is
bit-identical for any number of ranks; include/idhmc.h): RCCL over xGMI, either the library's own communicator (attach_global_eps_native:
ncclAllReduce on the context's stream, no Python in the loop) or torch.distributed's "nccl" backend through
the hook (attach_global_eps); "gloo" in the CPU tests.
The reference has no inter-chain communication at all (src/mcmc.jl:150-157); the global-eps mode is this
engine's addition (BASELINE.json north_star)."""
import os


def shard_range(total_chains, rank, world):
    """Contiguous block of global chain ids for `rank`: (first, count).  RNG streams are keyed by the
    global id, so results do not depend on `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(int(total_chains), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def allreduce_xchg(tensor, group=None):
    """SUM-all-reduce the exchange record in place; a no-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def attach_global_eps(engine, group=None):
    """Wire the engine's global-eps exchange to torch.distributed: the library writes the exchange record into a torch
    CUDA tensor on a torch stream it shares with the engine, and the hook all-reduces the tensor on that stream.
    A dedicated torch stream is used, not torch's default stream: the default stream's handle is NULL, which
    idhmc_set_stream reads as "the library's own (non-blocking) stream", and the legacy default stream does not
    order itself against non-blocking streams -- the all-reduce would race the kernels around it.
    Returns (tensor, stream); keep both alive as long as the engine."""
    import torch
    from .engine import XCHG_DOUBLES
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        buf = torch.zeros(XCHG_DOUBLES, dtype=torch.float64, device="cuda")
    stream.synchronize()
    engine.set_stream(stream.cuda_stream)

    def hook(_ptr):
        with torch.cuda.stream(stream):
            allreduce_xchg(buf, group)
    engine.set_allreduce_hook(hook, buf.data_ptr())
    return buf, stream


def attach_global_eps_native(engine, rank=None, world=None, group=None):
    """Give the engine its own RCCL communicator (idhmc_comm_init): rank 0 creates the 128-byte id, the
    process group (any backend) carries it to the other ranks, and from then on the library enqueues the
    4-double all-reduce itself on the context's stream.  Without torch.distributed (single process) pass
    rank=0, world=1.
    ncclCommInitRank is collective, so the ranks agree BEFORE it that every one of them can take part (RCCL loadable -- probed with
    a throw-away id -- and the id received): if any cannot, none calls comm_init and RuntimeError is raised on EVERY rank, so that
    callers can fall back to the hook (attach_global_eps) together instead of leaving ranks waiting inside the collective.  If
    comm_init itself fails on some rank after that, the ranks agree again, the ones that did get a communicator destroy it
    (a context that kept one would take the communicator branch of the pooled metric alone), and the error is raised everywhere."""
    if world is None:
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

        def all_ok(flag):
            ok = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            return int(ok[0]) == 1
        local_err = None
        try:                                # can this rank load RCCL at all?  (every rank probes: the id call needs no peer)
            probe = engine.comm_unique_id()
        except Exception as e:              # noqa: BLE001
            probe, local_err = None, e
        box = [probe if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = box[0]
        if uid is None and local_err is None:
            local_err = RuntimeError("rank 0 produced no RCCL id")
        if not all_ok(local_err is None):
            raise RuntimeError("RCCL communicator not attempted: some rank cannot take part (this rank: %s)" % (local_err if local_err else "ok"))
        try:
            engine.comm_init(world, rank, uid)
        except Exception as e:              # noqa: BLE001
            local_err = e
        if not all_ok(local_err is None):
            if local_err is None:
                engine.comm_destroy()
            raise RuntimeError("RCCL communicator not created on every rank (this rank: %s)" % (local_err if local_err else "ok, destroyed again"))
    else:
        if world != 1:
            raise ValueError("without a process group only a single-rank communicator can be made")
        uid = engine.comm_unique_id()
        engine.comm_init(world, rank, uid)
