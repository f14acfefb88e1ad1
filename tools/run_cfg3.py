"""configs[2]: 1024-dim diagonal Gaussian, 65 536 chains, full NUTS tree doubling + dual-averaging warm-up
(default stages 75/25/50/100/200/400/50) then N sampling transitions, on one MI355X (GPU box).
Reports warm-up and sampling phases separately; draws are reduced on the fly (running moments).
configs[4] is the same script under torchrun (one rank per GPU, C chains per rank, chain ids rank*C ...):
  EPS_MODE=global COMM=native python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
      --master-addr 127.0.0.1 --master-port 29511 tools/run_cfg3.py
the only collective is the library's 4-double RCCL all-reduce per warm-up transition; rank 0 prints."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D = 1024
C = int(os.environ.get("C", 65536))
N = int(os.environ.get("N", 200))
MODE = os.environ.get("EPS_MODE", "per_chain")
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
METRIC = os.environ.get("METRIC_MODE", "per_chain")       # per_chain (reference semantics) | pooled
opt = pkg.default_options(eps_mode=pkg.EPS_GLOBAL if MODE == "global" else pkg.EPS_PER_CHAIN,
                          metric_mode=pkg.METRIC_POOLED if METRIC == "pooled" else pkg.METRIC_PER_CHAIN)
RANK, WORLD, LOCAL = pkg.distributed.env_rank()
dist = None
if WORLD > 1:
    import torch, torch.distributed as dist
    torch.cuda.set_device(LOCAL)
    dist.init_process_group("nccl", device_id=torch.device("cuda", LOCAL))
eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, opt, seed=20261004, first_chain=RANK * C, device=LOCAL)
if MODE == "global" and WORLD > 1:
    if os.environ.get("COMM", "native") == "native":
        pkg.distributed.attach_global_eps_native(eng)          # RCCL communicator owned by the library
    else:
        _keep = pkg.distributed.attach_global_eps(eng)         # torch.distributed ("nccl" = RCCL) through the hook
elif os.environ.get("COMM") == "native":    # single process: a one-rank communicator, to have RCCL in the loop
    pkg.distributed.attach_global_eps_native(eng, rank=0, world=1)
_print = print
def print(*a, **k):
    if RANK == 0:
        _print(*a, **k)
print("device GiB", eng.device_bytes() / 2**30, "eps_mode", MODE, "comm", os.environ.get("COMM", "none"), flush=True)
t0 = time.perf_counter()
eng.random_position(); eng.set_eps(1.0); eng.refresh_momentum(0); eng.find_initial_stepsize(); eng.synchronize()
t1 = time.perf_counter()
e0 = eng.eps
print(f"initial stepsize search {t1-t0:.2f}s eps median {np.median(e0):.4g} min {e0.min():.3g} max {e0.max():.3g}", flush=True)
it = 0
stages = [(75, 0)] + [(25 << d, 1) for d in range(5)] + [(50, 0)]
wsteps = 0
tw0 = time.perf_counter()
for n, adapt in stages:
    s0 = eng.total_steps(); ts = time.perf_counter()
    eng.tuning_stage(n, adapt, it, store_draws=False, store_stats=False); eng.synchronize()
    dt = time.perf_counter() - ts; ds = eng.total_steps() - s0
    it += n; wsteps += ds
    st = eng.tree_stats()
    print(f"stage N={n:3d} metric={adapt} {dt:6.2f}s {ds/dt:.3e} leapfrog-steps/s mean steps/transition {ds/n/C:.1f} "
          f"eps median {np.median(eng.eps):.4g} last acc {st['acceptance_rate'].mean():.3f} depth {st['depth'].mean():.2f}", flush=True)
tw = time.perf_counter() - tw0
eng.moments_reset()
eng.diag_reset()          # the reference's diagnostics, reduced on the device: no record ever leaves it
s0 = eng.total_steps(); ts = time.perf_counter()
eng.mcmc(N, it, store_draws=False, store_stats=False); eng.synchronize()
dt = time.perf_counter() - ts; ds = eng.total_steps() - s0
mean, var, cnt = eng.moments()
st = eng.tree_stats()
pm = mean.mean(axis=0); pv = (var.mean(axis=0) * (N - 1) / N + mean.var(axis=0))
res = {"chains": C, "warmup_s": tw, "warmup_steps_per_s": wsteps / tw, "sampling_s": dt, "sampling_steps_per_s": ds / dt,
       "sampling_transitions_per_s": N * C / dt, "eps_median": float(np.median(eng.eps)),
       "acceptance_last": float(st["acceptance_rate"].mean()), "depth_last": float(st["depth"].mean()),
       "max_abs_mean_err_over_sigma": float(np.abs((pm - mu) / sig).max()),
       "var_ratio_min": float((pv / sig**2).min()), "var_ratio_max": float((pv / sig**2).max()),
       "minv_over_sigma2_median": float(np.median(eng.minv[:64] / sig**2)), "eps_mode": MODE, "metric_mode": METRIC,
       "rhat_max": float(pkg.rhat_from_moments(mean, var, N).max()), "draws_total": int(N) * C}
ess_tot = pkg.ess_from_moments(mean, var, N)          # replicated batch means, uncapped
res.update({"ess_total_min_over_dims": float(ess_tot.min()), "ess_total_median_over_dims": float(np.median(ess_tot)),
            "ess_per_draw_min": float(ess_tot.min() / (N * C)), "ess_per_draw_median": float(np.median(ess_tot) / (N * C))})
counters = eng.diag_counters()
ebfmi = eng.ebfmi()
res.update({"ebfmi_min": float(ebfmi.min()), "ebfmi_median": float(np.median(ebfmi))})
if dist is not None:      # integer counters: the sum over ranks is the counters of the whole run
    import torch
    tc = torch.from_numpy(counters.astype(np.int64)).cuda()
    dist.all_reduce(tc, op=dist.ReduceOp.SUM)
    counters = tc.cpu().numpy().astype(np.uint64)
summary = pkg.summary_from_counters(counters)
print(str(summary))        # the reference's show(::TreeStatisticsSummary), src/diagnostics.jl:103-127
res.update({"summary_N": summary.N, "acceptance_mean": summary.a_mean, "acceptance_quantiles_5_25_50_75_95": summary.a_quantiles.tolist(),
            "termination": summary.termination_counts, "depth_counts": summary.depth_counts.tolist()})
if dist is not None:
    import torch
    t = torch.tensor([float(wsteps), float(ds), tw, dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t[:2], op=dist.ReduceOp.SUM)
    dist.all_reduce(t[2:], op=dist.ReduceOp.MAX)
    res.update({"ranks": WORLD, "chains": C * WORLD, "warmup_steps_per_s": float(t[0] / t[2]),
                "sampling_steps_per_s": float(t[1] / t[3]), "warmup_s": float(t[2]), "sampling_s": float(t[3])})
print(json.dumps(res))
eng.close()
if dist is not None:
    dist.destroy_process_group()
