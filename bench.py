#!/usr/bin/env python3
"""bench.py -- leapfrog-steps/sec (all chains), 1024-dim Gaussian (BASELINE.json).

Workload (BASELINE.json configs[1]): 1024-dim diagonal Gaussian, mu_d = sin(d),
sigma_d log-spaced in [0.1, 10], M^-1 = sigma^2, 65 536 chains per GPU, fixed eps = 0.1.
One "step" = one fused single-step leapfrog sweep (idhmc_leapfrog(eps, 1)) over every chain of the
rank: the state (q, p, grad l) is read from and written back to HBM once per step.  N ranks hold
N x 65 536 independent chains (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM-bound fused
leapfrog kernel; achieved = 6*D*8 bytes x chains / mean launch time measured with HIP events on the
library's stream over the timed region) and `cpu_baseline` (the CPU oracle = a C port of the
reference's in-place path, timed on this host's cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D = 1024
CHAINS_PER_GPU = 65536
EPS = 0.1
BYTES_PER_STEP = 6 * D * 8          # read q,p,grad + write q',p',grad' 
HBM_PEAK_GBS = 8000.0               # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def workload():
    sig = np.logspace(-1, 1, D)
    mu = np.sin(np.arange(D, dtype=np.float64))
    return mu, sig


def host_isa():
    """what the CPU baseline ran on: model name and the vector ISA levels /proc/cpuinfo lists"""
    model, flags = "unknown", set()
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("flags") and not flags:
                flags = set(ln.split(":", 1)[1].split())
    except OSError:
        pass
    return model, [f for f in ("avx2", "fma", "avx512f", "avx512vl", "avx512dq") if f in flags]


def cpu_baseline(mu, sig, target_seconds):
    """Oracle (kind "port"): same density, D, eps, M^-1; chains scaled so it runs ~target_seconds.  Compiled HERE, on
    the machine it is timed on, with -O3 -march=native; if that build fails the shipped
    -mavx2 -mfma build is timed and the line says so."""
    import tempfile
    flags = "-O3 -mavx2 -mfma -ffp-contract=off (shipped build; the native build failed)"
    try:
        sys.path.insert(0, ROOT)
        import importlib.util
        spec = importlib.util.spec_from_file_location("_orc_native_builder", os.path.join(ROOT, "oracle", "oracle.py"))
        builder = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(builder)
        so, flags = builder.build_native(tempfile.mkdtemp(prefix="idhmc_orc_"))
        os.environ["IDHMC_ORACLE_LIB"] = so
    except Exception as e:                      # noqa: BLE001 -- any failure falls back to the shipped build, stated in the line
        sys.stderr.write("native oracle build failed (%s); timing the shipped build\n" % e)
    from oracle import oracle as O
    om = O.OracleModel.diag(mu, 1.0 / sig ** 2)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the GPU box gives one GPU's job a 16-core share of the host; IDHMC_CPU_THREADS overrides
    cores = int(os.environ.get("IDHMC_CPU_THREADS", min(avail, 16)))
    nch = max(cores * 4, 64)
    t = O.bench_leapfrog(om, nch, 200, EPS, minv=sig ** 2, nthreads=cores)     # calibration (warm)
    rate = nch * 200 / t
    sweeps = max(200, int(rate * target_seconds / nch))
    t = O.bench_leapfrog(om, nch, sweeps, EPS, minv=sig ** 2, nthreads=cores)
    # the same port on ONE thread (total and per-core figures), ~1/8 of the budget
    n1 = 16
    s1 = max(50, int(rate / cores * target_seconds / 8 / n1))
    t1 = O.bench_leapfrog(om, n1, s1, EPS, minv=sig ** 2, nthreads=1)
    model, isa = host_isa()
    return {"value": nch * sweeps / t, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
            "sample": "%d chains x %d fixed-eps leapfrog sweeps of the same 1024-dim diagonal Gaussian, "
                      "one chain per host thread (%.1f s)" % (nch, sweeps, t),
            "per_core": nch * sweeps / t / cores, "value_1_thread": n1 * s1 / t1,
            "thread_scaling": nch * sweeps / t / (n1 * s1 / t1),
            "note": "a reported baseline, not a target: %d of the host's %d hardware threads (the share of a one-GPU job; IDHMC_CPU_THREADS "
                    "overrides); per-thread rate at %d threads is below the one-thread rate because the threads share SMT siblings and the all-core clock"
                    % (cores, avail, cores),
            "isa": {"compiler_flags": "gcc " + flags, "host_cpu": model, "host_vector_isa": isa, "host_cores_visible": avail}}


def cpu_nuts_baseline(mu, sig, cores, target_seconds):
    """the same NUTS transitions (eps = 0.25, M^-1 = sigma^2, in-distribution start) on the host: the oracle's sample_tree
    (the in-place path of src/NUTS.jl:251-264), one chain per host thread at a time, timed inside the C library"""
    from oracle import oracle as O
    om = O.OracleModel.diag(mu, 1.0 / sig ** 2)
    nch = cores * 4
    q0 = mu + sig * np.random.default_rng(5).standard_normal((nch, D))
    t, steps = O.bench_nuts(om, nch, 5, 0.25, minv=sig ** 2, q0=q0, nthreads=cores)          # calibration
    trans = max(5, int(5 * target_seconds / max(t, 1e-3)))
    t, steps = O.bench_nuts(om, nch, trans, 0.25, minv=sig ** 2, q0=q0, nthreads=cores)
    return {"leapfrog_steps_per_s": steps / t, "transitions_per_s": nch * trans / t, "cores": cores, "kind": "port",
            "sample": "%d chains x %d transitions at eps = 0.25 (%.1f leapfrogs each), one chain per host thread (%.1f s)"
                      % (nch, trans, steps / (nch * trans), t)}


def power_and_clock(run_for, seconds=1.5):
    """socket power and shader clock (rocm-smi) while `run_for()` is called in a loop for `seconds`; None when rocm-smi is
    not there or refuses.  Outside every timed region."""
    import subprocess
    import threading
    if any(k.startswith("ROCPROF") or "rocprofiler" in v for k, v in os.environ.items() if k.startswith("ROCP") or k == "LD_PRELOAD"):
        return None     # under rocprofv3 the tool library initialises the GPU in every child: rocm-smi's `env python3` hop would be an exec after that
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20).stdout
                f = out.strip().splitlines()[-1].split(",")
                samples.append((float(f[5].strip("()").lower().replace("mhz", "")), float(f[-1])))
            except Exception:
                return
            time.sleep(0.2)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        run_for()
    stop[0] = True
    th.join()
    busy = samples[1:] if len(samples) > 2 else samples     # the first sample may predate the load
    if not busy:
        return None
    return {"socket_power_W": float(np.median([p_ for _, p_ in busy])), "sclk_MHz": float(np.median([c_ for c_, _ in busy])),
            "samples": len(busy), "board_limit_W": 1400.0, "sclk_nominal_MHz": 2400.0, "how": "rocm-smi --showclocks --showpower, in this run"}


def committed_profile(name):
    """a profile summary committed under profiles/ (numbers NOT measured in this run; the line labels them with the file)"""
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path)), "profiles/" + name, time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path)))
    except Exception:   # noqa: BLE001
        return None, None, None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--global-eps-transitions", type=int, default=30)
    ap.add_argument("--no-cfg3", action="store_true", help="skip the configs[2] end-to-end leg (cfg3_full)")
    ap.add_argument("--cfg3-chains", type=int, default=None, help="chains of the cfg3_full leg (default: --chains)")
    ap.add_argument("--cfg3-scale", type=float, default=1.0, help="scale of the cfg3_full leg's stage lengths and draws (tests: < 1)")
    return ap.parse_args()


def visible_gpus():
    """Number of GPUs this job sees, asked of a CHILD process: the launcher itself never initialises the GPU (a process that
    has must not start others on this pool) and never imports torch."""
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                         capture_output=True, text=True, timeout=600)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit("bench.py: could not count the GPUs (torch said: %s)" % out.stderr.strip()[-500:])


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here -- fresh children, one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment exactly as torch.distributed.run would set them -- relay
    rank 0's single JSON line, and fail loudly (non-zero, the rest of the ranks ended by PID) if any rank fails.  This is the
    many-chain form of the reference's one-worker-per-chain-block loop (src/mcmc.jl:150-157)."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    backend = os.environ.get("IDHMC_DIST_BACKEND", "nccl")
    ndev = visible_gpus()
    if ndev < 1:
        raise SystemExit("bench.py --gpus %d: no GPU visible (the product has no CPU path)" % n)
    if backend == "nccl" and ndev < n:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; RCCL needs one device per rank "
                         "(IDHMC_DIST_BACKEND=gloo rehearses the N-rank path with ranks sharing devices)" % (n, ndev))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IDHMC_BENCH_LAUNCH="self-spawned")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()))
    reader.start()
    failed = None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0:
                    failed = (r, rc)
        time.sleep(0.05)
    if failed is not None:
        for r in live:                  # the others may be waiting for the dead rank in a collective
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    reader.join()
    line = (got[0] or b"").decode()
    if failed is not None:
        sys.stderr.write("bench.py --gpus %d: rank %d exited with code %d; the other ranks were stopped\n" % (n, failed[0], failed[1]))
        raise SystemExit(1)
    if not line.strip().startswith("{"):
        raise SystemExit("bench.py --gpus %d: rank 0 printed no result line" % n)
    sys.stdout.write(line if line.endswith("\n") else line + "\n")
    sys.stdout.flush()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args)       # before torch is imported or the GPU touched in this process
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    run_rank(args)


def run_rank(args):
    # Exactly ONE line may reach stdout.  Libraries below us write there too (RCCL prints a five-line version banner from
    # ncclCommInitRank on rank 0), so everything but the final JSON line goes to stderr: fd 1 is pointed at fd 2 for the
    # whole run and the saved descriptor is used for the one line at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launch = os.environ.get("IDHMC_BENCH_LAUNCH", "external launcher (torch.distributed.run)" if world > 1 else "single process")
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    ndev = torch.cuda.device_count()
    # IDHMC_DIST_BACKEND=gloo rehearses the N-rank path on fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("IDHMC_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, ndev))
    local = local % ndev if backend != "nccl" else local
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)

    if os.environ.get("IDHMC_BENCH_FAIL_RANK") == str(rank):      # test hook: this rank dies after the rendezvous
        raise SystemExit(7)
    import inplacedhmc_jl_amd as pkg
    mu, sig = workload()
    C = args.chains
    opt = pkg.default_options(metric_mode=pkg.METRIC_SHARED)
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, opt, seed=1, first_chain=rank * C, device=local)
    eng.set_minv(sig ** 2)
    # q0 ~ mu + sigma N(0,1) (in-distribution start), p0 ~ W N(0,1) from the engine's Philox stream
    rng = np.random.default_rng(1 + rank)
    blk = 4096
    q0 = np.empty((C, D))
    for i in range(0, C, blk):
        q0[i:i + blk] = mu + sig * rng.standard_normal((min(blk, C - i), D))
    eng.set_q(q0)
    del q0
    eng.refresh_momentum(1)

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    pg, pn = eng.placement_info()          # where the state arrays went: probe rate of the placement kept
    placement = {"probe_GBps": pg, "candidates_tried": pn, **eng.placement_cost(),
                 "note": "the context tries placements of q, p, grad in HBM at creation and keeps the fastest (idhmc_placement_info)"}
    for _ in range(args.warmup):
        eng.leapfrog(EPS, 1)
    barrier()
    t0 = time.perf_counter()
    ms_kernel = eng.time_leapfrog(EPS, args.steps)      # K launches between two HIP events, then sync
    eng.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms_ranks = {"min": ms_kernel, "max": ms_kernel}
    if dist is not None:
        tt = torch.tensor([elapsed, ms_kernel, -ms_kernel], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, ms_kernel = float(tt[0]), float(tt[1])
        kernel_ms_ranks = {"min": -float(tt[2]), "max": float(tt[1])}
        dist.barrier()
    finite = bool(np.isfinite(eng.lq).all())

    # SURVEY.md 8d's second cfg2 variant, outside the timed region (rank 0 at N=1): M^-1 = I, eps = 0.1 sigma_min
    identity = None
    if world == 1:
        eng.set_minv(np.ones(D))
        eng.refresh_momentum(2)
        eng.time_leapfrog(0.01, 20)
        ms_i = eng.time_leapfrog(0.01, 200)
        identity = {"kernel_ms": ms_i, "achieved_GBps": BYTES_PER_STEP * C / (ms_i * 1e-3) / 1e9,
                    "leapfrog_steps_per_s": C / (ms_i * 1e-3), "eps": 0.01, "note": "same kernel, M^-1 = I"}
        eng.set_minv(sig ** 2)
        eng.refresh_momentum(3)

    # the same sweep with the gradient stream dropped (IDHMC_GRAD_RECOMPUTE: grad l is re-derived from q, 2 flops per
    # element, and not written back): 4 D 8 bytes of actual traffic per chain-step instead of the algorithmic 6 D 8
    regrad = None
    if world == 1:
        eng.set_leapfrog_grad_mode(pkg.GRAD_RECOMPUTE)
        eng.time_leapfrog(EPS, 20)
        ms_r = eng.time_leapfrog(EPS, 200)
        regrad = {"kernel_ms": ms_r, "leapfrog_steps_per_s": C / (ms_r * 1e-3),
                  "actual_GBps": 4 * D * 8 * C / (ms_r * 1e-3) / 1e9,
                  "algorithmic_GBps": BYTES_PER_STEP * C / (ms_r * 1e-3) / 1e9,
                  "note": "optional mode, bit-identical results; not the headline (the headline moves the reference's 6 streams)"}
        eng.set_leapfrog_grad_mode(pkg.GRAD_STORE)

    def note(msg):
        sys.stderr.write("bench.py[rank %d] %s\n" % (rank, msg))
        sys.stderr.flush()

    def guarded(leg):
        """a secondary leg never costs the headline line: at N = 1 its failure is recorded in its field; at N > 1 the rank
        fails (the launcher then stops the other ranks), because the other ranks would wait in the leg's collectives"""
        note("leg " + leg.__name__)
        if world > 1:
            return leg()
        try:
            return leg()
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc()
            return {"error": "%s: %s" % (type(e).__name__, e)}

    # secondary figure, outside the timed region (rank 0 at N=1): full NUTS transitions of the same density
    # (configs[2]'s kernel) at eps = 0.25 -- the phase point stays in registers inside a tree, so this path is
    # not HBM-bound and its leapfrog rate exceeds the streamed kernel's roofline
    def leg_nuts():
        eng.set_eps(0.25)
        for it in range(1, 4):
            eng.nuts_transition(it)
        eng.synchronize()
        s0 = eng.total_steps()
        ms_n = eng.time_transitions(10, 3)
        steps = eng.total_steps() - s0
        st = eng.tree_stats()
        nuts = {"leapfrog_steps_per_s": steps / (ms_n * 1e-3), "transitions_per_s": 10 * C / (ms_n * 1e-3),
                "mean_tree_depth": float(st["depth"].mean()), "eps": 0.25,
                "note": "one NUTS transition per chain per launch (k_nuts), 10 launches, HIP-event timed"}
        # the same at eps = 0.03 (trees of depth 7: 127 leapfrogs per transition), where the per-transition fixed work
        # (momentum refresh, regeneration, epilogue) no longer shows
        eng.set_eps(0.03)
        for it in range(20, 22):
            eng.nuts_transition(it)
        eng.synchronize()
        s7 = eng.total_steps()
        ms_7 = eng.time_transitions(5, 22)
        steps7 = eng.total_steps() - s7
        nuts["deep_trees"] = {"leapfrog_steps_per_s": steps7 / (ms_7 * 1e-3), "transitions_per_s": 5 * C / (ms_7 * 1e-3),
                              "mean_tree_depth": float(eng.tree_stats()["depth"].mean()), "eps": 0.03}
        # the same transitions as the library's drivers launch them where nothing leaves the device per transition (every warm-up
        # stage; sampling into device-side moments): n per launch, the device hands out (transition, chain) pairs (idhmc_nuts_transitions)
        def _fused(eps_f, n, it0):
            eng.set_eps(eps_f)
            eng.nuts_transitions(it0, 2)
            eng.time_transitions_fused(n, it0 + 2)          # (the first launches of a kind run slower: allocation-time work of the runtime)
            sf = eng.total_steps()
            ms_f = eng.time_transitions_fused(n, it0 + 2 + n)
            stf = eng.total_steps() - sf
            return {"transitions_per_launch": n, "leapfrog_steps_per_s": stf / (ms_f * 1e-3), "transitions_per_s": n * C / (ms_f * 1e-3),
                    "mean_tree_depth": float(eng.tree_stats()["depth"].mean()), "eps": eps_f}
        nuts["several_transitions_per_launch"] = _fused(0.25, 40, 1000)
        nuts["several_transitions_per_launch"]["deep_trees"] = _fused(0.03, 10, 2000)
        nuts["several_transitions_per_launch"]["placement_probe_ok"], nuts["several_transitions_per_launch"]["used_by_drivers"] = eng.fused_launch_info()
        nuts["several_transitions_per_launch"]["note"] = ("bit-identical to single launches (tests/test_gpu_fused.py); what it saves is the end of every "
                                                         "launch -- wavefronts finishing their last tree while the queue is empty -- and the gap to the next")
        # What bounds k_nuts: the phase point stays in registers inside a tree, so its algorithmic 6 D 8
        # bytes never move; the kernel is bound jointly by fp64 VALU issue and by the tree arena's traffic.
        # achieved = algorithmic flops (17 per element and leaf + 6 per element and merge, one merge per leaf) / time,
        # against the fp64 vector peak; the counter evidence is the committed PMC summary (not this run).
        flops = steps * D * (17 + 6)
        eng.set_eps(0.25)
        it_pw = [40]

        def _nuts_once():
            eng.time_transitions(5, it_pw[0])
            it_pw[0] += 5
        nuts["power"] = power_and_clock(_nuts_once)      # k_nuts runs at the board's power limit: the clock says so
        nuts["roofline"] = {"bound": "jointly fp64 VALU issue (79-81 % busy per SIMD) and the tree arena's traffic (4.5-5.6 TB/s at the fabric), by the "
                                     "counters of profiles/r03_nuts_pmc.json; neither is saturated; the phase point stays in registers inside a tree, so "
                                     "the state's HBM streams (the headline's roofline) do not bound this kernel",
                            "achieved": flops / (ms_n * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                            "frac": flops / (ms_n * 1e-3) / 1e12 / 78.6,
                            "algorithmic_flops_per_leapfrog": D * 23}
        prof, fname, fdate = committed_profile("r03_nuts_pmc.json")
        if prof is not None and C == CHAINS_PER_GPU:
            d4 = prof["configs"].get("d4", {}).get("derived", {})
            d7 = prof["configs"].get("d7", {}).get("derived", {})
            nuts["roofline"]["counters_deep_trees"] = {
                "valu_active_share_per_wave": d7.get("wave_cycle_shares", {}).get("SQ_ACTIVE_INST_VALU"), "waves_per_simd": 2,
                "arena_hbm_GBps": d7.get("hbm_GBps"), "arena_bytes_per_leapfrog": d7.get("hbm_bytes_per_leapfrog"),
                "valu_insts_per_leapfrog": d7.get("valu_insts_per_leapfrog")}
            nuts["roofline"]["counters"] = {
                "file": fname, "file_date": fdate, "measured_in_this_run": False,
                "valu_active_share_per_wave": d4.get("wave_cycle_shares", {}).get("SQ_ACTIVE_INST_VALU"),
                "waves_per_simd": 2, "arena_hbm_GBps": d4.get("hbm_GBps"), "arena_bytes_per_leapfrog": d4.get("hbm_bytes_per_leapfrog"),
                "valu_insts_per_leapfrog": d4.get("valu_insts_per_leapfrog")}
        return nuts

    nuts = guarded(leg_nuts) if world == 1 else None

    # configs[2] end to end, outside the timed region (rank 0 at N=1): the same density, random start, the reference's default
    # warm-up schedule (InitialStepsizeSearch, 75 / 25-50-100-200-400 with metric windows / 50 transitions, per-chain eps and
    # per-chain diagonal metric: src/warmup.jl:361-372, 269-314) and then 200 draws; draws are reduced on the device (running
    # moments, diagnostics counters) -- one draw of all chains is 512 MiB.  Beside it the SAME chains (same seed, global chain ids
    # 0 .. n-1) run on the CPU oracle and on a small context whose draws are stored: results do not depend on how many chains run
    # beside a chain, so the three must agree bit for bit -- `chains_bit_identical` -- and ESS per transition GPU / CPU follows.
    def leg_cfg3_full():
        sc = args.cfg3_scale
        Cc = args.cfg3_chains or C
        sched = {"init_steps": max(2, int(round(75 * sc))), "middle_steps": max(2, int(round(25 * sc))), "doubling_stages": 5,
                 "terminating_steps": max(2, int(round(50 * sc)))}
        Nd = max(4, int(round(200 * sc)))
        seed3 = 20261004
        opt3 = pkg.default_options(**sched)
        e3 = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), Cc, opt3, seed=seed3, device=local)
        t0_ = time.perf_counter()
        e3.random_position()
        e3.set_eps(opt3.eps_init)
        e3.refresh_momentum(0)
        e3.find_initial_stepsize()
        e3.synchronize()
        t_search = time.perf_counter() - t0_
        stages = [(sched["init_steps"], 0)] + [(sched["middle_steps"] << d_, 1) for d_ in range(5)] + [(sched["terminating_steps"], 0)]
        it_, per_stage = 0, []
        w0_, ws0 = time.perf_counter(), e3.total_steps()
        for n_, adapt_ in stages:
            s0_, ts_ = e3.total_steps(), time.perf_counter()
            e3.tuning_stage(n_, adapt_, it_, store_draws=False, store_stats=False)
            e3.synchronize()
            per_stage.append({"transitions": n_, "metric_window": bool(adapt_), "seconds": time.perf_counter() - ts_,
                              "leapfrog_steps": e3.total_steps() - s0_})
            it_ += n_
        t_warm, warm_steps = time.perf_counter() - w0_, e3.total_steps() - ws0
        e3.moments_reset()
        e3.diag_reset()
        s0_, ts_ = e3.total_steps(), time.perf_counter()
        e3.mcmc(Nd, it_, store_draws=False, store_stats=False)
        e3.synchronize()
        t_samp, samp_steps = time.perf_counter() - ts_, e3.total_steps() - s0_
        mean_, var_, _cnt = e3.moments()
        pm_ = mean_.mean(axis=0)
        pv_ = var_.mean(axis=0) * (Nd - 1) / Nd + mean_.var(axis=0)
        ess_tot = pkg.ess_from_moments(mean_, var_, Nd)            # replicated batch means over chains, uncapped
        summ = pkg.summary_from_counters(e3.diag_counters())
        ncmp = min(Cc, 64)
        big_q, big_eps, big_minv = e3.q[:ncmp].copy(), e3.eps[:ncmp].copy(), e3.minv[:ncmp].copy()
        big_stats = e3.tree_stats()[:ncmp].copy()
        res = {"workload": "configs[2]: %d-dim diagonal Gaussian, %d chains, default warm-up stages %s + %d draws, per-chain eps and metric"
                           % (D, Cc, "/".join(str(n_) for n_, _ in stages), Nd),
               "schedule_scale": sc, "stepsize_search_s": t_search,
               "warmup": {"seconds": t_warm, "transitions": it_, "leapfrog_steps_per_s": warm_steps / t_warm, "per_stage": per_stage},
               "sampling": {"seconds": t_samp, "draws": Nd, "leapfrog_steps_per_s": samp_steps / t_samp,
                            "transitions_per_s": Nd * Cc / t_samp, "mean_leapfrogs_per_transition": samp_steps / (Nd * Cc)},
               "acceptance_mean": summ.a_mean, "acceptance_target": float(opt3.da_delta),
               "termination": summ.termination_counts, "eps_median": float(np.median(e3.eps)),
               "rhat_max": float(pkg.rhat_from_moments(mean_, var_, Nd).max()),
               "ess_per_draw_min": float(ess_tot.min() / (Nd * Cc)), "ess_per_draw_median": float(np.median(ess_tot) / (Nd * Cc)),
               "max_abs_mean_err_in_se": float((np.abs(pm_ - mu) / (sig / np.sqrt(ess_tot))).max()),
               "var_ratio_range": [float((pv_ / sig ** 2).min()), float((pv_ / sig ** 2).max())],
               "minv_over_sigma2_median": float(np.median(e3.minv[:256] / sig ** 2)),
               "ebfmi_median": float(np.median(e3.ebfmi()))}
        e3.close()
        # the same chains, stored: a small context on the device ...
        small = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), ncmp, opt3, seed=seed3, device=local)
        gdraws, gstats = small.mcmc_with_warmup(Nd)                # [Nd][ncmp][D]
        same_small = bool(np.array_equal(small.q, big_q) and np.array_equal(small.eps, big_eps) and np.array_equal(small.minv, big_minv)
                          and np.array_equal(gstats[-1], big_stats))
        small.close()
        res["chains_compared"] = ncmp
        res["large_run_equals_small_run_bitwise"] = same_small
        # ... and on the CPU oracle (checker; never the thing measured)
        if not args.no_cpu:
            from oracle import oracle as O
            om = O.OracleModel.diag(mu, 1.0 / sig ** 2)
            oo = O.default_options(**sched)
            tc_ = time.perf_counter()
            rc_, ochains, ostats, oeps = O.threaded_mcmc(om, Nd, ncmp, oo, seed=seed3, first_chain=0,
                                                          nthreads=int(os.environ.get("IDHMC_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16))))
            t_cpu = time.perf_counter() - tc_
            odraws = np.transpose(ochains[:, :Nd, :D], (1, 0, 2))  # -> [Nd][ncmp][D]
            ident = bool(rc_ == 0 and np.array_equal(odraws, gdraws) and np.array_equal(oeps, big_eps)
                         and np.array_equal(ostats["steps"][:, :Nd].T, gstats["steps"]) and np.array_equal(ostats["pi"][:, :Nd].T, gstats["pi"]))
            ess_g = np.mean([pkg.ess(gdraws[:, c_, :]).mean() for c_ in range(ncmp)]) / Nd
            ess_c = np.mean([pkg.ess(odraws[:, c_, :]).mean() for c_ in range(ncmp)]) / Nd
            res["cpu_oracle"] = {"chains": ncmp, "seconds": t_cpu, "kind": "port",
                                 "note": "warm-up + draws of the same chains (same seed, chain ids 0..%d) on the host oracle" % (ncmp - 1)}
            res["chains_bit_identical"] = ident
            res["ess_per_transition"] = {"gpu": float(ess_g), "cpu": float(ess_c), "gpu_over_cpu": float(ess_g / ess_c),
                                         "estimator": "Geyer initial positive sequence per chain and coordinate, mean over both"}
        return res

    cfg3 = guarded(leg_cfg3_full) if (world == 1 and not args.no_cfg3) else None

    # configs[3], outside the timed region (rank 0 at N=1): 256-dim dense multivariate normal, 16 384 chains, the
    # Sigma^-1 (q - mu) gradient on the fp64 matrix cores.  Both ceilings are reported: 2 D^2 flops per chain-step against
    # the fp64 MFMA peak, 6 D 8 bytes of state per chain-step against HBM (the config sits near the ridge).
    def leg_dense():
        Dd, Cd = 256, 16384
        rngd = np.random.default_rng(7)
        Qd, _ = np.linalg.qr(rngd.standard_normal((Dd, Dd)))
        lam = np.logspace(-2, 0, Dd)
        Pd = (Qd / lam) @ Qd.T
        Pd = 0.5 * (Pd + Pd.T)
        mud = np.cos(np.arange(Dd, dtype=np.float64))
        deng = pkg.Engine(pkg.DenseMVN(mud, Pd), Cd, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1, device=local)
        Sg = (Qd * lam) @ Qd.T
        deng.set_q(mud + rngd.standard_normal((Cd, Dd)) @ np.linalg.cholesky(0.5 * (Sg + Sg.T)).T)
        deng.refresh_momentum(1)
        deng.time_leapfrog(0.02, 20)
        ms_d = min(deng.time_leapfrog(0.02, 500) for _ in range(3))     # back-to-back sweeps, as configs[3] defines the measurement
        NSd = 64
        deng.leapfrog(0.02, NSd)
        deng.synchronize()
        best = 1e9
        for _ in range(3):
            td = time.perf_counter()
            deng.leapfrog(0.02, NSd)
            deng.synchronize()
            best = min(best, time.perf_counter() - td)
        dense_power = power_and_clock(lambda: deng.time_leapfrog(0.02, 1000))
        deng.refresh_momentum(2)
        deng.set_eps(0.05)
        for it in (1, 2):
            deng.nuts_transition(it)
        sd0 = deng.total_steps()
        ms_dn = deng.time_transitions(5, 2)
        sdn = deng.total_steps() - sd0
        # 20 transitions per launch (idhmc_nuts_transitions: how the drivers run a warm-up stage): at 4 chains per resident wavefront
        # the end of a single-transition launch is a quarter of its time
        deng.nuts_transitions(100, 2)
        deng.time_transitions_fused(20, 102)
        sdf = deng.total_steps()
        ms_df = deng.time_transitions_fused(20, 122)
        rtf = (deng.total_steps() - sdf) / (ms_df * 1e-3)
        it_dn = [200]

        def _dense_nuts_once():
            deng.time_transitions_fused(20, it_dn[0])
            it_dn[0] += 20
        dense_nuts_power = power_and_clock(_dense_nuts_once)
        flop = 2.0 * Dd * Dd
        dpmc = None                      # HBM bytes per sweep from the committed counter passes (not measured in this run)
        try:
            dj = json.load(open(os.path.join(ROOT, "profiles", "r02_dense_pmc.json")))["configs"]["lanes4"]["derived"]
            dpmc = {"file": "profiles/r02_dense_pmc.json", "measured_in_this_run": False,
                    "hbm_bytes_per_sweep": dj["hbm_read_bytes_per_sweep"] + dj["hbm_write_bytes_per_sweep"],
                    "over_algorithmic": dj["hbm_over_algorithmic"], "l2_hit_rate": dj["l2_hit_rate"]}
        except Exception:
            pass
        r1, rn, rt = Cd / (ms_d * 1e-3), Cd * NSd / best, sdn / (ms_dn * 1e-3)
        dense = {"workload": "configs[3]: %d-dim dense multivariate normal, %d chains, fp64 MFMA gradient" % (Dd, Cd),
                 "mfma_peak_TFLOPs": 78.6, "hbm_peak_GBps": HBM_PEAK_GBS,
                 "single_step_sweeps": {"chain_steps_per_s": r1, "kernel_ms": ms_d, "mfma_TFLOPs": r1 * flop / 1e12,
                                        "mfma_frac": r1 * flop / 1e12 / 78.6, "state_GBps": r1 * 6 * Dd * 8 / 1e9,
                                        "hbm_frac": r1 * 6 * Dd * 8 / 1e9 / HBM_PEAK_GBS, "sweeps_timed": 500, "traffic": dpmc, "power": dense_power,
                                        "lanes": dict(zip(("in_use", "on_distinct_hardware_queues"), deng.lanes_info())),
                                        "note": "a sweep is four kernels on four streams (lanes of 256 tiles): memory and matrix "
                                                "phases of different lanes overlap, back-to-back sweeps pipeline; DESIGN 9"},
                 "steps_fused_64_per_call": {"chain_steps_per_s": rn, "mfma_TFLOPs": rn * flop / 1e12, "mfma_frac": rn * flop / 1e12 / 78.6,
                                             "note": "state stays on chip between the steps of a call: matrix-bound"},
                 "nuts": {"leapfrog_steps_per_s": rt, "mfma_TFLOPs": rt * flop / 1e12, "mfma_frac": rt * flop / 1e12 / 78.6,
                          "mean_tree_depth": float(deng.tree_stats()["depth"].mean()),
                          "note": "workgroup-cooperative MFMA gradient inside k_nuts (DenseMvnCoop); one transition per launch",
                          "several_transitions_per_launch": {"transitions_per_launch": 20, "leapfrog_steps_per_s": rtf, "mfma_TFLOPs": rtf * flop / 1e12,
                                                             "mfma_frac": rtf * flop / 1e12 / 78.6, "power": dense_nuts_power,
                                                             "note": "the same kernel, (transition, chain) pairs from one queue per XCD; bit-identical"}}}
        deng.close()
        return dense

    if world == 1:
        eng.close()
        eng = None
    dense = guarded(leg_dense) if world == 1 else None

    # configs[4]'s exchange, outside the timed region, at EVERY N: a global-eps NUTS warm-up leg of the same density
    # (65 536 chains per GPU, random start, per-chain stepsize searches pooled into one eps, then T dual-averaging
    # transitions) through the library's OWN RCCL communicator (idhmc_comm_*): one 4-double all-reduce for the initial
    # stepsize and one per transition, enqueued by the C++ driver on the context's stream.  The record shows how many
    # ranks RCCL saw, how many all-reduces ran, and that every rank ended with the same eps bits.
    if eng is not None:
        eng.close()
        eng = None

    def leg_global_eps():
        T = args.global_eps_transitions
        gopt = pkg.default_options(eps_mode=pkg.EPS_GLOBAL, metric_mode=pkg.METRIC_SHARED)
        geng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, gopt, seed=1, first_chain=rank * C, device=local)
        geng.set_minv(sig ** 2)
        _keep = None
        native_error = None
        if dist is None:
            try:
                pkg.distributed.attach_global_eps_native(geng, rank=0, world=1)
            except Exception as e:     # noqa: BLE001 -- no loadable RCCL: a lone rank's exchange is a no-op, go on without a communicator
                native_error = "%s: %s" % (type(e).__name__, e)
        elif backend == "nccl":
            try:
                pkg.distributed.attach_global_eps_native(geng)
            except RuntimeError as e:      # raised on every rank together: the same exchange through torch.distributed's RCCL instead
                native_error = str(e)
                _keep = pkg.distributed.attach_global_eps(geng)
        else:       # gloo rehearsal (ranks share a device, which RCCL refuses): the same exchange through the hook
            _keep = pkg.distributed.attach_global_eps(geng)
        geng.random_position()
        geng.refresh_momentum(0)
        geng.synchronize()
        if dist is not None:
            dist.barrier()
        g0 = time.perf_counter()
        geng.find_initial_stepsize()
        geng.tuning_stage(T, False, 0, store_stats=False)
        geng.synchronize()
        g_el = time.perf_counter() - g0
        g_steps = geng.total_steps()
        g_eps = float(geng.eps[0])
        g_ranks, _, g_allreduces = geng.comm_info()
        eps_same = True
        if dist is not None:
            dev = "cuda" if backend == "nccl" else "cpu"
            tt = torch.tensor([g_el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            g_el = float(tt[0])
            ts = torch.tensor([float(g_steps)], dtype=torch.float64, device=dev)
            dist.all_reduce(ts, op=dist.ReduceOp.SUM)
            g_steps = int(ts[0])
            bits = torch.tensor([np.float64(g_eps).view(np.int64)], dtype=torch.int64, device=dev)
            lo, hi = bits.clone(), bits.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            eps_same = bool(int(lo[0]) == int(hi[0]))
        global_eps = {"workload": "configs[4] shape: %d chains per GPU x %d GPU(s), D=%d, global dual-averaging eps, "
                                  "stepsize search + %d warm-up transitions" % (C, world, D, T),
                      "exchange": ("library-owned RCCL communicator (idhmc_comm_*)" if native_error is None else "none (single rank, no communicator)")
                              if _keep is None else
                                  "torch.distributed hook (%s rehearsal)" % backend,
                      "rccl_ranks": g_ranks, "rccl_ranks_match_n_gpus": bool(g_ranks == world),
                      "allreduces": g_allreduces if (_keep is None and native_error is None) else T + 2,     # search + one per transition + the stage's status agreement
                      "allreduce_doubles": pkg.XCHG_DOUBLES,
                      "seconds": g_el, "leapfrog_steps_per_s": g_steps / g_el, "eps_final": g_eps,
                      "eps_bits_identical_across_ranks": eps_same, "native_communicator_error": native_error,
                      "note": "exchange = exact fixed-point record (include/idhmc.h): eps is bit-identical for any rank count"}
        geng.close()
        return global_eps

    global_eps = guarded(leg_global_eps)

    if rank == 0:
        value = C * world * args.steps / elapsed
        achieved = BYTES_PER_STEP * C / (ms_kernel * 1e-3) / 1e9
        # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command (the guide: FETCH_SIZE and
        # WRITE_SIZE cannot share a pass, counters perturb timing): NOT measured in this run -- the line says which file
        traffic, traffic_source, profile_kernel_ms = None, None, None
        if C == CHAINS_PER_GPU:                              # the PMC passes were taken at the default size
            # the kernel runs in one of two modes, decided by where the allocator put the state arrays: the profile of the
            # SAME mode is the evidence for this line (good placement: r03_leapfrog_pmc.json; bad: ..._badplacement.json when one exists)
            good = placement["single_array_GBps"] > 0 and placement["probe_GBps"] >= 1.10 * placement["single_array_GBps"]
            names = ("r03_leapfrog_pmc.json",) if good else ("r03_leapfrog_pmc_badplacement.json", "r03_leapfrog_pmc.json")
            for nm in names:
                prof, fname, fdate = committed_profile(nm)
                if prof is not None:
                    traffic = prof.get("hbm_bytes_per_launch")
                    profile_kernel_ms = prof.get("avg_duration_ns", 0.0) * 1e-6
                    traffic_source = {"file": fname, "kernel_stats": fname.replace("_pmc_badplacement.json", "_kernel_stats_badplacement.csv").replace("_pmc.json", "_kernel_stats.csv"),
                                      "file_date": fdate, "measured_in_this_run": False,
                                      "profile_placement_mode": prof.get("placement_mode"), "this_run_placement_mode": "good" if good else "bad",
                                      "profile_frac_of_8000": prof.get("frac_of_8000_from_profile"),
                                      "how": "rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE passes of bench.py "
                                             "(tools/profile_bench.sh), FETCH_SIZE x2 (gfx950)"}
                    break
        out = {
            "metric": "leapfrog-steps/sec (all chains), 1024-dim Gaussian, 1/2/4/8 GPU",
            "value": value, "unit": "leapfrog-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 1024-dim diagonal Gaussian, %d chains per GPU, fixed eps=%.2f, "
                                   "M^-1=sigma^2, one fused leapfrog sweep per step" % (C, EPS),
                       "chains_per_gpu": C, "dim": D, "parallelism": "chains sharded over %d GPU(s); no collective in the timed leapfrog sweep "
                                      "(the one RCCL exchange of the path is reported under global_eps_warmup)" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "profile_kernel_ms": profile_kernel_ms,
                         "kernel": "k_leapfrog1<8, DiagGaussian<8>>", "kernel_ms": ms_kernel, "kernel_ms_ranks": kernel_ms_ranks,
                         "algorithmic_bytes_per_launch": BYTES_PER_STEP * C,
                         "frac_of_measured_copy_peak_6290": achieved / 6290.0},
            "launch": {"mode": launch, "ranks": world, "devices_visible": ndev, "backend": backend if world > 1 else None},
            "state_finite": finite,
            "state_placement": placement,
        }
        if identity is not None:
            out["identity_metric"] = identity
        if regrad is not None:
            out["leapfrog_grad_recompute"] = regrad
        if nuts is not None:
            out["nuts"] = nuts
        if cfg3 is not None:
            out["cfg3_full"] = cfg3
        if dense is not None:
            out["dense"] = dense
        out["global_eps_warmup"] = global_eps
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(mu, sig, args.cpu_seconds)
            if nuts is not None and "error" not in nuts:        # the same oracle build (native), a short NUTS sample beside the nuts field
                nuts["cpu_baseline"] = cpu_nuts_baseline(mu, sig, out["cpu_baseline"]["cores"], min(4.0, args.cpu_seconds / 4))
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
