"""bench.py prints ONE JSON line with the contract's keys (metric/value/unit/..., roofline, cpu_baseline), and the
2-rank launches -- `bench.py --gpus 2` starting its own ranks, and torch.distributed.run around it (gloo rehearsal: both ranks
on the box's one GPU) -- aggregate over ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]     # ONE line on stdout, nothing else (RCCL's banner goes to stderr)
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run([sys.executable, "bench.py", "--steps", "20", "--warmup", "2", "--chains", "2048", "--cpu-seconds", "1",
             "--cfg3-scale", "0.12"])
    for k in REQUIRED + ("cpu_baseline",):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 2 and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None                       # the PMC figure belongs to the default size only
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - 2048 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    assert d["state_finite"] is True
    assert d["state_placement"]["candidates_tried"] >= 1      # 2048 chains: arrays below the probe's threshold, one placement
    dn = d["dense"]                                   # configs[3] with both ceilings
    assert 0 < dn["single_step_sweeps"]["mfma_frac"] < 1 and 0 < dn["single_step_sweeps"]["hbm_frac"] < 1 and dn["nuts"]["leapfrog_steps_per_s"] > 0
    assert d["roofline"]["traffic_source"] is None and d["nuts"]["roofline"]["unit"] == "TFLOP/s"
    # the drivers' form of the same transitions: several per launch (idhmc_nuts_transitions)
    fz = d["nuts"]["several_transitions_per_launch"]
    assert fz["transitions_per_launch"] == 40 and fz["leapfrog_steps_per_s"] > 0 and fz["deep_trees"]["leapfrog_steps_per_s"] > 0
    assert dn["nuts"]["several_transitions_per_launch"]["leapfrog_steps_per_s"] > 0
    for pw in (d["nuts"]["power"], dn["single_step_sweeps"]["power"]):      # rocm-smi under load, in this run (None if it refuses)
        assert pw is None or (100 < pw["socket_power_W"] <= 1500 and 500 < pw["sclk_MHz"] <= 2500)
    f = d["cfg3_full"]                                # configs[2] end to end on a shortened schedule (9 / 3-6-12-24-48 / 6 + 24 draws)
    assert "error" not in f, f
    assert f["warmup"]["transitions"] == 9 + 3 + 6 + 12 + 24 + 48 + 6 and f["sampling"]["draws"] == 24
    assert f["warmup"]["leapfrog_steps_per_s"] > 0 and f["sampling"]["leapfrog_steps_per_s"] > 0
    assert f["chains_bit_identical"] is True and f["large_run_equals_small_run_bitwise"] is True and f["chains_compared"] == 64
    assert f["ess_per_transition"]["gpu_over_cpu"] == 1.0          # the same bits, hence the same estimate
    for k in ("acceptance_mean", "rhat_max", "ess_per_draw_min", "max_abs_mean_err_in_se", "var_ratio_range", "minv_over_sigma2_median"):
        assert k in f, k
    assert 0.5 < f["acceptance_mean"] <= 1.0 and f["termination"]["divergence"] == 0
    g = d["global_eps_warmup"]                        # the one RCCL exchange of the path, single-rank communicator here
    assert g["rccl_ranks"] == 1 and g["allreduces"] == 32 and g["eps_bits_identical_across_ranks"] is True


def test_two_rank_launch_aggregates():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(29600 + os.getpid() % 300), "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "2",
             "--chains", "2048"], env={"IDHMC_DIST_BACKEND": "gloo"})
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    g = d["global_eps_warmup"]                        # gloo rehearsal: the exchange runs through the hook
    assert g["eps_bits_identical_across_ranks"] is True and g["allreduces"] == 32 and "hook" in g["exchange"]
    assert abs(d["value"] - 2 * 2048 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses at N = 1): the parent starts the two ranks
    itself -- one worker per chain block, src/mcmc.jl:150-157 -- and relays rank 0's line"""
    env = {"IDHMC_DIST_BACKEND": "gloo"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        assert k not in os.environ, "the test must not run under a launcher"
    d = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "2", "--chains", "2048"], env=env)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert d["launch"]["mode"] == "self-spawned" and d["launch"]["ranks"] == 2
    assert abs(d["value"] - 2 * 2048 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    kr = d["roofline"]["kernel_ms_ranks"]
    assert 0 < kr["min"] <= kr["max"] == d["roofline"]["kernel_ms"]
    g = d["global_eps_warmup"]
    assert g["eps_bits_identical_across_ranks"] is True and g["allreduces"] == 32 and "hook" in g["exchange"]
    assert g["rccl_ranks_match_n_gpus"] is False           # gloo rehearsal on one device: no RCCL communicator, and the line says so


def test_a_failing_rank_fails_the_launch():
    """a rank that dies takes the launch down with a non-zero exit instead of leaving the others in a collective"""
    e = dict(os.environ, IDHMC_DIST_BACKEND="gloo", IDHMC_BENCH_FAIL_RANK="1")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--chains", "2048"],
                         cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "rank 1 exited" in out.stderr and out.stdout.strip() == ""
