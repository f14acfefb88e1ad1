"""bench.py's own launcher (`--gpus N` with no torch.distributed.run around it), the parts that need no GPU: the parent never
touches the GPU itself, and without enough devices it fails loudly BEFORE starting any rank (the product has no CPU path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


def test_gpus_2_without_gpus_fails_loudly_in_the_parent():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "256"])
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "bench.py --gpus 2" in out.stderr and "GPU" in out.stderr
    assert "Traceback" not in out.stderr                    # a message, not a crash


def test_flag_and_launcher_must_agree():
    """under a launcher, --gpus has to be the launcher's world size: a silent n_gpus = 1 line is what round 2 produced"""
    out = _bench(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and out.stdout.strip() == ""


def test_single_rank_without_gpu_says_so():
    import torch
    if torch.cuda.device_count() >= 1:
        pytest.skip("this machine has a GPU")
    out = _bench(["--steps", "2", "--warmup", "1", "--chains", "256"])
    assert out.returncode != 0 and "needs a GPU" in out.stderr and out.stdout.strip() == ""
