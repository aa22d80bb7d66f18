"""CPU: bench.py's rank launcher.  `--gpus N` without a launcher must start N ranks (the driver's N = 1 form is
`python bench.py --gpus 1`; ADVICE round 1: --gpus used to be parsed and ignored), and a launcher whose WORLD_SIZE
disagrees with --gpus is an error.  No GPU here, so the ranks stop at "needs a GPU" -- once per rank."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=3" in out.stderr


@pytest.mark.skipif(not _no_gpu(), reason="the GPU variant is tests/test_bench_contract_gpu.py")
@pytest.mark.timeout(300)
def test_gpus_2_starts_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BESSX_BENCH_ONE_DEVICE"] = "1"
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--n", "200", "--p", "20", "--kmax", "4"],
                         capture_output=True, text=True, env=env, timeout=280)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a GPU") == 2, out.stderr[-2000:]


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without devices")
def test_more_gpus_than_devices_is_refused_before_any_rank_starts():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BESSX_BENCH_ONE_DEVICE")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 2 and "visible" in out.stderr
