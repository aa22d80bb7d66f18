"""bench.py's output contract on a small problem (the driver parses this line): one JSON object with the metric,
the roofline of the kernel that streams X (HIP-event timing), and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "3000", "--p", "800", "--kmax", "30",
                          "--k-true", "10", "--steps", "2", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 30 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert abs(r["algorithmic_bytes_per_launch"] - 8.0 * 3000 * 800) < 1
    c = d["cpu_baseline"]
    assert c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0
    assert d["streaming_score_pass"]["same_selection"] is True and d["streaming_score_pass"]["same_candidates"] is True
    # the streaming formulation is a first-class leg: same steps / warm-up, flat keys inside `roofline` and `config`
    assert d["streaming_score_pass"]["steps"] == 2 and d["streaming_score_pass"]["warmup"] == 1
    for k in ("streaming_candidates_per_s", "streaming_ms_per_step", "streaming_kernel_frac", "streaming_whole_step_frac",
              "mfma_fp64_frac"):
        assert r[k] > 0, k
    assert abs(r["streaming_candidates_per_s"] - 30 / (r["streaming_ms_per_step"] * 1e-3)) < 1e-6 * r["streaming_candidates_per_s"]
    assert "covariance" in d["config"]["headline_mode"]
    assert d["group_XTX_ms_outside_step"] > 0
    assert abs(d["ms_per_step_incl_group_XTX"] - d["ms_per_step"] - d["group_XTX_ms_outside_step"]) < 1e-9
    t = d["whole_step"]["time"]
    assert 0.0 <= t["kernel_streaming_X_share"] <= 1.0
    assert all(v >= 0 for v in t.values())
    assert sum(d["pdas_iterations_histogram"].values()) == 30
