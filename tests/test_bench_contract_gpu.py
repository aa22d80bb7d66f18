"""bench.py's output contract on a small problem (the driver parses this line): one JSON object with the metric,
the roofline of the kernel that streams X (HIP-event timing), and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reject_constant(name):
    raise ValueError("not strict JSON: %s" % name)


def test_bench_line(gpu):
    detail = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    if os.path.exists(detail):
        os.unlink(detail)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "3000", "--p", "800", "--kmax", "30",
                          "--k-true", "10", "--steps", "2", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    # the driver parses the LAST line of stdout out of an 8 KB tail: one strict-JSON object, small
    last = lines[-1]
    assert len(last) < 6500, len(last)
    d = json.loads(last, parse_constant=_reject_constant)
    assert len([ln for ln in lines if ln.startswith("{")]) == 1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "workload" in d["config"] and "configs[1]" in d["config"]["workload"]
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 30 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-4 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the covariance form's panel launches: both roofs per launch width, the binding one on top, frac = achieved / peak
    assert (r["bound"], r["unit"], r["peak"]) in (("hbm", "GB/s", 8000.0), ("mfma", "TFLOP/s", 78.6))
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["achieved"] > 0
    assert abs(r["algorithmic_bytes_per_launch"] - 8.0 * 3000 * 800) < 1
    assert r["by_width"] and r["hbm_frac"] > 0 and r["mfma_fp64_frac"] > 0
    launches = 0
    for w, rec in r["by_width"].items():
        assert rec["bound"] == ("mfma" if rec["mfma_fp64_frac"] > rec["hbm_frac"] else "hbm")
        assert rec["flop_per_byte"] == (8.0 if w == "1_group" else 16.0)
        launches += rec["launches"]
    assert launches == r["launches_timed"]
    assert 0 < r["frac_sum_of_binding_roofs"] < 1.5
    c = d["cpu_baseline"]
    assert c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0 and c["sample"]
    assert d["streaming_score_pass"]["same_selection"] is True and d["streaming_score_pass"]["same_candidates"] is True
    # the streaming formulation is a first-class leg: same steps / warm-up, flat keys inside `roofline` and `config`
    assert d["streaming_score_pass"]["steps"] == 2 and d["streaming_score_pass"]["warmup"] == 1
    for k in ("streaming_candidates_per_s", "streaming_ms_per_step", "streaming_kernel_frac", "streaming_whole_step_frac"):
        assert r[k] > 0, k
    assert abs(r["streaming_candidates_per_s"] - 30 / (r["streaming_ms_per_step"] * 1e-3)) < 1e-4 * r["streaming_candidates_per_s"]
    assert "covariance" in d["config"]["headline_mode"]
    # group_XTX (src/path.cpp:37) is INSIDE the step; the figure without it is a side key
    assert d["group_XTX_ms_inside_step"] > 0
    assert d["value_excl_group_XTX"] > d["value"]
    assert 0.0 <= d["whole_step"]["kernel_streaming_X_share"] <= 1.0
    # the full record travels in a side file
    full = json.load(open(detail))
    assert sum(full["pdas_iterations_histogram"].values()) == 30
    assert full["whole_step"]["time"]["wall_ms"] > 0
    assert abs(full["value"] - d["value"]) < 1e-4 * d["value"]
