"""Multi-process rehearsals of `bench.py --gpus N` on the ONE device of the GPU box (BESSX_BENCH_ONE_DEVICE=1, gloo):
N ranks started by bench.py itself, sharing device 0 with this pytest process.

These run LAST (the file name sorts last and tests/conftest.py orders multi-process files behind every parity test):
round 4's driver run stopped under `-x` at a flake here and never reached the 383 parity tests behind it.  A rank
that fails writes its own error to BESSX_BENCH_ERRDIR (bench._report_rank_failure); the assertion shows those files
first -- the launcher's epilogue and the peers' "connection closed" tracebacks bury the one message that matters."""
import glob
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_errors(errdir):
    out = []
    for f in sorted(glob.glob(os.path.join(errdir, "*.err"))):
        with open(f) as fh:
            out.append("---- %s\n%s" % (os.path.basename(f), fh.read()[-4000:]))
    return "\n".join(out) or "(no per-rank error file)"


def _run_bench(extra, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    with tempfile.TemporaryDirectory(prefix="bessx_bench_err_") as errdir:
        e["BESSX_BENCH_ERRDIR"] = errdir
        e["BESSX_BENCH_DETAIL_PATH"] = os.path.join(errdir, "detail.json")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "3000", "--p", "800", "--kmax", "30",
                              "--k-true", "10", "--steps", "2", "--warmup", "1"] + extra, cwd=ROOT, capture_output=True,
                             text=True, timeout=timeout, env=e)
        assert out.returncode == 0, "bench.py %s -> rc %d\n%s\n==== [rank ..] lines on stdout\n%s\n==== stderr (tail)\n%s" % (
            " ".join(extra), out.returncode, _rank_errors(errdir),
            "\n".join(ln for ln in out.stdout.splitlines() if ln.startswith("[rank")), out.stderr[-1500:])
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) < 6500, out.stdout[-2000:]
        line = json.loads(lines[0])  # the compact line the driver parses ...
        with open(os.path.join(errdir, "detail.json")) as fh:
            full = json.load(fh)     # ... and the full record of the same run (bench.py: emit)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "config"):
        assert k in line, k
    assert abs(line["value"] - full["value"]) <= 1e-4 * full["value"] and line["n_gpus"] == full["n_gpus"]
    return full


def test_bench_gpus_2_starts_two_ranks_and_shards_the_k_path(gpu):
    """`python bench.py --gpus 2` with no launcher starts 2 ranks itself; the default N > 1 mode is the strong-scaling
    k-path split of ONE problem.  On the one-GPU box the ranks share the device (BESSX_BENCH_ONE_DEVICE=1, gloo)."""
    d = _run_bench(["--gpus", "2", "--rebalance", "off"], {"BESSX_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert abs(d["value"] - 30 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]  # ONE problem: 30 candidates / step
    rep = d["kpath_chunks_vs_single_chain"]
    assert rep["chunks"] == [[1, 15], [16, 30]] and rep["of"] == 30
    # the chunks are stitched into the single warm-start chain: EVERY candidate's support equals the single chain's
    assert rep["supports_equal_to_single_chain"] == rep["of"] and rep["differing_k"] == []
    assert rep["best_k_chunked"] == rep["best_k_single_chain"] and rep["ic_curve_max_rel_diff_to_single_chain"] < 1e-10
    assert rep["stitch_refits"] >= 1 and rep["stitch_rounds"] >= 1 and len(rep["stitch_seconds_per_rank"]) == 2
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    # the default partition is north_star's: replicas only, no Gram blocks between the ranks
    assert rep["prefill_columns"] == 0 and rep["pilot"] is None and rep["prefill"].startswith("none")
    assert "Gram column blocks" not in d["config"]["collective"] and "cooperative_prefill_variant" not in d
    lad = _run_bench(["--gpus", "2", "--chunk-start", "ladder"], {"BESSX_BENCH_ONE_DEVICE": "1"})
    rep = lad["kpath_chunks_vs_single_chain"]
    assert rep["supports_equal_to_single_chain"] == rep["of"] and rep["chunk_start"] == "ladder"


def test_bench_three_ranks_with_the_pilot_prefill(gpu):
    """--gpus 3 on the one-GPU box with the cooperative prefill and the pilot fit in front of the chunks: two data-path
    all-gathers of Gram column blocks (gloo here), and still every candidate of the single chain."""
    d = _run_bench(["--gpus", "3", "--prefill", "64", "--pilot", "12,64", "--no-cpu-baseline", "--rebalance", "off"],
                   {"BESSX_BENCH_ONE_DEVICE": "1"})
    rep = d["kpath_chunks_vs_single_chain"]
    assert d["n_gpus"] == 3 and rep["chunks"] == [[1, 10], [11, 20], [21, 30]]
    assert rep["supports_equal_to_single_chain"] == rep["of"] == 30 and rep["differing_k"] == []
    assert rep["prefill_columns"] == 64 and rep["pilot"] == [12, 64] and min(rep["prefill_seconds_per_rank"]) > 0
    assert "cooperative prefill" in d["config"]["collective"]
    # opt-in second figure beside the default partition: same steps, the measured prefill policy (320 columns at 3 ranks)
    d = _run_bench(["--gpus", "3", "--coop-variant", "--no-cpu-baseline", "--rebalance", "off"], {"BESSX_BENCH_ONE_DEVICE": "1"})
    rep, var = d["kpath_chunks_vs_single_chain"], d["cooperative_prefill_variant"]
    assert rep["prefill_columns"] == 0 and rep["supports_equal_to_single_chain"] == rep["of"] == 30
    assert var["prefill_columns"] == 320 and var["value"] > 0 and var["ic_curve_equal_to_default_partition"] is True


def test_bench_three_ranks_with_shared_fills_in_the_pilot_and_moving_chunk_boundaries(gpu):
    """The pilot fit's own fills shared between the ranks (one 32-column group each per fill) and chunk boundaries that
    move after every step towards equal time per rank: cache contents and starting points only -- every candidate of
    the single chain, whatever the boundaries of the last step were."""
    d = _run_bench(["--gpus", "3", "--prefill", "32", "--pilot", "12,0,96", "--no-cpu-baseline", "--rebalance", "on",
                    "--steps", "3"], {"BESSX_BENCH_ONE_DEVICE": "1"})
    rep = d["kpath_chunks_vs_single_chain"]
    assert d["n_gpus"] == 3 and rep["rebalance"] is True and rep["pilot"] == [12, 0, 96]
    ch = rep["chunks"]
    assert ch[0][0] == 1 and ch[-1][1] == 30 and all(a[1] + 1 == b[0] for a, b in zip(ch, ch[1:]))
    assert all(b >= a for a, b in ch)
    assert rep["supports_equal_to_single_chain"] == rep["of"] == 30 and rep["differing_k"] == []
    assert rep["best_k_chunked"] == rep["best_k_single_chain"]


def test_bench_weak_scaling_and_cv_workload_on_two_ranks(gpu):
    d = _run_bench(["--gpus", "2", "--shard", "replica", "--no-cpu-baseline"], {"BESSX_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["ic_curves_gathered"] == 2
    assert abs(d["value"] - 2 * 30 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    one = _run_bench(["--workload", "lm-cv-gs", "--no-cpu-baseline"])
    two = _run_bench(["--gpus", "2", "--workload", "lm-cv-gs", "--no-cpu-baseline"], {"BESSX_BENCH_ONE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    for k in ("selected_k", "fits_per_step", "pdas_iterations_per_step"):
        assert one[k] == two[k], k
    assert abs(one["cv_loss"] - two["cv_loss"]) <= 1e-12 * abs(one["cv_loss"])


def test_bench_cox_k_path_on_two_ranks(gpu):
    """`bench.py --workload cox-seq --gpus 2` (BASELINE configs[4] at reduced size): the Cox k-path in two contiguous
    chunks, IC curve all-gathered, chunks compared with the single chain after the timed region; both ways a chunk can
    reach its first sparsity level."""
    for start in ("ladder", "cold"):
        d = _run_bench(["--gpus", "2", "--workload", "cox-seq", "--chunk-start", start], {"BESSX_BENCH_ONE_DEVICE": "1"})
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["metric"].startswith("candidate subsets solved/sec (Cox")
        assert d["config"]["ranks_with_work"] == 2 and d["config"]["chunk_start"] == start
        assert abs(d["value"] - 30 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
        rep = d["kpath_chunks_vs_single_chain"]
        assert rep["chunks"] == [[1, 15], [16, 30]] and rep["of"] == 30 and rep["chunk_start"] == start
        assert rep["supports_equal_to_single_chain"] == rep["of"] and rep["best_k_chunked"] == rep["best_k_single_chain"]
        assert rep["ic_curve_max_rel_diff_to_single_chain"] < 1e-10 and rep["stitch_refits"] >= 1
        assert d["roofline"]["kernel"].startswith("k_cox_score1p") and d["roofline"]["achieved"] > 0
    one = _run_bench(["--workload", "cox-seq"])
    assert one["n_gpus"] == 1 and one["selected_k"] == d["kpath_chunks_vs_single_chain"]["best_k_single_chain"]
