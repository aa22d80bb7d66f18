"""The final stdout line of bench.py must stay small and strict JSON whatever the run collected (CPU test: no GPU needed,
bench.compact_line is pure Python).  Round 5's line was 20 KB and the driver's parser returned null."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reject_constant(name):
    raise ValueError("not strict JSON: %s" % name)


def test_compact_line_stays_small_for_a_large_record():
    """The line of the default run (all other configs, segments, histograms): round 5's was 20 KB and the driver parsed
    nothing.  Built here from a synthetic full-size record."""
    sys.path.insert(0, ROOT)
    import bench
    big = {"metric": "m", "value": 1.0, "unit": "candidates/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.0,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w" * 200, "headline_mode": "h" * 300, "units_sharded": "u" * 100},
           "roofline": dict({"bound": "mfma", "achieved": 1.0, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.5, "traffic": 4.3e9,
                             "traffic_source": "t" * 400, "by_width": {"1_group": {"launches": 2}, "2_groups": {"launches": 3}}},
                            **{"streaming_%d" % i: float("nan") if i == 0 else 1.0 / 3 for i in range(12)}),
           "cpu_baseline": {"value": 0.02, "unit": "candidates/s", "cores": 1, "kind": "reference", "sample": "s" * 300,
                            "segments": [{"seconds_per_candidate": [1.0 / 3] * 40}] * 2},
           "pdas_iterations_histogram": {str(i): i for i in range(20)},
           "other_configs": {name: {"candidates_per_s": 1.0 / 3, "ms_per_path": 2.0 / 3, "passes_over_X": 100.0,
                                    "whole_path_frac_of_hbm": 0.5, "score_kernel": {"frac": 0.5, "junk": "j" * 500},
                                    "workload": "x" * 400, "cpu_baseline": {"value": 1.0, "kind": "reference", "cores": 1,
                                                                            "segments": ["y" * 1000]}}
                             for name in ("lmcv", "grouped_lm", "powell_l0l2", "screened_lm", "default_sequence", "poisson",
                                          "logistic", "cox")}}
    text = json.dumps(bench.compact_line(big), allow_nan=False)
    assert len(text) <= bench.LINE_LIMIT
    d = json.loads(text, parse_constant=_reject_constant)
    assert d["roofline"]["streaming_0"] is None and set(d["other_configs"]) == set(big["other_configs"])
    assert d["cpu_baseline"]["kind"] == "reference" and "segments" not in d["cpu_baseline"]
