"""The panel kernels of the covariance form alone (k_cov_panel_dp, one or two 32-column groups per pass over X; with
BESSX_TEST_HOOKS=panel=lds the kernels of rounds 2-4: k_cov_panel_lds2, 32 Gram columns per pass, and k_cov_panel_pair,
64 per pass): timed with HIP events through the cooperative-prefill entry points (bessx_session_cov_prefill_*), which
run exactly the fill a parked fit runs -- list, panel, reduce -- on columns of the caller's choice.
  python tools/panel_bench.py [n p] [repeats] [--check]
Prints one JSON line per variant: ms per launch, TB/s of X streamed, TFLOP/s on the fp64 matrix cores, both against
the peaks of MI355X_MICROARCH.md.  --check: first compares both kernels with NumPy on a small problem.
Sits under rocprofv3 --pmc (tools/collect_profiles_r05.sh) for the counters of profiles/r05_panel_*."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, p = (int(args[0]), int(args[1])) if len(args) >= 2 else (50000, 10000)
reps = int(args[2]) if len(args) >= 3 else 20
HBM, MFMA = 8000.0, 78.6

if "--check" in sys.argv:
    Xs, ys, _, _ = synth.make_lm(4000, 1500, 10)
    Xc = Xs - Xs.mean(axis=0)
    Xn = np.sqrt(4000.0) * Xc / np.sqrt((Xc * Xc).sum(axis=0))
    cols = np.arange(7, 7 + 3 * 128, 3, dtype=np.int32)[:128]
    want = Xn.T @ Xn[:, cols]  # (p x 128)
    with capi.Session(Xs, ys, score_mode=2) as s:
        s.cov_prefill_begin(cols)
        s.cov_prefill_compute(0, 1)   # one group: k_cov_panel_lds2
        s.cov_prefill_compute(1, 1)
        s.cov_prefill_compute(2, 2)   # two groups: k_cov_panel_pair
        got = s.cov_prefill_export(0, 4).reshape(128, 1500).T
        s.cov_prefill_end()
    err = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    print(json.dumps({"check": "panel kernels against NumPy X^T X_S (n=4000, p=1500, 128 columns)", "max_rel_err": err}))
    assert err < 1e-12, err

X, y, _, _ = synth.make_lm(n, p, min(100, p // 4))
with capi.Session(X, y, score_mode=2) as s:
    del X
    ngroups = 8
    cols = (np.arange(ngroups * 32, dtype=np.int32) * 37 + 11) % p
    s.cov_prefill_begin(cols)
    lds = "panel=lds" in os.environ.get("BESSX_TEST_HOOKS", "")  # (round 5: k_cov_panel_dp is the default)
    for name, per_launch in (((("k_cov_panel_lds2" if lds else "k_cov_panel_dp") + " (32 columns per pass)"), 1),
                             ((("k_cov_panel_pair" if lds else "k_cov_panel_dp") + " (64 columns per pass)"), 2)):
        for g in range(0, ngroups, per_launch):  # warm-up: code objects, clocks
            s.cov_prefill_compute(g, per_launch)
        s.enable_kernel_timing(True)
        s.score_pass_stats(reset=True)
        t0 = time.time()
        for r in range(reps):
            s.cov_prefill_compute((r * per_launch) % ngroups, per_launch)
        wall = time.time() - t0
        st = s.score_pass_stats()
        s.enable_kernel_timing(False)
        ms = 1e3 * st["seconds"] / st["launches"]
        gbps = 8.0 * n * p / (ms * 1e-3) / 1e9
        tf = 2.0 * n * p * 32 * per_launch / (ms * 1e-3) / 1e12
        print(json.dumps({"kernel": name, "n": n, "p": p, "launches": st["launches"], "ms_per_launch": round(ms, 4),
                          "X_streamed_GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM, 4),
                          "fp64_mfma_TFLOPs": round(tf, 2), "frac_of_fp64_mfma_peak": round(tf / MFMA, 4),
                          "wall_ms_per_call_incl_reduce_and_sync": round(1e3 * wall / reps, 4)}), flush=True)
    s.cov_prefill_end()
