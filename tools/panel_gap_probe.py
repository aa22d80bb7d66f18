"""Does a panel pass take longer after the device sat (nearly) idle for a while?  The fill of a parked fit, timed with
HIP events (score_pass_stats), with host-side pauses of g microseconds between passes.  python tools/panel_gap_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

n, p = 50000, 10000
X, y, _, _ = synth.make_lm(n, p, 100)
with capi.Session(X, y, score_mode=2) as s:
    del X
    cols = (np.arange(8 * 32, dtype=np.int32) * 37 + 11) % p
    s.cov_prefill_begin(cols)
    for per_launch in (1, 2):
        for g in range(0, 8, per_launch):
            s.cov_prefill_compute(g, per_launch)
        for gap_us in (0, 50, 100, 250, 500, 1000, 3000, 0):
            s.enable_kernel_timing(True)
            s.score_pass_stats(reset=True)
            for r in range(24):
                s.cov_prefill_compute((r * per_launch) % 8, per_launch)
                t1 = time.perf_counter() + gap_us * 1e-6
                while time.perf_counter() < t1:
                    pass
            st = s.score_pass_stats()
            s.enable_kernel_timing(False)
            print(json.dumps({"groups_per_pass": per_launch, "idle_gap_us": gap_us,
                              "ms_per_pass": round(1e3 * st["seconds"] / st["launches"], 4)}), flush=True)
    s.cov_prefill_end()
