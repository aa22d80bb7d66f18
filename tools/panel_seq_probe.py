"""The panel passes of one configs[1] path -- pair, single, single, single, pair, single, 0.2 ms apart, then 5 ms of
nothing -- replayed alone through the prefill entry points: does the fifth launch take longer than the first?"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

n, p = 50000, 10000
X, y, _, _ = synth.make_lm(n, p, 100)


def pause(us):
    t1 = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t1:
        pass


with capi.Session(X, y, score_mode=2) as s:
    del X
    cols = (np.arange(8 * 32, dtype=np.int32) * 37 + 11) % p
    s.cov_prefill_begin(cols)
    for g in range(0, 8, 2):
        s.cov_prefill_compute(g, 2)
    seq = [(0, 2), (2, 1), (3, 1), (4, 1), (5, 2), (7, 1)]
    acc = [[] for _ in seq]
    for rep in range(12):
        for i, (g, k) in enumerate(seq):
            s.enable_kernel_timing(True)
            s.score_pass_stats(reset=True)
            s.cov_prefill_compute(g, k)
            st = s.score_pass_stats()
            acc[i].append(1e3 * st["seconds"] / max(1, st["launches"]))
            pause(200 if i != 0 else 0)
        pause(5000)
    print(json.dumps({"launch": ["%d group(s)" % k for _, k in seq], "median_ms": [round(float(np.median(a)), 3) for a in acc]}))
    s.cov_prefill_end()
