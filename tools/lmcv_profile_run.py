"""configs[3] paths (LM gs_path + 5-fold CV) with pauses between them, for tools/pipe_trace.sh (PIPE_TRACE_RUNNER)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

n, p = 50000, 10000
X, y, _, _ = synth.make_lm(n, p, 100)
with capi.Session(X, y) as s:
    s.set_cv(5, synth.make_cv_folds(n, 5))
    for _ in range(4):
        t0 = time.time()
        out = s.gs_path(1, 200, ic_type=3, is_cv=True)
        print("path ms", round(1e3 * (time.time() - t0), 2), flush=True)
        time.sleep(0.03)
    print(s.counters())
