"""one logged configs[1] path (BESSX_TEST_HOOKS=...,kchunks_log=1): what every chain did when"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
with capi.Session(X, y) as s:
    s.set_kpath_chains(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    for i in range(4):
        t0 = time.time()
        sys.stderr.write("==== path %d\n" % i)
        sys.stderr.flush()
        out = s.sequential_path(seq, ic_type=3)
        print("path ms", round(1e3 * (time.time() - t0), 2), flush=True)
