import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth
X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
with capi.Session(X, y) as s:
    s.enable_kernel_timing(True)
    for i in range(4):
        sys.stderr.write("== path %d\n" % i); sys.stderr.flush()
        s.sequential_path(seq, ic_type=3)
