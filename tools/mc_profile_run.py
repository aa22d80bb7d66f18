import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth
X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
with capi.Session(X, y) as s:
    s.set_kpath_chains(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    for _ in range(6):
        out = s.sequential_path(seq, ic_type=3)
    print(s.counters())
