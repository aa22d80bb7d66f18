import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth
fam = sys.argv[1] if len(sys.argv) > 1 else "logistic"
if fam == "logistic":
    X, y, _, _ = synth.make_logistic(100000, 5000, 50)
    s = capi.Session(X, y, data_type=2, model_type=2)
    seq = np.arange(1, 101)
else:
    X, y, _, _ = synth.make_lm(50000, 10000, 100)
    s = capi.Session(X, y, score_mode=1)
    seq = np.arange(1, 201)
with s:
    for i in range(2):
        sys.stderr.write("== path %d\n" % i); sys.stderr.flush()
        s.sequential_path(seq, ic_type=3)
