"""Per-chain event log (BESSX_TEST_HOOKS=kchunks_log=1) of one path with shared passes.  python tools/shared_pass_log.py lm|logistic C"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fam, C = sys.argv[1], int(sys.argv[2])
os.environ["BESSX_TEST_HOOKS"] = "kchunks_log=1"
from bess_amd import capi, synth  # noqa: E402
if fam == "lm":
    X, y, _, _ = synth.make_lm(50000, 10000, 100)
    kw, kmax = dict(score_mode=1), 200
else:
    X, y, _, _ = synth.make_logistic(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=2), 100
with capi.Session(X, y, **kw) as s:
    s.set_kpath_chains(C)
    for i in range(2):
        print("---- path", i, file=sys.stderr, flush=True)
        t0 = time.time()
        out = s.sequential_path(np.arange(1, kmax + 1), ic_type=3)
        print("---- %.1f ms" % (1e3 * (time.time() - t0)), file=sys.stderr, flush=True)
    print(s.counters())
