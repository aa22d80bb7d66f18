"""Probe: two chunk chains of a GLM / Cox sequential path at the same time on one GPU (two sessions, a stream and a host
thread each; X read by both): does one chain's IRLS / Newton work overlap the other's passes over X?
   python tools/glm_two_chains_probe.py logistic|poisson|cox [C]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402
from bess_amd import dist as bdist  # noqa: E402

fam = sys.argv[1]
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if fam == "logistic":
    X, y, _, _ = synth.make_logistic(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=2), 100
elif fam == "poisson":
    X, y, _, _ = synth.make_poisson(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=3), 100
else:
    n, p = (int(os.environ.get("COX_N", 200000)), int(os.environ.get("COX_P", 20000)))
    X, _, y, _, _ = synth.make_cox(n, p, 75)
    kw, kmax = dict(data_type=3, model_type=4), 150
seq = np.arange(1, kmax + 1)
sess = [capi.Session(X, y, **kw) for _ in range(C)]
del X
t0 = time.time()
single = sess[0].sequential_path(seq, ic_type=3)
t_single = time.time() - t0
bounds = [bdist.partition(kmax, C, r)[0] for r in range(C)] + [kmax]
starts = {}
for r in range(1, C):
    h = sess[0].sequential_path_chain(seq[:bounds[r]], ic_type=3)
    starts[r] = (h["last_idx"], h["last_val"], h["last_coef0"])


def chunk(r):
    kw2 = dict(init_idx=starts[r][0], init_val=starts[r][1], init_coef0=starts[r][2]) if r else {}
    return sess[r].sequential_path_chain(seq[bounds[r]:bounds[r + 1]], ic_type=3, **kw2)


alone = []
for r in range(C):
    t0 = time.time()
    chunk(r)
    alone.append(time.time() - t0)
res = [None] * C
bar = threading.Barrier(C + 1)


def work(r):
    bar.wait()
    res[r] = chunk(r)
    bar.wait()


best = 1e9
for rep in range(2):
    th = [threading.Thread(target=work, args=(r,)) for r in range(C)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.time()
    bar.wait()
    best = min(best, time.time() - t0)
    for t in th:
        t.join()
same = sum(int(np.array_equal(res[r]["cand_support"][i, :bounds[r] + i + 1], single["cand_support"][bounds[r] + i, :bounds[r] + i + 1]))
           for r in range(C) for i in range(bounds[r + 1] - bounds[r]))
print(json.dumps({"family": fam, "chains": C, "single_chain_s": round(t_single, 4), "chunks_alone_s": [round(a, 4) for a in alone],
                  "sum_alone_s": round(sum(alone), 4), "all_at_once_s": round(best, 4),
                  "speedup_over_the_single_chain": round(t_single / best, 2), "supports_equal": same, "of": kmax}))
