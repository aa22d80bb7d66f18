"""Probe: C warm-start chunk chains of configs[1] running AT THE SAME TIME on one GPU (one session, stream and host
thread each, caches already holding their columns) against the same chunks one after another -- do latency-bound chains
share the device without slowing each other?   python tools/concurrent_chains_probe.py [C]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402
from bess_amd import dist as bdist  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, p, kmax = 50000, 10000, 200
X, y, _, _ = synth.make_lm(n, p, 100)
seq = np.arange(1, kmax + 1)
sess = [capi.Session(X, y) for _ in range(C)]
del X
single = sess[0].sequential_path(seq, ic_type=3)
t0 = time.time()
single = sess[0].sequential_path(seq, ic_type=3)
t_single = time.time() - t0
starts = {}
for r in range(1, C):
    lo = bdist.partition(kmax, C, r)[0]
    h = sess[0].sequential_path_chain(seq[:lo], ic_type=3)
    starts[r] = (h["last_idx"], h["last_val"], h["last_coef0"])


def chunk(r):
    lo, hi = bdist.partition(kmax, C, r)
    kw = {}
    if r:
        kw = dict(init_idx=starts[r][0], init_val=starts[r][1], init_coef0=starts[r][2])
    return sess[r].sequential_path_chain(seq[lo:hi], ic_type=3, keep_caches=True, **kw)


for r in range(C):  # fills happen here
    sess[r].sequential_path_chain(seq[:1], ic_type=3)  # (cold caches)
    chunk(r)
alone = []
for r in range(C):
    b = 1e9
    for rep in range(3):
        t0 = time.time()
        chunk(r)
        b = min(b, time.time() - t0)
    alone.append(b)
res = [None] * C
bar = threading.Barrier(C + 1)


def work(r):
    bar.wait()
    res[r] = chunk(r)
    bar.wait()


best = 1e9
for rep in range(5):
    th = [threading.Thread(target=work, args=(r,)) for r in range(C)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.time()
    bar.wait()
    best = min(best, time.time() - t0)
    for t in th:
        t.join()
print(json.dumps({"chains": C, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "single_chain_ms": round(1e3 * t_single, 2),
                  "chunks_alone_ms": [round(1e3 * a, 2) for a in alone], "sum_alone_ms": round(1e3 * sum(alone), 2),
                  "all_at_once_ms": round(1e3 * best, 2)}))
