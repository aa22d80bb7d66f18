"""What would the k-path of configs[1] cost as C chunk chains on ONE shared Gram column cache (coarse chain over the
chunk boundaries first, chunks warm from its models, stitched)?  Run one after another in ONE session with keep_caches:
passes over X in all, and the time of each part.   python tools/chunked_onegpu_probe.py [C]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402
from bess_amd import dist as bdist  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, p, kmax = 50000, 10000, 200
X, y, _, _ = synth.make_lm(n, p, 100)
seq = np.arange(1, kmax + 1)
with capi.Session(X, y) as s:
    del X
    single = s.sequential_path(seq, ic_type=3)
    t0 = time.time()
    single = s.sequential_path(seq, ic_type=3)
    t_single = time.time() - t0
    p_single = s.counters()["passes_over_X"]
    for rep in range(2):
        base = s.counters()["passes_over_X"]
        bounds = [bdist.partition(kmax, C, r)[0] for r in range(C)] + [kmax]
        t0 = time.time()
        # coarse chain: the levels in front of every chunk but the first, warm from one another
        models, init = {}, None
        s.sequential_path_chain(seq[:1], ic_type=3)  # (cold caches; k = 1 is chunk 0's first candidate anyway)
        for r in range(1, C):
            kw = dict(init_idx=init[0], init_val=init[1], init_coef0=init[2]) if init else {}
            h = s.sequential_path_chain([int(seq[bounds[r] - 1])], ic_type=3, keep_caches=True, **kw)
            init = (h["last_idx"], h["last_val"], h["last_coef0"])
            models[r] = init
        t_coarse = time.time() - t0
        p_coarse = s.counters()["passes_over_X"] - base
        chunks, t_chunks = [], []
        for r in range(C):
            t1 = time.time()
            kw = dict(init_idx=models[r][0], init_val=models[r][1], init_coef0=models[r][2]) if r else {}
            chunks.append(s.sequential_path_chain(seq[bounds[r]:bounds[r + 1]], ic_type=3, keep_caches=True, **kw))
            t_chunks.append(time.time() - t1)
        p_chunks = s.counters()["passes_over_X"] - base - p_coarse
        refits, t_st = [], []
        for r in range(1, C):
            t1 = time.time()
            prev = chunks[r - 1]
            res = s.sequential_path_chain(seq[bounds[r]:bounds[r + 1]], ic_type=3, keep_caches=True,
                                          init_idx=prev["last_idx"], init_val=prev["last_val"], init_coef0=prev["last_coef0"],
                                          stop_support=chunks[r]["cand_support"], stop_beta=chunks[r]["cand_beta"])
            m = int(res["n_candidates"])
            chunks[r]["cand_support"][:m] = -1
            chunks[r]["cand_support"][:m, :res["cand_support"].shape[1]] = res["cand_support"][:m]
            refits.append(m)
            t_st.append(time.time() - t1)
        p_all = s.counters()["passes_over_X"] - base
        same = 0
        for r in range(C):
            for i in range(bounds[r + 1] - bounds[r]):
                k = bounds[r] + i
                same += int(np.array_equal(chunks[r]["cand_support"][i, :k + 1], single["cand_support"][k, :k + 1]))
        print(json.dumps({"chunks": C, "single_chain_ms": round(1e3 * t_single, 2), "single_chain_passes": p_single,
                          "coarse_ms": round(1e3 * t_coarse, 2), "coarse_passes": p_coarse,
                          "chunks_ms": [round(1e3 * t, 2) for t in t_chunks], "chunks_sum_ms": round(1e3 * sum(t_chunks), 2),
                          "chunk_passes": p_chunks, "stitch_ms": [round(1e3 * t, 2) for t in t_st], "refits": refits,
                          "passes_in_all": p_all, "supports_equal": same, "of": kmax}))
