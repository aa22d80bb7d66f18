"""Development aid: one chunk of a sharded k-path as a rank of `bench.py --gpus N` runs it -- the sparsity levels
k0 .. k0 + 24 of BASELINE configs[1] -- started cold (Algorithm::fit from the empty model at k0) and up a ladder
(warm-start chain over k0/8, k0/4, k0/2 first, `--chunk-start ladder`); time, passes over X, and whether the chunk's
supports equal the single chain's."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from bess_amd import capi, synth
X, y, _, _ = synth.make_lm()
with capi.Session(X, y) as s:
    single = s.sequential_path(np.arange(1, 201), ic_type=3)
    for k0 in (26, 51, 101, 151, 176):
        chunk = np.arange(k0, k0 + 25)
        for name, lead in (("cold", []), ("ladder", sorted({k for k in (k0 // 8, k0 // 4, k0 // 2) if 1 <= k < k0}))):
            seq = np.concatenate([np.array(lead, dtype=chunk.dtype), chunk])
            s.sequential_path(seq, ic_type=3)
            t = time.perf_counter(); o = s.sequential_path(seq, ic_type=3); dt = time.perf_counter() - t
            sup = o["cand_support"][len(lead):]
            same = sum(np.array_equal(sup[i, :k0 + i], single["cand_support"][k0 - 1 + i, :k0 + i]) for i in range(25))
            print("k0", k0, name, "ms %.2f" % (dt * 1e3), "passes", s.counters()["passes_over_X"], "supports equal to single chain %d/25" % same)
