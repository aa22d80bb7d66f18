"""Development aid: the eight chunks of BASELINE configs[4]'s k-path (Cox n=200000 p=20000, k = 1..150) as the ranks of
`bench.py --workload cox-seq --gpus 8` would run them, timed one after another on ONE GPU: cold and ladder start, and
whether the chunk's supports equal the single chain's.  The slowest chunk bounds the 8-GPU step."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from bess_amd import capi, synth
from bess_amd import dist as bdist
X, _, st, _, _ = synth.make_cox()
with capi.Session(X, st, data_type=3, model_type=4) as s:
    del X
    s.sequential_path(np.arange(1, 4), ic_type=3)
    t = time.perf_counter(); single = s.sequential_path(np.arange(1, 151), ic_type=3); t1 = time.perf_counter() - t
    print("single chain k=1..150: %.3f s" % t1, flush=True)
    for r in range(8):
        lo, hi = bdist.partition(150, 8, r)
        chunk = np.arange(lo + 1, hi + 1)
        k0 = lo + 1
        for name, lead in (("cold", []), ("ladder", sorted({k for k in (k0 // 8, k0 // 4, k0 // 2) if 1 <= k < k0}))):
            if r == 0 and name == "ladder":
                continue
            seq = np.concatenate([np.array(lead, dtype=chunk.dtype), chunk])
            t = time.perf_counter(); o = s.sequential_path(seq, ic_type=3); dt = time.perf_counter() - t
            sup = o["cand_support"][len(lead):]
            same = sum(np.array_equal(sup[i, :k0 + i], single["cand_support"][k0 - 1 + i, :k0 + i]) for i in range(len(chunk)))
            print("rank %d k=%d..%d %s: %.3f s, PDAS iterations %d, supports equal to single chain %d/%d" %
                  (r, lo + 1, hi, name, dt, o["n_pdas_iters"], same, len(chunk)), flush=True)
