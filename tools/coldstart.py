"""Development aid: a chunk of a sharded path started cold (what a rank of `bench.py --shard kpath` runs): time of
k0..k0+24 from an empty model, with and without the pair panel kernel for fills of two groups."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from bess_amd import capi, synth
X, y, _, _ = synth.make_lm()
for auto in ("1", "0"):
    os.environ["BESSX_PANEL_PAIR_AUTO"] = auto
    with capi.Session(X, y) as s:
        for k0 in (26, 101, 176):
            seq = np.arange(k0, k0 + 25)
            s.sequential_path(seq, ic_type=3)
            t = time.perf_counter(); o = s.sequential_path(seq, ic_type=3); dt = time.perf_counter() - t
            print("pair_auto", auto, "k0", k0, "ms %.2f" % (dt * 1e3), "passes", s.counters()["passes_over_X"], "ic %.6f" % float(np.min(o["cand_ic"])))
