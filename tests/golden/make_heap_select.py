"""Golden vectors for the heap-select branch of std::nth_element inside max_k (src/utilities.cpp:179-188): a score vector
a PDAS iteration of a fuzz case produced (63 distinct values, k = 10: the depth limit 2 floor(log2 63) = 10 of the
introselect runs out), and variants of it with EQUAL scores at the selection boundary -- there the selected set is
whatever libstdc++'s __heap_select leaves in front.  Expected selections come from the reference's own max_k (compiled
harness, oracle/_ref).  Run in the build container:  python tests/golden/make_heap_select.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bess_amd import synth  # noqa: E402
from oracle import port_ctypes as P, ref_ctypes as R  # noqa: E402
import ctypes  # noqa: E402

X, y, _, _ = synth.make_lm(800, 246, 8, seed=1040777304)
A = P.screening(X, y, None, 1, 63)
P.trace(np.ascontiguousarray(X[:, A]), y, sequence=np.arange(1, 12), ic_type=2)
buf, k0 = np.zeros(4096), ctypes.c_int(0)
n = P.lib().bess_oracle_last_heap_select(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 4096, ctypes.byref(k0))
base = buf[:n].copy()
assert n == 63 and k0.value == 10
order = np.argsort(-base)
scores, ks, want = [base], [10], [R.max_k(base, 10)]
for k in range(3, 30):
    for width in (2, 3, 4, 6):
        for lo in range(max(0, k - width + 1), k + 1):
            hi = lo + width
            if not (lo <= k - 1 and hi > k):
                continue
            sc = base.copy()
            sc[order[lo:hi]] = base[order[lo]]  # ranks lo .. hi-1 tie across the boundary between ranks k-1 and k
            h0 = P.nth_heap_selects()
            P.max_k(sc, k)
            if P.nth_heap_selects() != h0:
                scores.append(sc)
                ks.append(k)
                want.append(R.max_k(sc, k))
kmax = max(ks)
W = np.full((len(ks), kmax), -1, dtype=np.int32)
for i, w in enumerate(want):
    W[i, :len(w)] = w
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "heap_select_ties.npz")
np.savez(out, scores=np.array(scores), k=np.array(ks, dtype=np.int32), selected=W)
print("wrote", out, len(ks), "vectors")
