"""Tuning aid: the one-pass Cox score kernel (k_cox_score1p) on a synthetic n x p matrix (bessx_op_cox_score_bench; a
minute on the GPU box instead of the 35 s set-up of a full-size Cox session).  Round 3 tried, each at 0.77-0.81 of 8 TB/s
like the kernel as it is: the four waves of a block on ONE row block (shared rows of the four n-vectors), no fences
around the private LDS tile; and 64-row load groups (512 contiguous bytes of two columns per wave instruction, the
pattern of k_xtv): 0.53.   python tools/cox_score_bench.py [n p repeats]"""
import ctypes
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bess_amd import capi  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
n, p, rep = (a + [200000, 20000, 5][len(a):])[:3]
variants = a[3:] or [1, 0]
L = capi.lib()
for v in variants:
    g, ms = ctypes.c_double(0), ctypes.c_double(0)
    rc = L.bessx_op_cox_score_bench(n, p, v, rep, ctypes.byref(g), ctypes.byref(ms))
    print("variant", v, "rc", rc, "GB/s %.0f" % g.value, "ms %.3f" % ms.value, "frac of 8 TB/s %.3f" % (g.value / 8000.0), flush=True)
