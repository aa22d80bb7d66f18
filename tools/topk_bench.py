"""Timing of the two top-k kernels (bit-by-bit vs radix threshold search) at the BASELINE sizes."""
import sys
sys.path.insert(0, '.')
from bess_amd import capi
for length in (5000, 10000, 20000, 32768, 100000):
    for k in (10, 100, 200):
        a = capi.op_topk_bench(length, k, 0)
        b = capi.op_topk_bench(length, k, 1)
        print("len %6d k %4d   bitwise %6.2f us   radix %6.2f us" % (length, k, a, b))
