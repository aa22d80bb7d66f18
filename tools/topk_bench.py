"""Timing of the top-k selection kernel (k_topk) at the BASELINE sizes and beyond (two-level for len > 32768)."""
import sys
sys.path.insert(0, '.')
from bess_amd import capi
for length in (5000, 10000, 20000, 32768, 100000):
    for k in (10, 100, 200):
        print("len %6d k %4d   %6.2f us" % (length, k, capi.op_topk_bench(length, k)))
