"""The multi-chain score pass (k_xtv_mc) against the single pass (k_xtv<8,16>) on an n x p matrix: ms per launch and
GB/s of X (8 n p bytes per launch, whatever the number of chains it serves).  python tools/xtv_multi_bench.py [n p]"""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bess_amd import capi  # noqa: E402

n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50000, 10000)
g, ms = capi.op_xtv_bench(n, p, 0, 20)
rows = [{"kernel": "k_xtv<8,16,false>", "chains": 1, "two": False, "ms": ms, "GBps": g}]
for two in (False, True):
    for nc in (1, 2, 3, 4, 6, 8):
        g, ms = capi.op_xtv_multi_bench(n, p, nc, two, 20)
        rows.append({"kernel": "k_xtv_mc", "chains": nc, "two": two, "ms": ms, "GBps": g, "ms_per_chain": ms / nc})
for r in rows:
    print(json.dumps(r))
