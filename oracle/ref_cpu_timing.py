"""Time the COMPILED REFERENCE (oracle/_ref/libbess_ref.so, the reference's own Eigen CPU path, package flags
-O2 -DNDEBUG -std=c++11, single thread by construction) on this host, on a bounded sample of BASELINE configs[1]:
the first candidates of the same path on the same full-size data.  Test/measurement infrastructure only.

  python oracle/ref_cpu_timing.py [kmax [n p]]
  BESS_REF_LIB=oracle/_ref/libbess_ref_fast.so OMP_NUM_THREADS=32 python oracle/ref_cpu_timing.py 3
      the "optimistic" build of the same sources (-O3 -march=x86-64-v3 -fopenmp; oracle/Makefile ref_fast)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bess_amd import synth  # noqa: E402
from oracle import ref_ctypes as R  # noqa: E402

kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, p = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50000, 10000)
X, y, _, _ = synth.make_lm(n, p, 100)
t0 = time.time()
t = R.trace(X, y, ic_type=3, sequence=[1])
t1 = time.time() - t0
t0 = time.time()
t = R.trace(X, y, ic_type=3, sequence=list(range(1, kmax + 1)))
tk = time.time() - t0
per_cand = (tk - t1) / (kmax - 1) if kmax > 1 else t1
fast = "fast" in os.path.basename(R.REF_LIB)
print(json.dumps({"kind": "reference (-O3 -march=x86-64-v3 -fopenmp)" if fast else "reference (package flags)",
                  "n": n, "p": p, "cores_used": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count())) if fast else 1,
                  "host_cores": os.cpu_count(),
                  "seconds_k1_incl_setup": t1, "seconds_k1..%d" % kmax: tk, "seconds_per_candidate_steady": per_cand,
                  "candidates_per_s_steady": 1.0 / per_cand, "candidates_per_s_incl_setup": kmax / tk,
                  "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")}))
