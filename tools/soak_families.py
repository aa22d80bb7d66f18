"""Determinism soak of the GLM / Cox / grouped-Cox paths: repeated paths on one session must be bit-identical
(fixed-order reductions, no floating-point atomics).  python tools/soak_families.py"""
import sys, numpy as np
sys.path.insert(0,'.')
from bess_amd import capi, synth
X,y,_,_=synth.make_logistic(20000,2000,20)
with capi.Session(X,y,data_type=2,model_type=2) as s:
    base=s.sequential_path(np.arange(1,41),ic_type=3)
    for r in range(15):
        o=s.sequential_path(np.arange(1,41),ic_type=3)
        for k in ("cand_support","cand_beta","cand_ic","cand_iters"): assert np.array_equal(o[k],base[k]),(r,k)
print("logistic soak ok")
X,_,st,_,_=synth.make_cox(20000,2000,15)
with capi.Session(X,st,data_type=3,model_type=4) as s:
    base=s.sequential_path(np.arange(1,31),ic_type=3)
    for r in range(8):
        o=s.sequential_path(np.arange(1,31),ic_type=3)
        for k in ("cand_support","cand_beta","cand_ic","cand_iters"): assert np.array_equal(o[k],base[k]),(r,k)
print("cox soak ok")
gi=np.arange(0,2000,4).astype(np.int32)
with capi.Session(X[:, :2000],st,data_type=3,model_type=4,algorithm_type=2,g_index=gi) as s:
    base=s.sequential_path(np.arange(1,6),ic_type=3)
    o=s.sequential_path(np.arange(1,6),ic_type=3)
    for k in ("cand_support","cand_beta","cand_ic","cand_iters"): assert np.array_equal(o[k],base[k]),k
print("cox groups soak ok", base["cand_iters"])
