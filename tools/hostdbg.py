import sys, time, numpy as np
sys.path.insert(0,'.')
from bess_amd import capi, synth
X,y,_,_=synth.make_lm()
with capi.Session(X,y) as s:
    s.sequential_path(np.arange(1,201), ic_type=3)
    t=time.perf_counter(); s.sequential_path(np.arange(1,201), ic_type=3); print("ms", (time.perf_counter()-t)*1e3)
