"""Development aid: time per phase of the conjugate-gradient solve kernels (k_cg / k_cgr), accumulated by the kernels
themselves (100 MHz wall clock).  Needs a library built with -DBESSX_CG_PROFILE:
  make -C bess_amd/csrc prof DEFS=-DBESSX_CG_PROFILE
  BESSX_LIB_PATH=bess_amd/csrc/build_prof/libbessx.so python tools/cgprof.py"""
import sys, ctypes, numpy as np
sys.path.insert(0,'.')
from bess_amd import capi, synth
X,y,_,_=synth.make_lm()
L=capi.lib()
buf=(ctypes.c_ulonglong*16)()
with capi.Session(X,y) as s:
    s.sequential_path(np.arange(1,201), ic_type=3)
    L.bessx_debug_cg_profile(buf,1)
    s.sequential_path(np.arange(1,201), ic_type=3)
    L.bessx_debug_cg_profile(buf,1)
v=list(buf)
n=v[15]
print("solves",n,"iters/solve",v[14]/n)
names=["idx setup","gather","rhs/x0","init resid","cg loop","final resid+loss","commit"]
for i,nm in enumerate(names): print("%-18s total %.3f ms  per solve %.2f us"%(nm, v[i]*10e-6, v[i]*0.01/n))
its=max(v[14],1)
for i,nm in [(7,"loop: top"),(8,"loop: matvec"),(9,"loop: ridge+dot+alpha"),(10,"loop: updates+dot+beta")]:
    print("%-26s per step %.3f us"%(nm, v[i]*0.01/its))
