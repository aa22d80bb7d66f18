"""Development aid: time per phase of the conjugate-gradient solve kernels (k_cg / k_cgr), accumulated by the kernels
themselves (100 MHz wall clock).  Needs a library built with -DBESSX_CG_PROFILE:
  make -C bess_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DBESSX_CG_PROFILE" -B all
  python tools/cgprof.py"""
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
