"""Development aid: run a short configs[1] path with whatever library BESSX_LIB_PATH names and ignore errors
(instrumented builds of the panel kernel that skip loads / stores / barriers compute garbage); meant to sit under
rocprofv3 --kernel-trace --stats so that the panel kernel's time can be read off."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth
X, y, _, _ = synth.make_lm()
try:
    with capi.Session(X, y) as s:
        s.sequential_path(np.arange(1, 41), ic_type=3)
        print("ok")
except Exception as e:
    print("error (expected for instrumented builds):", str(e)[:100])
