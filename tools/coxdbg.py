import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from bess_amd import capi, synth
from oracle import port_ctypes as P
X, _, st, _, _ = synth.make_cox(600, 100, 6)
kw = dict(data_type=3, model_type=4, ic_type=4, sequence=[8])
want = P.trace(X, st, **kw)
s = capi.Session(X, st, data_type=3, model_type=4)
s.trace_enable(True)
got = s.sequential_path([8], ic_type=4)
for f in (want['fits'][0], got['trace']['fits'][0]):
    print(len(f['iters']))
    for a,b in zip(f['iters'], f['betas']): print(a, np.array2string(b, precision=10))
