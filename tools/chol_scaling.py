"""Accuracy and timing of the register-resident Cholesky solve (k_chol) over the sizes it covers."""
import sys
sys.path.insert(0, '.')
import numpy as np
from bess_amd import capi
rng = np.random.default_rng(0)
for m in [15, 31, 47, 63, 79, 95, 127, 159, 191, 223, 254]:
    a = rng.standard_normal((m + 50, m)); g = a.T @ a + np.eye(m); b = rng.standard_normal(m)
    x = capi.op_chol_solve(g, b)
    print("m %3d  residual %.2e  %7.2f us" % (m, np.abs(g @ x - b).max(), capi.op_chol_bench(m)))
