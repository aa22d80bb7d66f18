import sys; sys.path.insert(0, '.')
import numpy as np
from bess_amd import capi
rng = np.random.default_rng(0)
for m in [15, 31, 47, 63, 95, 127, 159, 191, 223, 255]:
    a = rng.standard_normal((m + 50, m)); g = a.T @ a + np.eye(m); b = rng.standard_normal(m)
    for _ in range(3):
        x = capi.op_chol_solve(g, b)
    print(m, np.abs(g @ x - b).max())
