"""How fast is the library GEMM on the fill's shape -- X^T (p x n) times X_S (n x c), fp64, n = 50000, p = 10000?
(the panel kernel: 32 columns 0.76 ms, 64 columns 1.37 ms.)   python tools/gemm_probe.py"""
import time

import torch

n, p = 50000, 10000
for layout in ("col_major_X", "row_major_X"):
    if layout == "col_major_X":   # X stored column after column (the library's layout): X^T is a row-major p x n matrix
        Xt = torch.randn(p, n, dtype=torch.float64, device="cuda")
        A = Xt
    else:
        X = torch.randn(n, p, dtype=torch.float64, device="cuda")
        A = X.t()
    for c in (32, 64, 128, 256):
        B = (A[:c, :].t().contiguous() if layout == "col_major_X" else A[:c, :].t().contiguous())
        for _ in range(3):
            C = A @ B
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            C = A @ B
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 10
        print("%s: %3d columns: %.3f ms  (%.1f us per column, %.1f TFLOP/s, %.2f TB/s of X)" %
              (layout, c, 1e3 * dt, 1e6 * dt / c, 2.0 * n * p * c / dt / 1e12, 8.0 * n * p / dt / 1e12), flush=True)
    del A
