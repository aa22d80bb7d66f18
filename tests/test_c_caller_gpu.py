"""The drop-in boundary from a plain C host: tests/c_caller/c_caller.c is compiled against include/bessx.h,
linked with bess_amd/libbessx.so and run as a process of its own -- no Python, no torch, GPU_MAX_HW_QUEUES and every
BESSX_* variable unset -- on a problem written to disk; its candidates must be those of the ctypes binding.
At full size (configs[1]) the same program also gives the time per path a C or R host gets, which the library must
reach without the caller exporting anything (round 4 needed GPU_MAX_HW_QUEUES=8, set by `import bess_amd`)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from bess_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_caller", "c_caller.c")


def _build(tmp_path):
    exe = str(tmp_path / "c_caller")
    lib_dir = os.path.join(ROOT, "bess_amd")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"), SRC,
                           "-o", exe, "-L", lib_dir, "-lbessx", "-Wl,-rpath," + lib_dir])
    return exe


def _run(exe, tmp_path, X, y, kmax, repeats):
    xf, yf = str(tmp_path / "X.bin"), str(tmp_path / "y.bin")
    np.ascontiguousarray(X, dtype=np.float64).tofile(xf)
    np.ascontiguousarray(y, dtype=np.float64).tofile(yf)
    env = {k: v for k, v in os.environ.items() if not (k.startswith("BESSX_") or k == "GPU_MAX_HW_QUEUES")}
    out = subprocess.run([exe, xf, yf, str(X.shape[0]), str(X.shape[1]), str(kmax), str(repeats)], capture_output=True,
                         text=True, timeout=900, env=env)
    os.unlink(xf)
    assert out.returncode == 0, out.stderr[-2000:]
    sup, iters, best, ms, chains = {}, None, None, None, None
    for ln in out.stdout.splitlines():
        w = ln.split()
        if w[0] == "support":
            sup[int(w[1])] = np.array([int(v) for v in w[3:]], dtype=np.int32)
        elif w[0] == "iters":
            iters = np.array([int(v) for v in w[1:]], dtype=np.int32)
        elif w[0] == "best":
            best = (int(w[1]), float(w[2]))
        elif w[0] == "ms_per_path":
            ms = (float(w[1]), float(w[2]))
        elif w[0] == "chains":
            chains = (int(w[1]), int(w[3]))
    return sup, iters, best, ms, chains


def test_c_host_gets_the_candidates_of_the_python_binding(gpu, tmp_path):
    exe = _build(tmp_path)
    X, y, _, _ = synth.make_lm(3000, 800, 10)
    sup, iters, best, _, _ = _run(exe, tmp_path, X, y, 30, 0)
    with gpu.Session(X, y) as s:
        want = s.sequential_path(np.arange(1, 31), ic_type=3)
    assert sorted(sup) == list(range(1, 31))
    for k in range(1, 31):
        assert np.array_equal(sup[k], want["cand_support"][k - 1][:k]), k
    assert np.array_equal(iters, want["cand_iters"])
    assert best[0] == want["best_T0"] and abs(best[1] - want["ic"]) <= 1e-12 * abs(want["ic"])


def test_c_host_at_full_size_without_any_environment_variable(gpu, tmp_path):
    """configs[1] from the C host: every candidate of the compiled reference's golden, and the time per path a caller
    gets who exports nothing (printed; the bench line's `c_host` carries the same measurement)."""
    exe = _build(tmp_path)
    X, y, _, _ = synth.make_lm()
    sup, iters, best, ms, chains = _run(exe, tmp_path, X, y, 200, 10)
    g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_lm.npz"))
    assert list(iters) == list(g["fit_iters"])
    off = 0
    for it, t in zip(g["fit_iters"], g["fit_T0"]):
        last = off + (int(it) - 1) * int(t)
        assert np.array_equal(sup[int(t)], g["A_flat"][last:last + t]), int(t)
        off += int(it) * int(t)
    assert best[0] == 100 and abs(best[1] - float(g["best_ic"])) <= 1e-9 * abs(float(g["best_ic"]))
    print("C host, configs[1], no environment variable: %.2f ms per path (min), %.2f (median); chains %s" % (ms + (chains,)))
    assert ms[0] < 40.0  # (one chain took 18.6 ms in round 4; a regression to the streaming form's 190 ms would show)
