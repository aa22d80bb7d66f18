"""Waits of the host on the device have a deadline (BESSX_WAIT_TIMEOUT_S): a stream that is stuck -- here behind a
host function that sleeps -- makes the call return BESSX_ERR_HIP with the stream's status instead of spinning for ever."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, time
import numpy as np
sys.path.insert(0, %r)
from bess_amd import capi, synth
X, y, _, _ = synth.make_lm(800, 120, 5)
mode = sys.argv[1]
with capi.Session(X, y) as s:
    if mode == "cv":
        s.set_cv(4, synth.make_cv_folds(800, 4))
        s.gs_path(1, 12, ic_type=3, is_cv=True)
    else:
        s.sequential_path(np.arange(1, 8), ic_type=3)
    s.debug_block_stream(2500)
    t0 = time.time()
    try:
        if mode == "cv":
            s.gs_path(1, 12, ic_type=3, is_cv=True)
        elif mode == "fit":
            s.fit(5)
        else:
            s.sequential_path(np.arange(1, 8), ic_type=3)
        print("NOERROR")
    except capi.BessxError as e:
        print("ERR", e.code, round(time.time() - t0, 2), str(e))
"""


@pytest.mark.parametrize("mode", ["seq", "fit", "cv"])
def test_blocked_stream_returns_an_error_instead_of_hanging(gpu, mode):
    env = dict(os.environ, BESSX_WAIT_TIMEOUT_S="0.3")
    out = subprocess.run([sys.executable, "-c", CODE % ROOT, mode], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith(("ERR", "NOERROR"))][-1]
    assert line.startswith("ERR 2 "), line
    waited = float(line.split()[2])
    assert 0.25 <= waited < 2.0, line  # gave up at the deadline, long before the stream came back
    assert "BESSX_WAIT_TIMEOUT_S" in line


def test_deadline_is_not_met_by_ordinary_work(gpu):
    """The default deadline (30 s) is far from anything a path does; a generous explicit one changes nothing."""
    import numpy as np
    from bess_amd import synth
    X, y, _, _ = synth.make_lm(800, 120, 5)
    with gpu.Session(X, y) as s:
        s.debug_block_stream(300)  # shorter than the deadline: the call just takes that much longer
        out = s.sequential_path(np.arange(1, 8), ic_type=3)
    assert out["n_candidates"] == 7
