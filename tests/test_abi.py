"""CPU: the C-ABI library loads without a GPU, exports every symbol include/bessx.h declares, and refuses to
compute (BESSX_ERR_HIP) instead of falling back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

from bess_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "bessx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bessx_[A-Za-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    names = _declared()
    assert len(names) >= 20
    for nm in names:
        assert hasattr(lib, nm), "libbessx.so does not export " + nm
    assert sorted(capi.SYMBOLS) == names


def test_pybind_module_exposes_pywrap_bess():
    from bess_amd import _cbess
    assert callable(_cbess.pywrap_bess)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_gpu():
    x = np.random.default_rng(0).standard_normal((50, 5))
    y = x[:, 0] + 0.1
    with pytest.raises(capi.BessxError) as e:
        capi.Session(x, y)
    assert e.value.code == 2  # BESSX_ERR_HIP
    with pytest.raises(capi.BessxError):
        capi.op_topk(np.arange(10.0), 3)
    with pytest.raises(capi.BessxError):
        capi.pywrap_bess(x, y, 1, np.ones(50), True, 1, 1, 20, 0, 1, True, 4, False, 5, range(5), np.ones(50), [1, 2],
                         [0.0], 0, 0, 0, 1e-4, 0, 0, 100, False, 1, 1, [], 0.0, 5)


def test_argument_validation_needs_no_gpu():
    lib = capi.lib()
    assert lib.bessx_session_create(None, None) == 1  # BESSX_ERR_ARG
    assert b"null" in lib.bessx_last_error()
    h = ctypes.c_void_p()
    pb = capi.Problem()
    pb.n, pb.p = 0, 3
    assert lib.bessx_session_create(ctypes.byref(h), ctypes.byref(pb)) == 1
