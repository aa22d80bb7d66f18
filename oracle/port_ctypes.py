"""ctypes loader for oracle/libbess_oracle.so (the plain-C restatement, bess_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BESS_ORACLE_LIB: another build of the same restatement (`make -C oracle asan`: address + undefined-behaviour sanitizers)
PORT_LIB = os.environ.get("BESS_ORACLE_LIB") or os.path.join(_HERE, "libbess_oracle.so")

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)
_i = ctypes.c_int
_d = ctypes.c_double

_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "port"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PORT_LIB):
            build()
        _lib = ctypes.CDLL(PORT_LIB)
        _lib.bess_oracle_run.restype = _i
        _lib.bess_oracle_run.argtypes = (
            [_D, _i, _i, _D, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _I, _I, _i, _D, _i, _i, _i, _I, _i]
            + [_D, _D, _D, _D])
        _lib.bess_oracle_run3.restype = _i
        _lib.bess_oracle_run3.argtypes = (
            [_D, _i, _i, _D, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _I, _I, _i, _D, _i, _i, _i, _d, _d, _i, _i,
             _I, _i, _I, _i] + [_D, _D, _D, _D, _D])
        _lib.bess_oracle_run2.restype = _i
        _lib.bess_oracle_run2.argtypes = (
            [_D, _i, _i, _D, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _I, _I, _i, _D, _i, _i, _i, _d, _d, _i, _i,
             _I, _i] + [_D, _D, _D, _D, _D])
        _lib.bess_oracle_trace_size.restype = _i
        _lib.bess_oracle_trace_size.argtypes = [_i]
        _lib.bess_oracle_trace_copy_int.restype = None
        _lib.bess_oracle_trace_copy_int.argtypes = [_i, _I]
        _lib.bess_oracle_trace_copy_double.restype = None
        _lib.bess_oracle_trace_copy_double.argtypes = [_i, _D]
        _lib.bess_oracle_screening.restype = _i
        _lib.bess_oracle_screening.argtypes = [_D, _i, _i, _D, _D, _i, _i, _I, _i, _I]
        _lib.bess_oracle_max_k.restype = None
        _lib.bess_oracle_max_k.argtypes = [_D, _i, _i, _I]
        _lib.bess_oracle_sym_solve.restype = _i
        _lib.bess_oracle_sym_solve.argtypes = [_D, _i, _D, _D]
    return _lib


def _dp(a):
    return a.ctypes.data_as(_D)


def _ip(a):
    return a.ctypes.data_as(_I)


def parse_trace(size_fn, copy_int, copy_double):
    """Shared by the oracle and the compiled-reference harness: flat trace -> list of fits."""
    def geti(which):
        n = size_fn(which)
        a = np.zeros(max(n, 1), dtype=np.int32)
        copy_int(which, _ip(a))
        return a[:n]

    def getd(which):
        n = size_fn(which)
        a = np.zeros(max(n, 1), dtype=np.float64)
        copy_double(which, _dp(a))
        return a[:n]

    meta = geti(0).reshape(-1, 4)
    a_flat = geti(1)
    beta_flat = getd(2)
    coef0_calls = getd(3)
    fits = []
    for c, (l, T0, train_n, off) in enumerate(meta):
        if l == 1:
            fits.append({"T0": int(T0), "train_n": int(train_n), "iters": [], "betas": [], "coef0s": []})
        nxt = meta[c + 1][3] if c + 1 < len(meta) else a_flat.size
        fits[-1]["iters"].append(a_flat[off:nxt].copy())
        fits[-1]["betas"].append(beta_flat[off:nxt].copy())
        fits[-1]["coef0s"].append(float(coef0_calls[c]))
    return fits, getd(4), getd(5)


def trace(x, y, weight=None, data_type=1, is_normal=True, algorithm_type=1, model_type=1, max_iter=20, path_type=1,
          is_warm_start=True, ic_type=4, is_cv=False, K=5, cv_fold_id=None, sequence=(1,), lambda_seq=(0.0,),
          s_min=1, s_max=1, g_index=None, always_select=(), lambda_min=0.0, lambda_max=0.0, nlambda=100,
          powell_path=1):
    """Same signature and return value as oracle.ref_ctypes.trace."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, p = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    weight = np.ones(n) if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    g_ptr, g_len = None, 0
    if g_index is not None:
        g_index = np.ascontiguousarray(g_index, dtype=np.int32)
        g_ptr, g_len = _ip(g_index), g_index.size
    sequence = np.ascontiguousarray(sequence, dtype=np.int32)
    lambda_seq = np.ascontiguousarray(lambda_seq, dtype=np.float64)
    always_select = np.ascontiguousarray(always_select, dtype=np.int32)
    fold_ptr = None
    if cv_fold_id is not None:
        cv_fold_id = np.ascontiguousarray(cv_fold_id, dtype=np.int32)
        fold_ptr = _ip(cv_fold_id)
    beta = np.zeros(p)
    coef0 = np.zeros(1)
    loss = np.zeros(1)
    ic = np.zeros(1)
    L = lib()
    lam_out = np.zeros(1)
    rc = L.bess_oracle_run3(_dp(x), n, p, _dp(y), _dp(weight), data_type, int(is_normal), algorithm_type, model_type,
                            max_iter, path_type, int(is_warm_start), ic_type, int(is_cv), K, fold_ptr, _ip(sequence),
                            sequence.size, _dp(lambda_seq), lambda_seq.size, s_min, s_max, lambda_min, lambda_max,
                            nlambda, powell_path, g_ptr, g_len, _ip(always_select), always_select.size, _dp(beta),
                            _dp(coef0), _dp(loss), _dp(ic), _dp(lam_out))
    if rc != 0:
        raise ValueError("bess_oracle_run rejected its arguments (code %d)" % rc)
    fits, loss_calls, ic_calls = parse_trace(L.bess_oracle_trace_size, L.bess_oracle_trace_copy_int,
                                             L.bess_oracle_trace_copy_double)
    return {"beta": beta, "coef0": float(coef0[0]), "train_loss": float(loss[0]), "ic": float(ic[0]),
            "lambda": float(lam_out[0]), "fits": fits, "loss_calls": loss_calls, "ic_calls": ic_calls}


def screening(x, y, weight, model_type, screening_size, always_select=()):
    """screening(), src/screening.cpp:26-105: indices of the kept columns (ascending)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, p = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    w = np.ones(n) if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    al = np.ascontiguousarray(always_select, dtype=np.int32)
    out = np.zeros(screening_size, dtype=np.int32)
    rc = lib().bess_oracle_screening(_dp(x), n, p, _dp(y), _dp(w), model_type, screening_size, _ip(al), al.size,
                                     _ip(out))
    if rc != 0:
        raise ValueError("bess_oracle_screening rejected its arguments")
    return out


def trace_screened(x, y, screening_size, **kw):
    """bessCpp with is_screening (src/bess.cpp:57-61, 186-209): screen, run the path on the kept columns, scatter
    the coefficients back.  always_select is re-indexed like src/screening.cpp:90-102."""
    A = screening(x, y, kw.get("weight"), kw.get("model_type", 1), screening_size, kw.get("always_select", ()))
    kw = dict(kw)
    if len(kw.get("always_select", ())):
        kw["always_select"] = [int(np.where(A == a)[0][0]) for a in kw["always_select"]]
    t = trace(np.ascontiguousarray(np.asarray(x)[:, A]), y, **kw)
    beta = np.zeros(np.asarray(x).shape[1])
    beta[A] = t["beta"]
    t["beta_screened"] = t["beta"]
    t["beta"] = beta
    t["screening_A"] = A
    return t


def last_timing():
    """(set-up seconds, path seconds) of the last trace() call: copy + normalise + group_XTX, and the path itself."""
    f = lib().bess_oracle_last_timing
    f.restype = None
    a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
    f(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def nth_heap_selects():
    """How often max_k has taken the heap-select branch of the restated std::nth_element in this process."""
    f = lib().bess_oracle_nth_heap_selects
    f.restype = ctypes.c_long
    return int(f())


def max_k(score, k):
    score = np.ascontiguousarray(score, dtype=np.float64)
    out = np.zeros(max(k, 1), dtype=np.int32)
    lib().bess_oracle_max_k(_dp(score), score.size, k, _ip(out))
    return out[:k]


def sym_solve(a, b):
    """a: k x k symmetric (lower triangle read), b: k."""
    a = np.asfortranarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    k = b.size
    x = np.zeros(max(k, 1))
    lib().bess_oracle_sym_solve(a.ctypes.data_as(_D), k, _dp(b), _dp(x))
    return x[:k]


def normalize(x, y, weight, data_type, is_normal=True, add_weight=False):
    """Data::normalize (+ add_weight) of the oracle on a copy: returns (x, y, x_mean, x_norm, y_mean)."""
    x = np.array(x, dtype=np.float64, order="F")
    n, p = x.shape
    y = np.array(y, dtype=np.float64)
    w = np.ascontiguousarray(weight, dtype=np.float64)
    xm, xn, ym = np.zeros(p), np.zeros(p), ctypes.c_double(0.0)
    L = lib()
    L.bess_oracle_normalize.argtypes = [_D, ctypes.c_int, ctypes.c_int, _D, _D, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        _D, _D, ctypes.POINTER(ctypes.c_double)]
    rc = L.bess_oracle_normalize(_dp(x), n, p, _dp(y), _dp(w), data_type, int(is_normal), int(add_weight), _dp(xm),
                                 _dp(xn), ctypes.byref(ym))
    if rc:
        raise RuntimeError("bess_oracle_normalize failed")
    return x, y, xm, xn, ym.value
