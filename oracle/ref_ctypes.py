"""ctypes loader for oracle/_ref/libbess_ref.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The library is the reference's own C++ (compiled by oracle/Makefile from the sources
under /root/reference/src) plus oracle/ref_harness.cpp.  Only tests/, the golden
generator, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BESS_REF_LIB selects another build of the same sources (oracle/Makefile: ref_fast), for CPU timing only
REF_LIB = os.environ.get("BESS_REF_LIB") or os.path.join(_HERE, "_ref", "libbess_ref.so")

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)
_i = ctypes.c_int
_d = ctypes.c_double


def available():
    return os.path.exists(REF_LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(REF_LIB)
        _lib.bess_ref_pywrap.restype = None
        _lib.bess_ref_pywrap.argtypes = (
            [_D, _i, _i, _D, _i, _i, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _I, _i, _D, _i, _I, _i, _D, _i]
            + [_i, _i, _i, _d, _d, _d, _i, _i, _i, _i, _I, _i, _d]
            + [_D, _i, _D, _i, _D, _i, _D, _i, _D, _D, _i, _D, _i, _D, _i, _I, _i, _I]
        )
        _lib.bess_ref_trace.restype = _i
        _lib.bess_ref_trace.argtypes = (
            [_D, _i, _i, _D, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _I, _I, _i, _D, _i, _i, _i, _I, _i, _I, _i]
            + [_D, _D, _D, _D]
        )
        _lib.bess_ref_trace2.restype = _i
        _lib.bess_ref_trace2.argtypes = (
            [_D, _i, _i, _D, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _I, _I, _i, _D, _i, _i, _i, _d, _d, _i, _i,
             _I, _i, _I, _i] + [_D, _D, _D, _D, _D])
        _lib.bess_ref_time_chain.restype = _i
        _lib.bess_ref_time_chain.argtypes = [_D, _i, _i, _D, _D, _i, _i, _i, _i, _i, _i, _I, _i, _I, _D, _i, _d, _d,
                                             _D, _I, _I, _D]
        _lib.bess_ref_screening.restype = _i
        _lib.bess_ref_screening.argtypes = [_D, _i, _i, _D, _D, _i, _i, _I, _i, _I]
        _lib.bess_ref_screening_groups.restype = _i
        _lib.bess_ref_screening_groups.argtypes = [_D, _i, _i, _D, _D, _i, _i, _I, _i, _I, _i, _I]
        _lib.bess_ref_trace_size.restype = _i
        _lib.bess_ref_max_k.restype = None
        _lib.bess_ref_max_k.argtypes = [_D, _i, _i, _I]
        _lib.bess_ref_trace_size.argtypes = [_i]
        _lib.bess_ref_trace_copy_int.restype = None
        _lib.bess_ref_trace_copy_int.argtypes = [_i, _I]
        _lib.bess_ref_trace_copy_double.restype = None
        _lib.bess_ref_trace_copy_double.argtypes = [_i, _D]
    return _lib


def _dp(a):
    return a.ctypes.data_as(_D)


def _ip(a):
    return a.ctypes.data_as(_I)


def pywrap_bess(x, y, data_type, weight, is_normal, algorithm_type, model_type, max_iter, exchange_num, path_type,
                is_warm_start, ic_type, is_cv, K, g_index, state, sequence, lambda_sequence, s_min, s_max, K_max,
                epsilon, lambda_min, lambda_max, n_lambda, is_screening, screening_size, powell_path, always_select,
                tao):
    """The reference's pywrap_bess (src/bess.cpp:218-281); returns (beta, coef0, train_loss, ic)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, p = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    weight = np.ascontiguousarray(weight, dtype=np.float64)
    state = np.ascontiguousarray(state, dtype=np.float64)
    g_index = np.ascontiguousarray(g_index, dtype=np.int32)
    sequence = np.ascontiguousarray(sequence, dtype=np.int32)
    lambda_sequence = np.ascontiguousarray(lambda_sequence, dtype=np.float64)
    always_select = np.ascontiguousarray(always_select, dtype=np.int32)
    beta = np.zeros(p)
    one = [np.zeros(1) for _ in range(7)]
    a_out = np.zeros(p, dtype=np.int32)
    l_out = np.zeros(1, dtype=np.int32)
    lib().bess_ref_pywrap(
        _dp(x), n, p, _dp(y), y.size, data_type, _dp(weight), weight.size, int(is_normal), algorithm_type, model_type,
        max_iter, exchange_num, path_type, int(is_warm_start), ic_type, int(is_cv), K, _ip(g_index), g_index.size,
        _dp(state), state.size, _ip(sequence), sequence.size, _dp(lambda_sequence), lambda_sequence.size, s_min, s_max,
        K_max, epsilon, lambda_min, lambda_max, n_lambda, int(is_screening), screening_size, powell_path,
        _ip(always_select), always_select.size, tao, _dp(beta), p, _dp(one[0]), 1, _dp(one[1]), 1, _dp(one[2]), 1,
        _dp(one[3]), _dp(one[4]), 1, _dp(one[5]), 1, _dp(one[6]), 1, _ip(a_out), p, _ip(l_out))
    return beta, float(one[0][0]), float(one[1][0]), float(one[2][0])


def trace(x, y, weight=None, data_type=1, is_normal=True, algorithm_type=1, model_type=1, max_iter=20, path_type=1,
          is_warm_start=True, ic_type=4, is_cv=False, K=5, cv_fold_id=None, sequence=(1,), lambda_seq=(0.0,),
          s_min=1, s_max=1, g_index=None, always_select=(), lambda_min=0.0, lambda_max=0.0, nlambda=100,
          powell_path=1):
    """Run one reference path through the tracing harness (path_type 3 = Powell path pgs_path).

    Returns a dict: best model (beta, coef0, train_loss, ic) plus the trace
      fits: list of dicts {T0, train_n, iters: [A arrays], betas: [beta_A arrays], coef0s: [...]}
      loss_calls / ic_calls: top-level metric values in call order.
    """
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, p = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    weight = np.ones(n) if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    g_index = np.arange(p, dtype=np.int32) if g_index is None else np.ascontiguousarray(g_index, dtype=np.int32)
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
    rc = L.bess_ref_trace2(_dp(x), n, p, _dp(y), _dp(weight), data_type, int(is_normal), algorithm_type, model_type,
                           max_iter, path_type, int(is_warm_start), ic_type, int(is_cv), K, fold_ptr, _ip(sequence),
                           sequence.size, _dp(lambda_seq), lambda_seq.size, s_min, s_max, lambda_min, lambda_max,
                           nlambda, powell_path, _ip(g_index), g_index.size, _ip(always_select), always_select.size,
                           _dp(beta), _dp(coef0), _dp(loss), _dp(ic), _dp(lam_out))
    if rc not in (0, 2):
        raise RuntimeError("bess_ref_trace failed")
    truncated = rc == 2  # BESS_REF_BUDGET_S ran out: whole fits so far, no best model (oracle/ref_harness.cpp)

    def geti(which):
        a = np.zeros(max(L.bess_ref_trace_size(which), 1), dtype=np.int32)
        L.bess_ref_trace_copy_int(which, _ip(a))
        return a[:L.bess_ref_trace_size(which)]

    def getd(which):
        a = np.zeros(max(L.bess_ref_trace_size(which), 1), dtype=np.float64)
        L.bess_ref_trace_copy_double(which, _dp(a))
        return a[:L.bess_ref_trace_size(which)]

    meta = geti(0).reshape(-1, 4)
    a_flat = geti(1)
    beta_flat = getd(2)
    coef0_calls = getd(3)
    fits = []
    for c, (l, T0, train_n, off) in enumerate(meta):
        if l == 1:
            fits.append({"T0": int(T0), "train_n": int(train_n), "iters": [], "betas": [], "coef0s": []})
        # the active set has T0 entries for singleton groups (all BASELINE configs)
        nxt = meta[c + 1][3] if c + 1 < len(meta) else a_flat.size
        fits[-1]["iters"].append(a_flat[off:nxt].copy())
        fits[-1]["betas"].append(beta_flat[off:nxt].copy())
        fits[-1]["coef0s"].append(float(coef0_calls[c]))
    return {"beta": beta, "coef0": float(coef0[0]), "train_loss": float(loss[0]), "ic": float(ic[0]),
            "lambda": float(lam_out[0]), "fits": fits, "loss_calls": getd(4), "ic_calls": getd(5),
            "truncated": truncated}


def max_k(score, k):
    """The reference's max_k (src/utilities.cpp:179-188) on a score vector: the k selected indices, ascending."""
    score = np.ascontiguousarray(score, dtype=np.float64)
    out = np.zeros(max(k, 1), dtype=np.int32)
    lib().bess_ref_max_k(_dp(score), score.size, k, _ip(out))
    return out[:k]


def time_chain(x, y, sequence, init_idx=(), init_val=(), init_coef0=0.0, budget_s=15.0, weight=None, data_type=1,
               is_normal=True, algorithm_type=1, model_type=1, max_iter=20, ic_type=3):
    """Wall time per candidate of one warm-start chain of the reference's own Algorithm / Metric objects
    (oracle/ref_harness.cpp: bess_ref_time_chain), optionally started from a given model (normalised scale) so that
    candidates from the far end of a path can be timed alone.  Returns {"seconds": [...], "iters": [...],
    "setup_seconds": s}; stops after the first candidate that ends beyond budget_s."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, p = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    w = np.ones(n) if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    seq = np.ascontiguousarray(sequence, dtype=np.int32)
    ii = np.ascontiguousarray(init_idx, dtype=np.int32)
    iv = np.ascontiguousarray(init_val, dtype=np.float64)
    sec = np.zeros(max(seq.size, 1))
    its = np.zeros(max(seq.size, 1), dtype=np.int32)
    done = ctypes.c_int(0)
    setup = ctypes.c_double(0.0)
    rc = lib().bess_ref_time_chain(_dp(x), n, p, _dp(y), _dp(w), data_type, int(is_normal), algorithm_type, model_type,
                                   max_iter, ic_type, _ip(seq), seq.size, _ip(ii), _dp(iv), ii.size, init_coef0,
                                   budget_s, _dp(sec), _ip(its), ctypes.byref(done), ctypes.byref(setup))
    if rc != 0:
        raise RuntimeError("bess_ref_time_chain failed")
    return {"seconds": sec[:done.value].copy(), "iters": its[:done.value].copy(), "setup_seconds": setup.value}


def screening(x, y, weight, model_type, screening_size, always_select=()):
    """The reference's screening() (src/screening.cpp:26-105): kept column indices."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, p = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    w = np.ones(n) if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    al = np.ascontiguousarray(always_select, dtype=np.int32)
    out = np.zeros(screening_size, dtype=np.int32)
    lib().bess_ref_screening(_dp(x), n, p, _dp(y), _dp(w), model_type, screening_size, _ip(al), al.size, _ip(out))
    return out


def screening_groups(x, y, weight, model_type, screening_size, g_index, always_select=()):
    """The reference's screening() with groups of size > 1: kept GROUP numbers (ascending)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, p = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    w = np.ones(n) if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    gi = np.ascontiguousarray(g_index, dtype=np.int32)
    al = np.ascontiguousarray(always_select, dtype=np.int32)
    out = np.zeros(screening_size, dtype=np.int32)
    lib().bess_ref_screening_groups(_dp(x), n, p, _dp(y), _dp(w), model_type, screening_size, _ip(gi), gi.size, _ip(al),
                                    al.size, _ip(out))
    return out
