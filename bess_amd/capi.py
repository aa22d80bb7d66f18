"""ctypes binding of libbessx.so (the C ABI declared in include/bessx.h).

This is plumbing only: every function forwards to the HIP library.  There is no Python or
NumPy fallback -- if the library is missing or no GPU is visible the call raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BESSX_LIB_PATH: development aid (an instrumented build of the same library, e.g. -DBESSX_KTRACE for tools/ktrace.py)
LIB_PATH = os.environ.get("BESSX_LIB_PATH") or os.path.join(_HERE, "libbessx.so")

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)
_i = ctypes.c_int
_d = ctypes.c_double
_ll = ctypes.c_longlong
_vp = ctypes.c_void_p

# names every build of libbessx.so must export (checked by tests/test_abi.py against include/bessx.h)
SYMBOLS = [
    "bessx_last_error", "bessx_device_info", "bessx_pywrap_bess", "bessx_bessCpp", "bessx_session_create",
    "bessx_session_destroy", "bessx_session_set_cv", "bessx_session_get_cv_folds", "bessx_session_sequential_path", "bessx_session_gs_path",
    "bessx_session_pgs_path", "bessx_session_get_screening", "bessx_session_get_screening_groups", "bessx_session_score_mode", "bessx_session_counter",
    "bessx_session_trace_enable", "bessx_session_trace_size", "bessx_session_trace_copy_int",
    "bessx_session_trace_copy_double", "bessx_session_get_normalization", "bessx_session_score_pass_stats",
    "bessx_session_enable_kernel_timing", "bessx_session_submodel_steps", "bessx_session_fit", "bessx_session_fit_width", "bessx_session_reset_caches",
    "bessx_session_sequential_path_chain", "bessx_session_cv_eval", "bessx_session_debug_block_stream",
    "bessx_session_set_fill_hook", "bessx_session_set_kpath_chains",
    "bessx_session_marginal_scores", "bessx_session_cov_prefill_begin", "bessx_session_cov_prefill_compute",
    "bessx_session_cov_prefill_export", "bessx_session_cov_prefill_import", "bessx_session_cov_prefill_end",
    "bessx_session_cov_prefill_extend", "bessx_session_cov_state", "bessx_op_xtv", "bessx_op_topk", "bessx_op_gram",
    "bessx_op_chol_solve", "bessx_op_topk_bench", "bessx_op_chol_bench", "bessx_op_normalize", "bessx_op_stream_copy_gbps", "bessx_op_xtv_bench", "bessx_op_cox_score_bench",
    "bessx_op_xtv_multi", "bessx_op_xtv_multi_bench",
    "bessx_comm_unique_id", "bessx_comm_init", "bessx_comm_rank", "bessx_comm_world", "bessx_comm_allgather_f64",
    "bessx_comm_destroy",
]


MAX_SPARSITY = 16382  # T0_HARD of bessx_host.h: the largest bessx_problem.max_sparsity


class BessxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbessx error %d: %s" % (code, msg))
        self.code = code


class Problem(ctypes.Structure):
    _fields_ = [("n", _i), ("p", _i), ("x", _D), ("x_col_major", _i), ("y", _D), ("weight", _D), ("data_type", _i),
                ("is_normal", _i), ("model_type", _i), ("algorithm_type", _i), ("max_iter", _i),
                ("is_warm_start", _i), ("always_select", _I), ("always_select_len", _i), ("device", _i),
                ("group_index", _I), ("group_index_len", _i),
                ("is_screening", _i), ("screening_size", _i), ("score_mode", _i), ("max_sparsity", _i)]


class RResult(ctypes.Structure):
    _fields_ = [("beta", _D), ("coef0", _d), ("train_loss", _d), ("ic", _d), ("lambda_", _d), ("all_capacity", _i),
                ("n_all", _i), ("beta_all", _D), ("coef0_all", _D), ("train_loss_all", _D), ("ic_all", _D),
                ("screening_A", _I)]


class PathResult(ctypes.Structure):
    _fields_ = [("beta", _D), ("coef0", _d), ("train_loss", _d), ("ic", _d), ("lambda_", _d), ("best_T0", _i),
                ("best_iters", _i), ("capacity", _i), ("n_candidates", _i), ("cand_T0", _I), ("cand_lambda", _D),
                ("cand_iters", _I), ("cand_train_loss", _D), ("cand_ic", _D), ("cand_coef0", _D),
                ("cand_support", _I), ("cand_beta", _D), ("max_T0", _i), ("device_seconds", _d), ("n_fits", _ll),
                ("n_pdas_iters", _ll)]


class PathChain(ctypes.Structure):
    _fields_ = [("init_idx", _I), ("init_val", _D), ("init_len", _i), ("init_coef0", _d), ("keep_caches", _i),
                ("stop_support", _I), ("stop_beta", _D), ("stop_rows", _i), ("stop_row_len", _i), ("stop_rtol", _d),
                ("stopped_at", _i), ("last_idx", _I), ("last_val", _D), ("last_cap", _i), ("last_len", _i),
                ("last_coef0", _d), ("lead_levels", _I), ("lead_len", _i)]


_lib = None


def lib():
    """Load libbessx.so; raises ImportError loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "bess_amd: %s is missing. Build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C bess_amd/csrc). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.bessx_last_error.restype = ctypes.c_char_p
        L.bessx_device_info.argtypes = [ctypes.c_char_p, _i]
        L.bessx_pywrap_bess.argtypes = (
            [_D, _i, _i, _D, _i, _i, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _I, _i, _D, _i, _I, _i, _D, _i]
            + [_i, _i, _i, _d, _d, _d, _i, _i, _i, _i, _I, _i, _d]
            + [_D, _i, _D, _i, _D, _i, _D, _i, _D, _D, _i, _D, _i, _D, _i, _I, _i, _I])
        L.bessx_bessCpp.argtypes = ([_D, _i, _i, _D, _i, _D, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _D, _i, _I, _i, _D,
                                     _i, _i, _i, _i, _d, _d, _d, _i, _i, _i, _i, _I, _i, _I, _i, _d,
                                     ctypes.POINTER(RResult)])
        L.bessx_session_create.argtypes = [ctypes.POINTER(_vp), ctypes.POINTER(Problem)]
        L.bessx_session_destroy.argtypes = [_vp]
        L.bessx_session_destroy.restype = None
        L.bessx_session_counter.argtypes = [_vp, _i]
        L.bessx_session_counter.restype = ctypes.c_longlong
        L.bessx_session_score_mode.argtypes = [_vp]
        L.bessx_session_score_mode.restype = _i
        L.bessx_session_get_screening.argtypes = [_vp, _I, _i]
        L.bessx_session_get_screening.restype = _i
        L.bessx_session_get_screening_groups.argtypes = [_vp, _I, _i]
        L.bessx_session_get_screening_groups.restype = _i
        L.bessx_session_set_cv.argtypes = [_vp, _i, _I, ctypes.c_uint]
        L.bessx_session_get_cv_folds.argtypes = [_vp, _I]
        L.bessx_session_sequential_path.argtypes = [_vp, _I, _i, _D, _i, _i, _i, ctypes.POINTER(PathResult)]
        L.bessx_session_gs_path.argtypes = [_vp, _i, _i, _i, _i, ctypes.POINTER(PathResult)]
        L.bessx_session_pgs_path.argtypes = [_vp, _i, _i, _d, _d, _i, _i, _i, _i, ctypes.POINTER(PathResult)]
        L.bessx_session_trace_enable.argtypes = [_vp, _i]
        L.bessx_session_trace_size.argtypes = [_vp, _i]
        L.bessx_session_trace_copy_int.argtypes = [_vp, _i, _I]
        L.bessx_session_trace_copy_double.argtypes = [_vp, _i, _D]
        L.bessx_session_get_normalization.argtypes = [_vp, _D, _D, _D]
        L.bessx_session_score_pass_stats.argtypes = [_vp, _i, _D, ctypes.POINTER(_ll), _D]
        L.bessx_session_enable_kernel_timing.argtypes = [_vp, _i]
        L.bessx_session_submodel_steps.argtypes = [_vp, _i, ctypes.POINTER(_ll)]
        L.bessx_session_fit.argtypes = [_vp, _i, _d, _i, _I, _D, _i, _d, _I, _D, _D, _I, _D, _D]
        L.bessx_session_reset_caches.argtypes = [_vp]
        L.bessx_session_sequential_path_chain.argtypes = [_vp, _I, _i, _D, _i, _i, _i, ctypes.POINTER(PathChain),
                                                          ctypes.POINTER(PathResult)]
        L.bessx_session_cv_eval.argtypes = [_vp, _i, _d, _i, _I, _D, _i, _d, _I, _i, _I, _D, _D, _I, _D, _D]
        L.bessx_session_debug_block_stream.argtypes = [_vp, _i]
        L.bessx_session_marginal_scores.argtypes = [_vp, _D]
        L.bessx_session_cov_prefill_begin.argtypes = [_vp, _I, _i]
        L.bessx_session_cov_prefill_extend.argtypes = [_vp, _I, _i]
        L.bessx_session_cov_state.argtypes = [_vp, _D, _I]
        L.bessx_session_cov_prefill_compute.argtypes = [_vp, _i, _i]
        L.bessx_session_cov_prefill_export.argtypes = [_vp, _i, _i, _vp, _i]
        L.bessx_session_cov_prefill_import.argtypes = [_vp, _i, _i, _vp, _i]
        L.bessx_session_cov_prefill_end.argtypes = [_vp]
        L.bessx_op_xtv.argtypes = [_D, _i, _i, _i, _D, _D, _D, _D]
        L.bessx_op_topk.argtypes = [_D, _i, _i, _I]
        L.bessx_op_gram.argtypes = [_D, _i, _i, _i, _I, _i, _D, _D]
        L.bessx_op_chol_solve.argtypes = [_D, _i, _D, _D]
        L.bessx_op_normalize.argtypes = [_D, _i, _i, _D, _D, _i, _i, _i, _D, _D, _D]
        L.bessx_op_stream_copy_gbps.argtypes = [_ll, _i, _D]
        L.bessx_op_xtv_bench.argtypes = [_i, _i, _i, _i, _D, _D]
        L.bessx_comm_unique_id.argtypes = [ctypes.c_char_p]
        L.bessx_comm_init.argtypes = [ctypes.POINTER(_vp), _i, _i, ctypes.c_char_p, _i]
        L.bessx_comm_rank.argtypes = [_vp]
        L.bessx_comm_world.argtypes = [_vp]
        L.bessx_comm_allgather_f64.argtypes = [_vp, _D, _i, _D]
        L.bessx_comm_destroy.argtypes = [_vp]
        L.bessx_comm_destroy.restype = None
        L.bessx_op_xtv_multi.argtypes = [_D, _i, _i, _i, _D, _D, _i, _D, _D]
        L.bessx_op_xtv_multi_bench.argtypes = [_i, _i, _i, _i, _i, _D, _D]
        L.bessx_op_topk_bench.argtypes = [_i, _i, _i, _i, _D]
        L.bessx_op_chol_bench.argtypes = [_i, _i, _D]
        _lib = L
    return _lib


def last_error():
    """Text of the calling thread's last library error (bessx_last_error)."""
    return lib().bessx_last_error().decode("utf-8", "replace")


def _check(rc):
    if rc != 0:
        raise BessxError(rc, lib().bessx_last_error().decode("utf-8", "replace"))


def _dp(a):
    return a.ctypes.data_as(_D) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_I) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def device_info():
    buf = ctypes.create_string_buffer(512)
    _check(lib().bessx_device_info(buf, 512))
    return buf.value.decode()


def pywrap_bess(x, y, data_type, weight, is_normal, algorithm_type, model_type, max_iter, exchange_num, path_type,
                is_warm_start, ic_type, is_cv, K, g_index, state, sequence, lambda_sequence, s_min, s_max, K_max,
                epsilon, lambda_min, lambda_max, n_lambda, is_screening, screening_size, powell_path, always_select,
                tao, beta_out_len, coef0_out_len=1, train_loss_out_len=1, ic_out_len=1, aic_out_len=1,
                bic_out_len=1, gic_out_len=1, A_out_len=None):
    """Same 38 positional arguments and 10-tuple result as the reference's SWIG wrapper
    (python/src/bess.i:17-30, called at python/bess/linear.py:360-375):
    (beta, coef0, train_loss, ic, nullloss, aic, bic, gic, A_out, l_out)."""
    x = _f64(x)
    n, p = x.shape
    y = _f64(y).reshape(-1)
    weight = _f64(weight)
    state = _f64(state)
    g_index = _i32(g_index)
    sequence = _i32(sequence)
    lambda_sequence = _f64(lambda_sequence)
    always_select = _i32(always_select)
    if A_out_len is None:
        A_out_len = p
    beta = np.zeros(beta_out_len)
    coef0, loss, ic, nullloss = np.zeros(coef0_out_len), np.zeros(train_loss_out_len), np.zeros(ic_out_len), np.zeros(1)
    aic, bic, gic = np.zeros(aic_out_len), np.zeros(bic_out_len), np.zeros(gic_out_len)
    a_out = np.zeros(A_out_len, dtype=np.int32)
    l_out = np.zeros(1, dtype=np.int32)
    _check(lib().bessx_pywrap_bess(
        _dp(x), n, p, _dp(y), y.size, data_type, _dp(weight), weight.size, int(is_normal), algorithm_type, model_type,
        max_iter, exchange_num, path_type, int(is_warm_start), ic_type, int(is_cv), K, _ip(g_index), g_index.size,
        _dp(state), state.size, _ip(sequence), sequence.size, _dp(lambda_sequence), lambda_sequence.size, s_min, s_max,
        K_max, epsilon, lambda_min, lambda_max, n_lambda, int(is_screening), screening_size, powell_path,
        _ip(always_select), always_select.size, tao, _dp(beta), beta.size, _dp(coef0), coef0.size, _dp(loss),
        loss.size, _dp(ic), ic.size, _dp(nullloss), _dp(aic), aic.size, _dp(bic), bic.size, _dp(gic), gic.size,
        _ip(a_out), a_out.size, _ip(l_out)))
    return beta, coef0, loss, ic, float(nullloss[0]), aic, bic, gic, a_out, int(l_out[0])


def bessCpp(x, y, data_type, weight, is_normal, algorithm_type, model_type, max_iter, exchange_num, path_type,
            is_warm_start, ic_type, is_cv, K, state, sequence, lambda_seq, s_min, s_max, K_max, epsilon, lambda_min,
            lambda_max, nlambda, is_screening, screening_size, powell_path, g_index, always_select, tao):
    """The R-facing entry (src/bess.h:20-33) through the C ABI (bessx_bessCpp): same 30 arguments, x handed over
    column-major like an R matrix; returns the named entries of the list the R build returns."""
    x = np.asfortranarray(x, dtype=np.float64)
    n, p = x.shape
    y, weight, state, lambda_seq = _f64(y).reshape(-1), _f64(weight), _f64(state), _f64(lambda_seq)
    sequence, g_index, always_select = _i32(sequence), _i32(g_index), _i32(always_select)
    cap = sequence.size * lambda_seq.size if path_type == 1 else 2 * (s_max - s_min + 1) + 128
    beta = np.zeros(p)
    beta_all = np.zeros((p, cap), order="F")
    coef0_all, loss_all, ic_all = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    scr = np.zeros(max(screening_size, 1), dtype=np.int32)
    r = RResult()
    r.beta, r.all_capacity = _dp(beta), cap
    r.beta_all, r.coef0_all, r.train_loss_all, r.ic_all, r.screening_A = (_dp(beta_all), _dp(coef0_all), _dp(loss_all),
                                                                         _dp(ic_all), _ip(scr))
    _check(lib().bessx_bessCpp(_dp(x), n, p, _dp(y), data_type, _dp(weight), int(is_normal), algorithm_type, model_type,
                               max_iter, exchange_num, path_type, int(is_warm_start), ic_type, int(is_cv), K, _dp(state),
                               state.size, _ip(sequence), sequence.size, _dp(lambda_seq), lambda_seq.size, s_min, s_max,
                               K_max, epsilon, lambda_min, lambda_max, nlambda, int(is_screening), screening_size,
                               powell_path, _ip(g_index), g_index.size, _ip(always_select), always_select.size, tao,
                               ctypes.byref(r)))
    k = min(r.n_all, cap)
    out = {"beta": beta, "coef0": r.coef0, "train_loss": r.train_loss, "ic": r.ic, "lambda": r.lambda_,
           "beta_all": np.array(beta_all[:, :k]), "coef0_all": coef0_all[:k], "train_loss_all": loss_all[:k],
           "ic_all": ic_all[:k]}
    if path_type == 1:  # list over lambda of p x len(sequence); ic_all as the len(sequence) x len(lambda) matrix
        ns, nl = sequence.size, lambda_seq.size
        out["beta_all"] = [np.array(beta_all[:, j * ns:(j + 1) * ns]) for j in range(nl)]
        out["coef0_all"] = [coef0_all[j * ns:(j + 1) * ns] for j in range(nl)]
        out["train_loss_all"] = [loss_all[j * ns:(j + 1) * ns] for j in range(nl)]
        out["ic_all"] = ic_all[:ns * nl].reshape(nl, ns).T
    if is_screening:
        out["screening_A"] = scr[:screening_size].copy()
    return out


class Session:
    """The state bessCpp builds (Data + Algorithm + Metric, src/bess.cpp:61-165), resident in HBM."""

    def __init__(self, x, y, weight=None, data_type=1, is_normal=True, model_type=1, algorithm_type=1, max_iter=20,
                 is_warm_start=True, always_select=(), x_col_major=False, device=-1, g_index=None,
                 is_screening=False, screening_size=0, score_mode=0, max_sparsity=0):
        x = np.asfortranarray(x, dtype=np.float64) if x_col_major else _f64(x)
        self.n, self.p = x.shape
        y = _f64(y).reshape(-1)
        if y.size != self.n:
            raise ValueError("X.shape(0) should be equal to y.size")
        w = None if weight is None else _f64(weight)
        al = _i32(always_select)
        gi = None if g_index is None else _i32(g_index)
        self._gsize_max = 1
        if gi is not None and gi.size:
            self._gsize_max = int(np.max(np.diff(np.append(gi, self.p))))
        pb = Problem(self.n, self.p, _dp(x), int(x_col_major), _dp(y), _dp(w), data_type, int(is_normal), model_type,
                     algorithm_type, max_iter, int(is_warm_start), _ip(al), al.size, device, _ip(gi),
                     0 if gi is None else gi.size, int(is_screening), int(screening_size), int(score_mode),
                     int(max_sparsity))
        h = _vp()
        _check(lib().bessx_session_create(ctypes.byref(h), ctypes.byref(pb)))
        self._h = h
        self.K = 0
        self.is_warm_start = bool(is_warm_start)
        self.p_kept = lib().bessx_session_get_screening(h, None, 0)  # columns the session works on

    def score_mode(self):
        """1 = streaming score pass, 2 = covariance updates (what bessx_problem.score_mode = 0 resolved to)."""
        return int(lib().bessx_session_score_mode(self._h))

    def counters(self):
        """Diagnostics of the covariance form (bessx_session_counter)."""
        names = {0: "chained_fits", 1: "cg_fallbacks", 2: "passes_over_X", 3: "chained_queued", 7: "cv_side_by_side_rounds",
                 8: "cv_union_fills", 9: "tie_rescues", 10: "cache_restarts", 11: "cv_contexts_dropped",
                 12: "cv_fold_contexts", 13: "shared_wide_fills", 14: "kpath_chunked_paths", 15: "kpath_stitch_refits",
                 16: "kpath_chunk_fills", 17: "kpath_chains_last_path",
                 18: "kpath_stitch_giveups", 19: "group_XTX_ns",
                 20: "kpath_merged_chunk_phases", 21: "kpath_chains_taken_over", 22: "kpath_coarse_us",
                 23: "kpath_chunks_us", 24: "kpath_stitch_us", 25: "panel_launches_one_group",
                 26: "panel_ns_one_group", 27: "panel_launches_two_groups", 28: "panel_ns_two_groups",
                 29: "shared_pass_launches", 30: "shared_pass_chain_slots", 31: "shared_pass_partial_batches",
                 32: "own_queue_streams_created_by_the_process"}  # (4-6: mechanisms removed in round 3)
        return {n: int(lib().bessx_session_counter(self._h, i)) for i, n in names.items()}

    def screening(self):
        """screening_A: original column of every kept column."""
        a = np.zeros(self.p_kept, dtype=np.int32)
        lib().bessx_session_get_screening(self._h, _ip(a), a.size)
        return a

    def screening_groups(self):
        """Kept original group numbers after screening with groups of size > 1 (empty otherwise)."""
        n = lib().bessx_session_get_screening_groups(self._h, None, 0)
        a = np.zeros(max(n, 1), dtype=np.int32)
        lib().bessx_session_get_screening_groups(self._h, _ip(a), n)
        return a[:n]

    def close(self):
        if getattr(self, "_h", None):
            lib().bessx_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_cv(self, K, fold_id=None, seed=123):
        f = None if fold_id is None else _i32(fold_id)
        _check(lib().bessx_session_set_cv(self._h, K, _ip(f), seed))
        self.K = K

    def cv_folds(self):
        """Test fold of every row as set_cv fixed it (given, or drawn from the seed)."""
        f = np.zeros(self.n, dtype=np.int32)
        _check(lib().bessx_session_get_cv_folds(self._h, _ip(f)))
        return f

    def trace_enable(self, on=True):
        _check(lib().bessx_session_trace_enable(self._h, int(on)))

    def enable_kernel_timing(self, on=True):
        _check(lib().bessx_session_enable_kernel_timing(self._h, int(on)))

    def score_pass_stats(self, reset=False):
        sec, nb, cnt = _d(0), _d(0), _ll(0)
        _check(lib().bessx_session_score_pass_stats(self._h, int(reset), ctypes.byref(sec), ctypes.byref(cnt),
                                                    ctypes.byref(nb)))
        return {"seconds": sec.value, "launches": cnt.value, "algorithmic_bytes": nb.value}

    def submodel_steps(self, reset=False):
        """IRLS / Newton steps of the restricted fits since the last reset (0 for LM)."""
        cnt = _ll(0)
        _check(lib().bessx_session_submodel_steps(self._h, int(reset), ctypes.byref(cnt)))
        return cnt.value

    def normalization(self):
        xm, xn, ym = np.zeros(self.p_kept), np.zeros(self.p_kept), _d(0)
        _check(lib().bessx_session_get_normalization(self._h, _dp(xm), _dp(xn), ctypes.byref(ym)))
        return xm, xn, ym.value

    def _run(self, call, capacity, max_T0):
        beta = np.zeros(self.p)
        arr = {
            "cand_T0": np.zeros(capacity, dtype=np.int32), "cand_lambda": np.zeros(capacity),
            "cand_iters": np.zeros(capacity, dtype=np.int32), "cand_train_loss": np.zeros(capacity),
            "cand_ic": np.zeros(capacity), "cand_coef0": np.zeros(capacity),
            "cand_support": np.full((capacity, max_T0), -1, dtype=np.int32), "cand_beta": np.zeros((capacity, max_T0)),
        }
        res = PathResult()
        res.beta = _dp(beta)
        res.capacity = capacity
        res.max_T0 = max_T0
        for k, v in arr.items():
            setattr(res, k, _ip(v) if v.dtype == np.int32 else _dp(v))
        _check(call(ctypes.byref(res)))
        nc = min(res.n_candidates, capacity)
        out = {"beta": beta, "coef0": res.coef0, "train_loss": res.train_loss, "ic": res.ic, "lambda": res.lambda_,
               "best_T0": res.best_T0, "best_iters": res.best_iters, "n_candidates": res.n_candidates,
               "device_seconds": res.device_seconds, "n_fits": res.n_fits, "n_pdas_iters": res.n_pdas_iters}
        for k, v in arr.items():
            out[k] = v[:nc]
        out["trace"] = self._trace()
        return out

    def sequential_path(self, sequence, lambda_seq=(0.0,), ic_type=4, is_cv=False):
        seq = _i32(sequence)
        lam = _f64(lambda_seq)
        L = lib()
        return self._run(lambda r: L.bessx_session_sequential_path(self._h, _ip(seq), seq.size, _dp(lam), lam.size,
                                                                   ic_type, int(is_cv), r),
                         seq.size * lam.size, min(self.p, (int(seq.max()) if seq.size else 1) * self._gsize_max))

    def sequential_path_chain(self, sequence, lambda_seq=(0.0,), ic_type=4, is_cv=False, init_idx=(), init_val=(),
                              init_coef0=0.0, keep_caches=False, stop_support=None, stop_beta=None, stop_rtol=1e-9,
                              lead_levels=()):
        """sequential_path as one link of a longer warm-start chain (bessx_session_sequential_path_chain): starts from
        the given (normalised) model, optionally on the caches of the previous call, and stops after the first
        candidate that equals the caller's own row of stop_support / stop_beta.  Adds to the path result: stopped_at,
        last_idx / last_val / last_coef0 (the model the next candidate of the chain starts from)."""
        seq, lam = _i32(sequence), _f64(lambda_seq)
        ii, iv = _i32(init_idx), _f64(init_val)
        max_T0 = min(self.p, (int(seq.max()) if seq.size else 1) * self._gsize_max)
        ch = PathChain()
        ch.init_idx, ch.init_val, ch.init_len, ch.init_coef0 = _ip(ii), _dp(iv), ii.size, float(init_coef0)
        ch.keep_caches = int(bool(keep_caches))
        lead = _i32(lead_levels)  # (bessx_path_chain.lead_levels: a coarse warm-start chain in front of the link)
        ch.lead_levels, ch.lead_len = (_ip(lead), lead.size) if lead.size else (None, 0)
        ss = sb = None
        if stop_support is not None:
            ss = np.full((len(stop_support), max_T0), -1, dtype=np.int32)
            sb = np.zeros((len(stop_support), max_T0))
            for r, row in enumerate(stop_support):
                row = np.asarray(row)
                row = row[row >= 0]
                ss[r, :row.size] = row
                if stop_beta is not None:
                    sb[r, :row.size] = np.asarray(stop_beta[r])[:row.size]
            ch.stop_support, ch.stop_rows, ch.stop_row_len, ch.stop_rtol = _ip(ss), ss.shape[0], max_T0, float(stop_rtol)
            ch.stop_beta = _dp(sb) if stop_beta is not None else None
        li, lv = np.zeros(max_T0, dtype=np.int32), np.zeros(max_T0)
        ch.last_idx, ch.last_val, ch.last_cap = _ip(li), _dp(lv), max_T0
        L = lib()
        out = self._run(lambda r: L.bessx_session_sequential_path_chain(self._h, _ip(seq), seq.size, _dp(lam), lam.size,
                                                                        ic_type, int(is_cv), ctypes.byref(ch), r),
                        seq.size * lam.size, max_T0)
        n_last = min(ch.last_len, max_T0)
        out.update({"stopped_at": int(ch.stopped_at), "last_idx": li[:n_last].copy(), "last_val": lv[:n_last].copy(),
                    "last_coef0": float(ch.last_coef0)})
        return out

    def cv_eval(self, T0, lam=0.0, want_full=True, init_idx=(), init_val=(), init_coef0=0.0, folds=()):
        """One cross-validated candidate restricted to `folds` (ascending), the full-data fit in front when want_full
        (bessx_session_cv_eval).  Returns the list of fit records [full,] fold..., each like fit()'s."""
        ii, iv, fo = _i32(init_idx), _f64(init_val), _i32(folds)
        w = self.fit_width(T0)
        nrec = int(bool(want_full)) + fo.size
        sup, b = np.zeros((max(nrec, 1), w), dtype=np.int32), np.zeros((max(nrec, 1), w))
        c0, tr, te = np.zeros(max(nrec, 1)), np.zeros(max(nrec, 1)), np.zeros(max(nrec, 1))
        it = np.zeros(max(nrec, 1), dtype=np.int32)
        _check(lib().bessx_session_cv_eval(self._h, int(T0), float(lam), int(bool(want_full)), _ip(ii), _dp(iv), ii.size,
                                           float(init_coef0), _ip(fo), fo.size, _ip(sup), _dp(b), _dp(c0), _ip(it),
                                           _dp(tr), _dp(te)))
        recs = []
        for r in range(nrec):
            keep = sup[r] >= 0
            recs.append({"support": sup[r][keep], "beta": b[r][keep], "coef0": float(c0[r]), "iters": int(it[r]),
                         "train_loss": float(tr[r]), "test_loss": float(te[r])})
        return recs

    # ---- cooperative prefill of the Gram column cache (bessx_session_cov_prefill_*) ----
    def marginal_scores(self):
        bd = np.zeros(self.p_kept)
        _check(lib().bessx_session_marginal_scores(self._h, _dp(bd)))
        return bd

    def cov_prefill_begin(self, cols):
        c = _i32(cols)
        _check(lib().bessx_session_cov_prefill_begin(self._h, _ip(c), c.size))

    def cov_prefill_extend(self, cols):
        c = _i32(cols)
        _check(lib().bessx_session_cov_prefill_extend(self._h, _ip(c), c.size))

    def cov_state(self):
        """(scores of the last fit's last PDAS iteration, cache slot of every column or -1)."""
        bd, so = np.zeros(self.p_kept), np.zeros(self.p_kept, dtype=np.int32)
        _check(lib().bessx_session_cov_state(self._h, _dp(bd), _ip(so)))
        return bd, so

    def cov_prefill_compute(self, g0, ngroups):
        _check(lib().bessx_session_cov_prefill_compute(self._h, int(g0), int(ngroups)))

    def cov_prefill_export(self, g0, ngroups, device_ptr=None):
        """The p x 32 blocks of groups g0 .. g0+ngroups-1: into device memory at device_ptr, or returned as an array."""
        if device_ptr is not None:
            _check(lib().bessx_session_cov_prefill_export(self._h, int(g0), int(ngroups), _vp(int(device_ptr)), 1))
            return None
        out = np.empty(int(ngroups) * 32 * self.p_kept)
        _check(lib().bessx_session_cov_prefill_export(self._h, int(g0), int(ngroups), out.ctypes.data_as(_vp), 0))
        return out

    def cov_prefill_import(self, g0, ngroups, blocks=None, device_ptr=None):
        if device_ptr is not None:
            _check(lib().bessx_session_cov_prefill_import(self._h, int(g0), int(ngroups), _vp(int(device_ptr)), 1))
            return
        b = _f64(blocks)
        assert b.size == int(ngroups) * 32 * self.p_kept
        _check(lib().bessx_session_cov_prefill_import(self._h, int(g0), int(ngroups), b.ctypes.data_as(_vp), 0))

    def cov_prefill_end(self):
        _check(lib().bessx_session_cov_prefill_end(self._h))

    def set_kpath_chains(self, chains):
        """Chunk chains of sequential_path (bessx_session_set_kpath_chains): 0 automatic, 1 one chain, 2..8."""
        _check(lib().bessx_session_set_kpath_chains(self._h, int(chains)))

    _FILL_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)

    def set_fill_hook(self, fn, width=0):
        """Shared wide fills inside a fit (bessx_session_set_fill_hook): while set, a parked fit of the all-rows row set
        lists `width` columns -- the missing ones, then the best uncached ones by the current scores -- and calls
        fn(n_groups), which forms its share, exchanges the blocks and closes the list (cov_prefill_compute / export /
        import / end).  fn = None: private fills again.  An exception in fn fails the fit and is re-raised by it."""
        self._hook_error = None
        if fn is None:
            _check(lib().bessx_session_set_fill_hook(self._h, ctypes.cast(None, self._FILL_HOOK), None, 0))
            self._hook_ref = None
            return

        def tramp(_user, ng):
            try:
                fn(int(ng))
                return 0
            except BaseException as e:  # (an exception must not cross the C frames)
                self._hook_error = e
                return 1

        ref = self._FILL_HOOK(tramp)
        _check(lib().bessx_session_set_fill_hook(self._h, ref, None, int(width)))
        self._hook_ref = ref  # (kept alive as long as the library may call it)

    def debug_block_stream(self, milliseconds):
        """Test hook: everything queued on the session's stream waits behind a host function that sleeps."""
        _check(lib().bessx_session_debug_block_stream(self._h, int(milliseconds)))

    def gs_path(self, s_min, s_max, ic_type=4, is_cv=False):
        L = lib()
        return self._run(lambda r: L.bessx_session_gs_path(self._h, s_min, s_max, ic_type, int(is_cv), r),
                         2 * (s_max - s_min + 1) + 64, min(self.p, max(s_max, 1) * self._gsize_max))

    def pgs_path(self, s_min, s_max, lambda_min, lambda_max, n_lambda=100, powell_path=1, ic_type=4, is_cv=False):
        L = lib()
        return self._run(lambda r: L.bessx_session_pgs_path(self._h, s_min, s_max, lambda_min, lambda_max, n_lambda,
                                                            powell_path, ic_type, int(is_cv), r), 128,
                         min(self.p, max(s_max, 1) * self._gsize_max))

    def fit(self, T0, lam=0.0, fold=-1, init_idx=(), init_val=(), init_coef0=0.0):
        ii, iv = _i32(init_idx), _f64(init_val)
        w = self.fit_width(T0)  # T0 columns, or the columns of the T0 widest groups (groups of size > 1)
        sup, b = np.zeros(w, dtype=np.int32), np.zeros(w)
        c0, tr, te, it = _d(0), _d(0), _d(0), _i(0)
        _check(lib().bessx_session_fit(self._h, T0, lam, fold, _ip(ii), _dp(iv), ii.size, init_coef0, _ip(sup),
                                       _dp(b), ctypes.byref(c0), ctypes.byref(it), ctypes.byref(tr),
                                       ctypes.byref(te)))
        keep = sup >= 0  # (with groups the selected columns may be fewer than the width)
        return {"support": sup[keep], "beta": b[keep], "coef0": c0.value, "iters": it.value, "train_loss": tr.value,
                "test_loss": te.value}

    def fit_width(self, T0):
        """Most columns a fit of sparsity level T0 returns (bessx_session_fit_width)."""
        w = int(lib().bessx_session_fit_width(self._h, int(T0)))
        if w < 0:
            raise BessxError(1, "sparsity level outside [1, number of groups]")
        return w

    def reset_caches(self):
        """Start cold, like a path call does (bessx_session_reset_caches)."""
        _check(lib().bessx_session_reset_caches(self._h))

    def _trace(self):
        L = lib()

        def geti(which):
            n = L.bessx_session_trace_size(self._h, which)
            a = np.zeros(max(n, 1), dtype=np.int32)
            _check(L.bessx_session_trace_copy_int(self._h, which, _ip(a)))
            return a[:n]

        def getd(which):
            n = L.bessx_session_trace_size(self._h, which)
            a = np.zeros(max(n, 1))
            _check(L.bessx_session_trace_copy_double(self._h, which, _dp(a)))
            return a[:n]

        meta = geti(0).reshape(-1, 4)
        if meta.shape[0] == 0:
            return None
        a_flat, beta_flat, coef0_calls = geti(1), getd(2), getd(3)
        fits = []
        for c, (l, T0, train_n, off) in enumerate(meta):
            if l == 1:
                fits.append({"T0": int(T0), "train_n": int(train_n), "iters": [], "betas": [], "coef0s": []})
            nxt = meta[c + 1][3] if c + 1 < len(meta) else a_flat.size
            fits[-1]["iters"].append(a_flat[off:nxt].copy())
            fits[-1]["betas"].append(beta_flat[off:nxt].copy())
            fits[-1]["coef0s"].append(float(coef0_calls[c]))
        return {"fits": fits, "loss_calls": getd(4), "ic_calls": getd(5)}


# ---- single-kernel entry points (parity tests) -------------------------------------------------
def op_xtv(x, v, v2=None):
    x = np.asfortranarray(x, dtype=np.float64)
    n, p = x.shape
    v = _f64(v)
    out, out2 = np.zeros(p), np.zeros(p)
    v2a = None if v2 is None else _f64(v2)
    _check(lib().bessx_op_xtv(_dp(x), n, p, n, _dp(v), _dp(v2a), _dp(out), _dp(out2)))
    return (out, out2) if v2 is not None else out


def op_xtv_multi(x, vs, v2s=None):
    """X^T v for the rows of vs (nc x n) in one pass over X; with v2s also X^2^T v2."""
    x = np.asfortranarray(x, dtype=np.float64)
    n, p = x.shape
    vs = np.ascontiguousarray(vs, dtype=np.float64)
    nc = vs.shape[0]
    out, out2 = np.zeros((nc, p)), np.zeros((nc, p))
    v2a = None if v2s is None else np.ascontiguousarray(v2s, dtype=np.float64)
    _check(lib().bessx_op_xtv_multi(_dp(x), n, p, n, _dp(vs), _dp(v2a), nc, _dp(out), _dp(out2)))
    return (out, out2) if v2s is not None else out


def op_topk(score, k):
    score = _f64(score)
    out = np.zeros(max(k, 1), dtype=np.int32)
    _check(lib().bessx_op_topk(_dp(score), score.size, k, _ip(out)))
    return out[:k]


def op_gram(x, cols, w=None):
    x = np.asfortranarray(x, dtype=np.float64)
    n, p = x.shape
    cols = _i32(cols)
    wa = None if w is None else _f64(w)
    out = np.zeros((cols.size, cols.size), order="F")
    _check(lib().bessx_op_gram(_dp(x), n, p, n, _ip(cols), cols.size, _dp(wa), _dp(out)))
    return np.array(out)


def op_chol_solve(a, b):
    a = np.asfortranarray(a, dtype=np.float64)
    b = _f64(b)
    sol = np.zeros(b.size)
    _check(lib().bessx_op_chol_solve(_dp(a), b.size, _dp(b), _dp(sol)))
    return sol


def op_normalize(x, y, weight, data_type, is_normal=True, add_weight=False):
    x = np.array(x, dtype=np.float64, order="F")
    n, p = x.shape
    y = np.array(y, dtype=np.float64)
    w = _f64(weight)
    xm, xn, ym = np.zeros(p), np.zeros(p), _d(0)
    _check(lib().bessx_op_normalize(_dp(x), n, p, _dp(y), _dp(w), data_type, int(is_normal), int(add_weight),
                                    _dp(xm), _dp(xn), ctypes.byref(ym)))
    return x, y, xm, xn, ym.value


def op_stream_copy_gbps(nbytes=1 << 30, repeats=10):
    g = _d(0)
    _check(lib().bessx_op_stream_copy_gbps(nbytes, repeats, ctypes.byref(g)))
    return g.value


def op_chol_bench(m, repeats=200):
    us = _d(0)
    _check(lib().bessx_op_chol_bench(m, repeats, ctypes.byref(us)))
    return us.value


def op_topk_bench(length, k, variant=1, repeats=200):
    us = _d(0)
    _check(lib().bessx_op_topk_bench(length, k, variant, repeats, ctypes.byref(us)))
    return us.value


def op_xtv_multi_bench(n, p, nc, two=False, repeats=20):
    g, ms = _d(0), _d(0)
    _check(lib().bessx_op_xtv_multi_bench(n, p, nc, int(two), repeats, ctypes.byref(g), ctypes.byref(ms)))
    return g.value, ms.value


def op_xtv_bench(n, p, variant, repeats=20):
    g, ms = _d(0), _d(0)
    _check(lib().bessx_op_xtv_bench(n, p, variant, repeats, ctypes.byref(g), ctypes.byref(ms)))
    return g.value, ms.value
