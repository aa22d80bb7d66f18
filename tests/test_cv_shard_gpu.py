"""Fold-sharded cross-validated paths (bess_amd.dist.FoldShardedCV, SURVEY 8e / BASELINE configs[3]) on the GPU:
built on bessx_session_fit, they must reproduce the library's own single-process gs_path / sequential_path under
CV exactly (same chains, same order of operations), and therefore the oracle's."""
import numpy as np
import pytest

from bess_amd import synth
from bess_amd import dist as bdist
from oracle import port_ctypes as P

pytestmark = pytest.mark.gpu


def _both(capi, X, y, fold, path, **skw):
    with capi.Session(X, y, **skw) as s:
        s.set_cv(5, fold)
        ref = s.gs_path(1, 30, ic_type=3, is_cv=True) if path == "gs" else \
            s.sequential_path(np.arange(1, 16), ic_type=3, is_cv=True)
    with capi.Session(X, y, **skw) as s:
        s.set_cv(5, fold)
        cv = bdist.FoldShardedCV(s, 5, data_type=skw.get("data_type", 1))
        out = cv.gs_path(1, 30) if path == "gs" else cv.sequential_path(np.arange(1, 16))
    return ref, out


@pytest.mark.parametrize("path", ["gs", "seq"])
@pytest.mark.parametrize("score_mode", [1, 2])
def test_lm_cv_paths_from_the_fit_primitive(gpu, path, score_mode):
    X, y, _, _ = synth.make_lm(1000, 300, 10)
    fold = synth.make_cv_folds(1000, 5)
    ref, out = _both(gpu, X, y, fold, path, score_mode=score_mode)
    assert out["best_T0"] == ref["best_T0"] and out["n_fits"] == ref["n_fits"]
    assert out["n_pdas_iters"] == ref["n_pdas_iters"]
    np.testing.assert_array_equal(out["cand_T0"], ref["cand_T0"])
    np.testing.assert_allclose(out["cand_ic"], ref["cand_ic"], rtol=1e-12)
    np.testing.assert_array_equal(np.nonzero(out["beta"])[0], np.nonzero(ref["beta"])[0])
    np.testing.assert_allclose(out["beta"], ref["beta"], rtol=1e-10)
    np.testing.assert_allclose([out["coef0"], out["train_loss"], out["ic"]],
                               [ref["coef0"], ref["train_loss"], ref["ic"]], rtol=1e-10)
    kw = dict(path_type=2, s_min=1, s_max=30) if path == "gs" else dict(sequence=np.arange(1, 16))
    want = P.trace(X, y, ic_type=3, is_cv=True, K=5, cv_fold_id=fold, **kw)
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(out["beta"])[0], sup)
    np.testing.assert_allclose(out["beta"][sup], want["beta"][sup], rtol=1e-6)
    np.testing.assert_allclose([out["coef0"], out["train_loss"], out["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-8)


def test_logistic_cv_gs_path_from_the_fit_primitive(gpu):
    """The fold fits of a logistic candidate start from the intercept the path handed to the full-data fit
    (the previous candidate's), which the records carry between ranks."""
    X, y, _, _ = synth.make_logistic(1500, 200, 6, seed=4)
    fold = synth.make_cv_folds(1500, 5)
    ref, out = _both(gpu, X, y, fold, "gs", data_type=2, model_type=2)
    assert out["best_T0"] == ref["best_T0"] and out["n_fits"] == ref["n_fits"]
    np.testing.assert_allclose(out["cand_ic"], ref["cand_ic"], rtol=1e-10)
    np.testing.assert_allclose(out["beta"], ref["beta"], rtol=1e-8)
    np.testing.assert_allclose([out["coef0"], out["train_loss"], out["ic"]],
                               [ref["coef0"], ref["train_loss"], ref["ic"]], rtol=1e-8)


@pytest.mark.parametrize("path", ["gs", "seq"])
def test_cold_start_pairs_from_the_fit_primitive(gpu, path):
    """is_warm_start = False: every (candidate, fold) pair is an independent fit (FoldShardedCV._round_cold deals them
    to all ranks); here on one rank, against the library's own cold CV paths and the oracle."""
    X, y, _, _ = synth.make_lm(1000, 300, 10)
    fold = synth.make_cv_folds(1000, 5)
    with gpu.Session(X, y, is_warm_start=False) as s:
        s.set_cv(5, fold)
        ref = s.gs_path(1, 30, ic_type=3, is_cv=True) if path == "gs" else \
            s.sequential_path(np.arange(1, 16), ic_type=3, is_cv=True)
        cv = bdist.FoldShardedCV(s, 5, is_warm_start=False)
        out = cv.gs_path(1, 30) if path == "gs" else cv.sequential_path(np.arange(1, 16))
    assert out["best_T0"] == ref["best_T0"] and out["n_fits"] == ref["n_fits"]
    assert out["n_pdas_iters"] == ref["n_pdas_iters"]
    np.testing.assert_array_equal(out["cand_T0"], ref["cand_T0"])
    np.testing.assert_allclose(out["cand_ic"], ref["cand_ic"], rtol=1e-12)
    np.testing.assert_allclose(out["beta"], ref["beta"], rtol=1e-10)
    kw = dict(path_type=2, s_min=1, s_max=30) if path == "gs" else dict(sequence=np.arange(1, 16))
    want = P.trace(X, y, ic_type=3, is_cv=True, K=5, cv_fold_id=fold, is_warm_start=False, **kw)
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(out["beta"])[0], sup)
    np.testing.assert_allclose(out["beta"][sup], want["beta"][sup], rtol=1e-6)
    assert out["n_fits"] == len(want["fits"])


@pytest.mark.parametrize("fam", ["lm", "logistic"])
def test_grouped_cv_paths_from_the_fit_primitive(gpu, fam):
    """Groups of size > 1: T0 counts groups and a fit returns the columns of the selected groups (ragged widths);
    bessx_session_fit hands them over padded to bessx_session_fit_width, the records carry them between ranks, and
    the fold-sharded paths reproduce the library's own grouped CV paths and the oracle's."""
    n, p = 900, 120
    g_index = np.concatenate([np.arange(0, 60, 3), np.arange(60, 100, 2), np.arange(100, 120, 5)]).astype(np.int32)
    if fam == "lm":
        X, y, _, _ = synth.make_lm(n, p, 9, seed=12)
        skw = dict(algorithm_type=2)
    else:
        X, y, _, _ = synth.make_logistic(n, p, 6, seed=12)
        skw = dict(algorithm_type=2, data_type=2, model_type=2)
    fold = synth.make_cv_folds(n, 4, seed=5)
    G = len(g_index)
    with gpu.Session(X, y, g_index=g_index, **skw) as s:
        assert s.fit_width(3) == 15 and s.fit_width(5) == 23 and s.fit_width(G) == p  # the widest groups: 4 of 5 columns
        s.set_cv(4, fold)
        ref = (s.gs_path(1, 12, ic_type=3, is_cv=True), s.sequential_path(np.arange(1, 9), ic_type=3, is_cv=True))
        one = s.fit(4, 0.0, fold=1)
        assert 4 <= len(one["support"]) <= s.fit_width(4) and np.all(np.diff(one["support"]) > 0)
    with gpu.Session(X, y, g_index=g_index, **skw) as s:
        s.set_cv(4, fold)
        out = (bdist.FoldShardedCV(s, 4, data_type=skw.get("data_type", 1)).gs_path(1, 12),
               bdist.FoldShardedCV(s, 4, data_type=skw.get("data_type", 1)).sequential_path(np.arange(1, 9)))
    for o, r in zip(out, ref):
        assert o["best_T0"] == r["best_T0"] and o["n_fits"] == r["n_fits"] and o["n_pdas_iters"] == r["n_pdas_iters"]
        np.testing.assert_allclose(o["cand_ic"], r["cand_ic"], rtol=1e-9)
        np.testing.assert_array_equal(np.nonzero(o["beta"])[0], np.nonzero(r["beta"])[0])
        np.testing.assert_allclose(o["beta"], r["beta"], rtol=1e-8)
        np.testing.assert_allclose([o["coef0"], o["train_loss"], o["ic"]], [r["coef0"], r["train_loss"], r["ic"]], rtol=1e-8)
    want = P.trace(X, y, ic_type=3, is_cv=True, K=4, cv_fold_id=fold, path_type=2, s_min=1, s_max=12, g_index=g_index,
                   **skw)
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(out[0]["beta"])[0], sup)
    np.testing.assert_allclose(out[0]["beta"][sup], want["beta"][sup], rtol=1e-5)


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("path", ["gs", "seq"])
def test_fold_subsets_through_the_library(gpu, world, path):
    """bessx_session_cv_eval: every rank evaluates ITS folds of a candidate in one library call -- the chains side by
    side, union fills -- and the full-data fit if it owns that chain.  Ranks = threads with a session each; every rank
    ends with the library's own single-process CV path."""
    from helpers import run_ranks
    X, y, _, _ = synth.make_lm(1000, 300, 10)
    fold = synth.make_cv_folds(1000, 5)
    with gpu.Session(X, y, score_mode=2) as s:
        s.set_cv(5, fold)
        ref = s.gs_path(1, 30, ic_type=3, is_cv=True) if path == "gs" else \
            s.sequential_path(np.arange(1, 16), ic_type=3, is_cv=True)

    def rank_fn(rank, comm):
        with gpu.Session(X, y, score_mode=2) as sr:
            sr.set_cv(5, fold)
            cv = bdist.FoldShardedCV(sr, 5, world, rank, comm=comm)
            out = cv.gs_path(1, 30) if path == "gs" else cv.sequential_path(np.arange(1, 16))
            return out, sr.counters(), len([u for u in cv.units if u != 5])

    for out, cnt, nfolds in run_ranks(world, rank_fn):
        assert out["best_T0"] == ref["best_T0"] and out["n_fits"] == ref["n_fits"]
        assert out["n_pdas_iters"] == ref["n_pdas_iters"]
        np.testing.assert_array_equal(out["cand_T0"], ref["cand_T0"])
        np.testing.assert_allclose(out["cand_ic"], ref["cand_ic"], rtol=1e-12)
        np.testing.assert_allclose(out["beta"], ref["beta"], rtol=1e-10)
        np.testing.assert_allclose([out["coef0"], out["train_loss"], out["ic"]],
                                   [ref["coef0"], ref["train_loss"], ref["ic"]], rtol=1e-10)
        assert cnt["cv_fold_contexts"] == 5
        assert (cnt["cv_side_by_side_rounds"] > 0) == (nfolds > 0)  # the folds took the chains' route, not fit by fit


def test_cv_eval_records(gpu):
    """The records of one call: [full,] folds in order, each what bessx_session_fit returns for that unit."""
    X, y, _, _ = synth.make_lm(800, 200, 8, seed=2)
    fold = synth.make_cv_folds(800, 4, seed=1)
    with gpu.Session(X, y, score_mode=2) as a, gpu.Session(X, y, score_mode=2) as b:
        a.set_cv(4, fold)
        b.set_cv(4, fold)
        recs = a.cv_eval(6, 0.0, True, folds=[1, 3])
        assert len(recs) == 3
        want = [b.fit(6, 0.0, -1), b.fit(6, 0.0, 1), b.fit(6, 0.0, 3)]
        for r, w in zip(recs, want):
            np.testing.assert_array_equal(r["support"], w["support"])
            np.testing.assert_allclose(r["beta"], w["beta"], rtol=1e-9)
            assert r["iters"] == w["iters"]
            np.testing.assert_allclose([r["train_loss"], r["test_loss"]], [w["train_loss"], w["test_loss"]], rtol=1e-10)
        # the next candidate continues the folds' own chains (cv_initial_model_param) inside the library
        nxt = a.cv_eval(7, 0.0, False, init_idx=recs[0]["support"], init_val=recs[0]["beta"], folds=[1, 3])
        w1 = b.fit(7, 0.0, 1, want[1]["support"], want[1]["beta"])
        np.testing.assert_array_equal(nxt[0]["support"], w1["support"])
        assert nxt[0]["iters"] == w1["iters"]
        with pytest.raises(gpu.BessxError):
            a.cv_eval(6, 0.0, False, folds=[3, 1])
        with pytest.raises(gpu.BessxError):
            a.cv_eval(6, 0.0, False, folds=[4])


def test_cache_reset_is_complete_before_the_fold_chains_read_the_map(gpu):
    """Round 4's one red driver run (GPUTEST_r04: rank 0 of `bench.py --gpus 2 --workload lm-cv-gs`, "an active column was
    missing from the Gram column cache").  bessx_session_reset_caches cleared the shared slot map with memsets queued on
    the session's stream; a rank that owns no full-data fit starts its next path with fold chains on streams of their
    own, which could read the previous path's map before the memsets ran -- harmless until the clearing lands in the
    MIDDLE of a fit (between a selection's lookup and the next score pass), which it did once in ~130 rehearsals.  Here
    the window is held open again and again: the session's stream is blocked for 0-400 us or 1-3 ms (test hook), the caches are
    reset, fold-only evaluations follow at once; every record must be that of a session that was never blocked."""
    X, y, _, _ = synth.make_lm(3000, 800, 10)
    fold = synth.make_cv_folds(3000, 5)
    folds = [0, 2, 4]
    first = [12, 19, 8, 5, 6, 7]  # golden-section points and the start of the sweep of bench's rehearsal
    with gpu.Session(X, y, score_mode=2) as a, gpu.Session(X, y, score_mode=2) as b:
        for s in (a, b):
            s.set_cv(5, fold)
        want = None
        for rep in range(200):
            # windows from the bare cost of a host function on the stream (0) over the length of one fit (tens of
            # microseconds) to several evaluations (milliseconds)
            a.debug_block_stream(-(rep * 7 % 400) if rep % 4 else 1 + rep % 3)
            a.reset_caches()
            got = [a.cv_eval(T0, 0.0, False, folds=folds) for T0 in first]
            if want is None:
                b.reset_caches()
                want = [b.cv_eval(T0, 0.0, False, folds=folds) for T0 in first]
            for T0, recs, wrecs in zip(first, got, want):
                for r, w in zip(recs, wrecs):
                    np.testing.assert_array_equal(r["support"], w["support"], err_msg="repetition %d T0 %d" % (rep, T0))
                    assert r["iters"] == w["iters"]
                    np.testing.assert_allclose(r["beta"], w["beta"], rtol=1e-9)
                    np.testing.assert_allclose([r["train_loss"], r["test_loss"]], [w["train_loss"], w["test_loss"]],
                                               rtol=1e-10)
