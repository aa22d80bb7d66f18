"""The k-path as links of ONE warm-start chain (bessx_session_sequential_path_chain, src/path.cpp:60-64) and the
stitched chunks of a multi-rank run (bess_amd.dist.StitchedKPath): the gathered candidates equal the single chain's for
every k -- the same supports, criteria and coefficients -- for LM (both score forms), logistic and Cox, cold and ladder
starts.  The ranks are threads of this process, each with a session of its own on the one GPU."""
import numpy as np
import pytest

from bess_amd import dist as bdist
from bess_amd import synth
from helpers import run_ranks

pytestmark = pytest.mark.gpu


def _hard(fam, n, p, seed=3):
    """Correlated columns, weak signal: a chunk started cold settles in other local fixed points than the warm chain."""
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((n, p))
    X = Z.copy()
    for j in range(1, p):
        X[:, j] = 0.8 * X[:, j - 1] + 0.6 * Z[:, j]
    beta = np.zeros(p)
    beta[rng.choice(p, 12, replace=False)] = rng.uniform(0.3, 1.0, 12) * rng.choice([-1.0, 1.0], 12)
    eta = X @ beta
    if fam == "lm":
        return X, eta + rng.standard_normal(n), dict(data_type=1, model_type=1)
    if fam == "logistic":
        return X, (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(float), dict(data_type=2, model_type=2)
    t = (-np.log(rng.uniform(size=n)) / np.exp(0.5 * eta)) ** 0.5
    c = np.quantile(t, 0.8) * rng.uniform(size=n) * 2
    st = (t < c).astype(float)
    order = np.argsort(np.minimum(t, c), kind="stable")
    return np.ascontiguousarray(X[order]), st[order], dict(data_type=3, model_type=4)


CASES = [("lm", dict(score_mode=2)), ("lm", dict(score_mode=1)), ("logistic", {}), ("cox", {})]


@pytest.mark.parametrize("fam,extra", CASES)
def test_chain_links_equal_the_single_chain(gpu, fam, extra):
    X, y, kw = _hard(fam, 1200, 150)
    kw.update(extra)
    seq = np.arange(1, 31)
    with gpu.Session(X, y, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)
        for keep in (True, False):
            head = s.sequential_path_chain(seq[:13], ic_type=3)
            assert head["stopped_at"] == -1 and len(head["last_idx"]) == 13
            tail = s.sequential_path_chain(seq[13:], ic_type=3, init_idx=head["last_idx"], init_val=head["last_val"],
                                           init_coef0=head["last_coef0"], keep_caches=keep)
            np.testing.assert_array_equal(head["cand_support"], single["cand_support"][:13, :13])
            np.testing.assert_array_equal(tail["cand_support"], single["cand_support"][13:])
            np.testing.assert_array_equal(tail["cand_iters"], single["cand_iters"][13:])
            np.testing.assert_allclose(tail["cand_ic"], single["cand_ic"][13:], rtol=1e-11)
            np.testing.assert_allclose(tail["cand_beta"], single["cand_beta"][13:], rtol=1e-8, atol=1e-12)
        # stop at the first candidate that equals the caller's: here the very first one
        again = s.sequential_path_chain(seq[13:], ic_type=3, init_idx=head["last_idx"], init_val=head["last_val"],
                                        init_coef0=head["last_coef0"], keep_caches=True,
                                        stop_support=single["cand_support"][13:], stop_beta=single["cand_beta"][13:])
        assert again["stopped_at"] == 0 and again["n_candidates"] == 1
        # ... and never, against a table of other supports
        wrong = single["cand_support"][13:].copy()
        wrong[:, 0] = (wrong[:, 0] + 1) % 150
        walk = s.sequential_path_chain(seq[13:], ic_type=3, init_idx=head["last_idx"], init_val=head["last_val"],
                                       init_coef0=head["last_coef0"], keep_caches=True, stop_support=wrong)
        assert walk["stopped_at"] == -1 and walk["n_candidates"] == 17
        np.testing.assert_array_equal(walk["cand_support"], single["cand_support"][13:])


@pytest.mark.parametrize("fam,extra", CASES)
@pytest.mark.parametrize("world,ladder", [(2, False), (3, True), (4, False)])
def test_stitched_chunks_equal_the_single_chain(gpu, fam, extra, world, ladder):
    X, y, kw = _hard(fam, 1200, 150)
    kw.update(extra)
    seq = np.arange(1, 33)
    with gpu.Session(X, y, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)
        # what the chunks alone would have returned (no stitching): they do differ from the single chain on this design
        lo = bdist.partition(len(seq), world, world - 1)[0]
        cold = s.sequential_path(seq[lo:], ic_type=3)
    differs = not np.array_equal(cold["cand_support"], single["cand_support"][lo:])

    def rank_fn(rank, comm):
        a, b = bdist.partition(len(seq), world, rank)
        k0 = int(seq[a])
        lead = sorted({k for k in (k0 // 8, k0 // 4, k0 // 2) if 1 <= k < k0}) if (ladder and a > 0) else []
        with gpu.Session(X, y, **kw) as sr:
            sk = bdist.StitchedKPath(sr, seq, world, rank, ic_type=3, lead=lead, comm=comm)
            first = sk.step()
            second = sk.step()  # a step starts cold again: same work, same result
        np.testing.assert_array_equal(first["chunk"]["cand_support"], second["chunk"]["cand_support"])
        assert first["stitch_refits_per_rank"] == second["stitch_refits_per_rank"]
        return second

    res = run_ranks(world, rank_fn)
    W = single["cand_support"].shape[1]
    for r, rep in enumerate(res):
        a, b = bdist.partition(len(seq), world, r)
        c = rep["chunk"]
        sup = np.full((b - a, W), -1, dtype=np.int32)
        sup[:, :c["cand_support"].shape[1]] = c["cand_support"]
        np.testing.assert_array_equal(sup, single["cand_support"][a:b])
        np.testing.assert_array_equal(c["cand_iters"], single["cand_iters"][a:b])
        np.testing.assert_allclose(c["cand_ic"], single["cand_ic"][a:b], rtol=1e-11)
        np.testing.assert_allclose(rep["ic_curve"], single["cand_ic"], rtol=1e-11)
        assert rep["best_k"] == single["best_T0"]
    assert sum(res[0]["stitch_refits_per_rank"]) >= world - 1
    if differs and not ladder:
        assert sum(res[0]["stitch_refits_per_rank"]) > world - 1  # candidates were really replaced


@pytest.mark.parametrize("mode", [2, 1])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_coarse_lead_chunks_equal_the_single_chain(gpu, mode, world):
    """Round 6, the N-rank k-path's default for LM: every rank walks the one-GPU path's coarse levels below its chunk and
    the level just below it as LEAD fits (bessx_path_chain.lead_levels; no communication), then its chunk warm from the
    last lead model -- as chunk chains where the link is long enough (p >= 2048) --, and the stitch makes the gathered
    path the single chain's (src/path.cpp:60-64)."""
    X, y, _, _ = synth.make_lm(3000, 2304, 40, seed=11)
    seq = np.arange(1, 97)
    with gpu.Session(X, y, score_mode=mode) as s:
        single = s.sequential_path(seq, ic_type=3)

    def rank_fn(rank, comm):
        with gpu.Session(X, y, score_mode=mode) as sr:
            sk = bdist.StitchedKPath(sr, seq, world, rank, ic_type=3, comm=comm, coarse_lead=True)
            lv = sk.lead_levels()
            assert (rank == 0 and lv.size == 0) or (lv.size >= 1 and lv[-1] == sk.seq[0] - 1 and np.all(np.diff(lv) > 0))
            rep = sk.step()
            rep["chains"] = sr.counters()["kpath_chains_last_path"] if sr.counters()["kpath_chunked_paths"] else 1
        return rep

    res = run_ranks(world, rank_fn)
    W = single["cand_support"].shape[1]
    for r, rep in enumerate(res):
        a, b = bdist.partition(len(seq), world, r)
        c = rep["chunk"]
        sup = np.full((b - a, W), -1, dtype=np.int32)
        sup[:, :c["cand_support"].shape[1]] = c["cand_support"]
        np.testing.assert_array_equal(sup, single["cand_support"][a:b])
        np.testing.assert_array_equal(c["cand_iters"], single["cand_iters"][a:b])
        # (the same supports reached from another starting model: the iterative solve of the covariance form ends within
        # its 1e-13 residual bound of the same solution, the criterion agrees to ~1e-11)
        np.testing.assert_allclose(rep["ic_curve"], single["cand_ic"], rtol=1e-9)
        assert rep["best_k"] == single["best_T0"]
    if mode == 2 and world <= 4:  # links of >= 24 levels behind lead fits run as chunk chains (covariance form)
        assert any(rep["chains"] >= 2 for rep in res[1:])


def test_lead_fits_are_refused_where_they_do_not_apply(gpu):
    X, y, _, _ = synth.make_lm(600, 100, 5)
    with gpu.Session(X, y) as s:
        with pytest.raises(gpu.BessxError):  # levels must lie below the link's first
            s.sequential_path_chain([10, 11], ic_type=3, lead_levels=[5, 10])
        with pytest.raises(gpu.BessxError):  # ... and ascend
            s.sequential_path_chain([10, 11], ic_type=3, lead_levels=[6, 5])
        a = s.sequential_path_chain([10, 11, 12], ic_type=3, lead_levels=[3, 9])
        b = s.sequential_path(np.arange(1, 13), ic_type=3)
        # (on this easy design the lead chain 3 -> 9 -> 10 meets the chain 1 -> 2 -> ... -> 10)
        np.testing.assert_array_equal(a["cand_support"][:, :12], b["cand_support"][9:12, :12])
    Xl, yl, kw = _hard("logistic", 600, 60)
    with gpu.Session(Xl, yl, **kw) as s:
        with pytest.raises(gpu.BessxError) as e:
            s.sequential_path_chain([10, 11], ic_type=3, lead_levels=[5])
        assert e.value.code == 3


def test_chain_refuses_what_it_cannot_continue(gpu):
    X, y, _, _ = synth.make_lm(600, 100, 5)
    with gpu.Session(X, y) as s:
        s.set_cv(4, synth.make_cv_folds(600, 4))
        with pytest.raises(gpu.BessxError) as e:
            s.sequential_path_chain([3, 4], ic_type=3, is_cv=True, init_idx=[1, 2], init_val=[0.5, 0.5])
        assert e.value.code == 3
        with pytest.raises(gpu.BessxError):
            s.sequential_path_chain([3, 4], ic_type=3, init_idx=[1, 100], init_val=[0.5, 0.5])


def test_cooperative_prefill_fills_the_same_cache(gpu):
    """bessx_session_cov_prefill_*: three ranks (threads, a session each) form one group of 32 Gram columns each, exchange
    the p x 32 blocks and end with the cache one rank alone would have built -- bit for bit --, which is X^T x_c of the
    normalised design; a path that continues on it (keep_caches) returns what it returns without it, with fewer passes."""
    X, y, _, _ = synth.make_lm(1500, 400, 10)
    seq = np.arange(1, 41)
    with gpu.Session(X, y, score_mode=2) as s:
        plain = s.sequential_path(seq, ic_type=3)
        passes_plain = s.counters()["passes_over_X"]
        scores = s.marginal_scores()
        cols = np.argsort(-scores, kind="stable")[:96].astype(np.int32)
        s.cov_prefill_begin(cols)
        s.cov_prefill_compute(0, 3)
        full = s.cov_prefill_export(0, 3)
        s.cov_prefill_end()
        before = s.counters()["passes_over_X"]
        kept = s.sequential_path_chain(seq, ic_type=3, keep_caches=True)
        passes_kept = s.counters()["passes_over_X"] - before
        xm, xn, ym = s.normalization()
    np.testing.assert_array_equal(kept["cand_support"], plain["cand_support"])
    np.testing.assert_allclose(kept["cand_ic"], plain["cand_ic"], rtol=1e-11)
    assert passes_kept < passes_plain
    Xn = np.sqrt(1500.0) * (X - xm) / xn
    want = (Xn.T @ Xn[:, cols]).T.ravel()  # column after column, p entries each
    np.testing.assert_allclose(full, want, rtol=1e-10, atol=1e-7)
    # the first PDAS iteration of a cold fit ranks exactly these scores
    d = Xn.T @ (y - ym) / 1500.0
    np.testing.assert_allclose(scores, d * d, rtol=1e-9, atol=1e-12)

    # every rank ends with the same blocks: read them back before anything restarts the cache
    def rank_blocks(rank, comm):
        with gpu.Session(X, y, score_mode=2) as sr:
            sc = sr.marginal_scores()
            cl = np.argsort(-sc, kind="stable")[:96].astype(np.int32)
            sr.cov_prefill_begin(cl)
            lo, hi = bdist.partition(3, 3, rank)
            sr.cov_prefill_compute(lo, hi - lo)
            mine = sr.cov_prefill_export(lo, hi - lo)
            for r, blk in enumerate(comm.all_gather(mine, 3)):
                if r != rank:
                    sr.cov_prefill_import(r, 1, blk)
            got = sr.cov_prefill_export(0, 3)
            sr.cov_prefill_end()
            out = sr.sequential_path_chain(seq, ic_type=3, keep_caches=True)
            return got, out["cand_support"]

    for got, sup in run_ranks(3, rank_blocks):
        np.testing.assert_array_equal(got, full)
        np.testing.assert_array_equal(sup, plain["cand_support"])
    with gpu.Session(X, y, score_mode=1) as s1:
        with pytest.raises(gpu.BessxError) as e:
            s1.cov_prefill_begin(cols)
        assert e.value.code == 3
    with gpu.Session(X, y, score_mode=2) as s2:
        with pytest.raises(gpu.BessxError):
            s2.cov_prefill_begin(cols[:40])  # not a multiple of 32
        with pytest.raises(gpu.BessxError):
            s2.cov_prefill_begin(np.concatenate([cols[:31], cols[:1]]))  # a column twice


@pytest.mark.parametrize("world", [2, 4])
def test_stitched_chunks_with_the_cooperative_prefill(gpu, world):
    X, y, kw = _hard("lm", 1200, 400)
    seq = np.arange(1, 33)
    with gpu.Session(X, y, score_mode=2, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)

    def rank_fn(rank, comm):
        with gpu.Session(X, y, score_mode=2, **kw) as sr:
            return bdist.StitchedKPath(sr, seq, world, rank, ic_type=3, comm=comm, prefill=128).step()

    res = run_ranks(world, rank_fn)
    for r, rep in enumerate(res):
        a, b = bdist.partition(len(seq), world, r)
        c = rep["chunk"]["cand_support"]
        np.testing.assert_array_equal(c, single["cand_support"][a:b, :c.shape[1]])
        np.testing.assert_allclose(rep["ic_curve"], single["cand_ic"], rtol=1e-11)
        assert min(rep["prefill_seconds_per_rank"]) > 0.0


@pytest.mark.parametrize("world", [2, 4])
def test_stitched_chunks_with_the_pilot_prefill(gpu, world):
    """pilot_prefill: marginal list, the same pilot fit on every rank, a second shared list from the pilot's scores
    (appended to the cache in the same slots everywhere), chunks beyond the pilot's level started warm from its model.
    Starting points and cache contents only: the stitched path is the single chain's."""
    X, y, kw = _hard("lm", 1200, 400)
    seq = np.arange(1, 33)
    with gpu.Session(X, y, score_mode=2, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)

    def rank_fn(rank, comm):
        with gpu.Session(X, y, score_mode=2, **kw) as sr:
            sk = bdist.StitchedKPath(sr, seq, world, rank, ic_type=3, comm=comm, prefill=64, pilot=(12, 96))
            first = sk.step()
            again = sk.step()
            bd, slot = sr.cov_state()
        np.testing.assert_array_equal(first["chunk"]["cand_support"], again["chunk"]["cand_support"])
        return again, slot

    res = run_ranks(world, rank_fn)
    for r, (rep, slot) in enumerate(res):
        a, b = bdist.partition(len(seq), world, r)
        c = rep["chunk"]["cand_support"]
        np.testing.assert_array_equal(c, single["cand_support"][a:b, :c.shape[1]])
        np.testing.assert_allclose(rep["ic_curve"], single["cand_ic"], rtol=1e-11)
        assert int(np.sum(slot >= 0)) >= 64 + 96  # both shared lists are in every rank's cache


@pytest.mark.parametrize("world,n_first", [(2, 64), (4, 0), (3, 32)])
def test_pilot_fit_with_shared_wide_fills(gpu, world, n_first):
    """bessx_session_set_fill_hook: the pilot fit every rank runs identically parks on missing Gram columns; the ranks
    then form ONE 32-column group each of a list `wide` long (the missing columns + the best uncached ones by that
    iteration's scores) instead of every rank forming the same columns privately.  Same pilot model as with private
    fills, same stitched path as the single chain, fewer passes over X per rank."""
    X, y, kw = _hard("lm", 1200, 400)
    seq = np.arange(1, 33)
    with gpu.Session(X, y, score_mode=2, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)
        s.sequential_path_chain([1], ic_type=3)  # (cold caches)
        alone = s.sequential_path_chain([14], ic_type=3, keep_caches=True)

    def rank_fn(rank, comm):
        with gpu.Session(X, y, score_mode=2, **kw) as sr:
            calls = []
            orig = bdist._share_and_exchange

            def counted(session, w, r, c, ng):
                calls.append(ng)
                return orig(session, w, r, c, ng)

            model = None
            if rank == 0:  # (the pilot model itself, once: equal to the fit with private fills)
                with gpu.Session(X, y, score_mode=2, **kw) as s1:
                    s1.set_fill_hook(lambda ng: (s1.cov_prefill_compute(0, ng), s1.cov_prefill_end()), 32 * world)
                    model = s1.sequential_path_chain([14], ic_type=3)
                    s1.set_fill_hook(None)
                    assert s1.counters()["shared_wide_fills"] >= 1
            sk = bdist.StitchedKPath(sr, seq, world, rank, ic_type=3, comm=comm, prefill=n_first, pilot=(14, 64, 32 * world))
            first = sk.step()
            again = sk.step()
            wide_fills = sr.counters()["shared_wide_fills"]
            passes = sr.counters()["passes_over_X"]
        np.testing.assert_array_equal(first["chunk"]["cand_support"], again["chunk"]["cand_support"])
        return again, wide_fills, passes, model

    res = run_ranks(world, rank_fn)
    for r, (rep, wide_fills, passes, model) in enumerate(res):
        a, b = bdist.partition(len(seq), world, r)
        c = rep["chunk"]["cand_support"]
        np.testing.assert_array_equal(c, single["cand_support"][a:b, :c.shape[1]])
        np.testing.assert_allclose(rep["ic_curve"], single["cand_ic"], rtol=1e-11)
        assert wide_fills >= 2  # (two steps, at least one wide fill in each pilot fit)
    assert len({q[1] for q in res}) == 1  # every rank parked equally often
    model = res[0][3]
    np.testing.assert_array_equal(model["cand_support"], alone["cand_support"])
    np.testing.assert_allclose(model["cand_beta"], alone["cand_beta"], rtol=1e-9, atol=1e-13)
