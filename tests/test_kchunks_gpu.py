"""sequential_path of one LM problem as several chunk chains side by side on ONE Gram column cache
(bess_amd/csrc/bessx_kchunks.cpp; BESSX_KPATH_CHAINS): a coarse chain over the chunk boundaries, the chunks on fit
contexts of their own, fills of the shared cache while every other chain stands still, the stitch of
bess_amd.dist.StitchedKPath inside the library.  The path returned is the single warm-start chain's (src/path.cpp:60-64):
every candidate's support, iteration count, criterion and coefficients."""
import numpy as np
import pytest

from bess_amd import synth
from oracle import port_ctypes as P
from test_stitch_gpu import _hard

pytestmark = pytest.mark.gpu


def _same_path(a, b, beta_rtol=1e-8):
    np.testing.assert_array_equal(a["cand_support"], b["cand_support"])
    np.testing.assert_array_equal(a["cand_iters"], b["cand_iters"])
    np.testing.assert_array_equal(a["cand_T0"], b["cand_T0"])
    # (the loss of a near-perfect fit is a difference of large numbers, |y|^2 - beta.(X^T y): 1e-9 of the criterion)
    np.testing.assert_allclose(a["cand_ic"], b["cand_ic"], rtol=1e-9)
    np.testing.assert_allclose(a["cand_train_loss"], b["cand_train_loss"], rtol=1e-7)
    np.testing.assert_allclose(a["cand_beta"], b["cand_beta"], rtol=beta_rtol, atol=1e-12)
    np.testing.assert_allclose(a["cand_coef0"], b["cand_coef0"], rtol=1e-9, atol=1e-12)
    assert a["best_T0"] == b["best_T0"] and a["n_candidates"] == b["n_candidates"]
    assert a["n_fits"] == b["n_fits"] and a["n_pdas_iters"] == b["n_pdas_iters"]
    np.testing.assert_allclose(a["beta"], b["beta"], rtol=beta_rtol, atol=1e-12)
    np.testing.assert_allclose([a["coef0"], a["ic"], a["train_loss"]], [b["coef0"], b["ic"], b["train_loss"]], rtol=1e-7)


@pytest.mark.parametrize("chains", [2, 3, 4, 8])
def test_chunk_chains_return_the_single_chain(gpu, monkeypatch, chains):
    X, y, _, _ = synth.make_lm(2500, 700, 20, seed=11)
    seq = np.arange(1, 65)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y) as s:
        single = s.sequential_path(seq, ic_type=3)
        assert s.counters()["kpath_chunked_paths"] == 0
    monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
    with gpu.Session(X, y) as s:
        first = s.sequential_path(seq, ic_type=3)
        cnt = s.counters()
        assert cnt["kpath_chunked_paths"] == 1 and cnt["kpath_stitch_refits"] >= chains - 1
        again = s.sequential_path(seq, ic_type=3)          # the contexts and host threads are kept
        odd = s.sequential_path(seq[::2], ic_type=2)       # levels that are not consecutive
        lam = s.sequential_path(seq, [0.0, 0.1], ic_type=3)  # two lambdas: the snake order, not chunked
        gs = s.gs_path(1, 40, ic_type=3)
        assert s.counters()["kpath_chunked_paths"] == 3
    _same_path(first, single)
    _same_path(again, single)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y) as s:
        _same_path(odd, s.sequential_path(seq[::2], ic_type=2))
        _same_path(lam, s.sequential_path(seq, [0.0, 0.1], ic_type=3))
        want_gs = s.gs_path(1, 40, ic_type=3)
    np.testing.assert_array_equal(gs["cand_support"], want_gs["cand_support"])
    want = P.trace(X, y, ic_type=3, sequence=seq)
    np.testing.assert_allclose(first["cand_ic"], want["ic_calls"], rtol=1e-8)
    assert list(first["cand_iters"]) == [len(f["iters"]) for f in want["fits"]]


def test_chunk_chains_on_a_design_where_cold_chunks_go_astray(gpu, monkeypatch):
    """Correlated columns, weak signal: chunks started from the coarse chain's models land in other local fixed points
    than the warm chain -- the stitch replaces whole runs of candidates, in more than one round."""
    X, y, kw = _hard("lm", 1200, 400)
    seq = np.arange(1, 49)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y, score_mode=2, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)
    for chains in (3, 6):
        monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
        with gpu.Session(X, y, score_mode=2, **kw) as s:
            got = s.sequential_path(seq, ic_type=3)
            assert s.counters()["kpath_chunked_paths"] == 1
            refits = s.counters()["kpath_stitch_refits"]
        np.testing.assert_array_equal(got["cand_support"], single["cand_support"])
        np.testing.assert_allclose(got["cand_ic"], single["cand_ic"], rtol=1e-10)
        assert got["best_T0"] == single["best_T0"]
        assert refits >= chains - 1


def test_chunk_chains_fill_the_shared_cache_one_at_a_time(gpu, monkeypatch):
    """Many chunks on a design whose coarse chain leaves columns uncached: chunk chains park, wait for the others to
    stand still, fill, and the path is still the single chain's.  Also after a CV path and single fits on the session."""
    X, y, _, _ = synth.make_lm(3000, 1500, 60, seed=5)
    seq = np.arange(1, 129)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y) as s:
        single = s.sequential_path(seq, ic_type=3)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "8")
    with gpu.Session(X, y) as s:
        s.set_cv(3, synth.make_cv_folds(3000, 3, seed=4))
        cv = s.sequential_path(np.arange(1, 9), ic_type=3, is_cv=True)   # (CV paths are not chunked)
        one = s.fit(17)
        for _ in range(3):
            got = s.sequential_path(seq, ic_type=3)
            _same_path(got, single)
        cnt = s.counters()
    assert len(one["support"]) == 17 and cv["n_candidates"] == 8
    # (a session with CV folds shares its fills with the folds' caches: no chunk chains there)
    assert cnt["kpath_chunked_paths"] == 0
    with gpu.Session(X, y) as s:
        for _ in range(3):
            _same_path(s.sequential_path(seq, ic_type=3), single)
        cnt = s.counters()
    assert cnt["kpath_chunked_paths"] == 3
    assert cnt["kpath_chunk_fills"] >= 0


def test_no_chunk_chains_where_they_do_not_apply(gpu, monkeypatch):
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "4")
    X, y, _, _ = synth.make_lm(1500, 300, 10, seed=2)
    seq = np.arange(1, 41)
    with gpu.Session(X, y, score_mode=1) as s:  # streaming form: chunked since round 5, but not below the automatic size
        s.sequential_path(seq, ic_type=3)
        assert s.counters()["kpath_chunked_paths"] == 1
        s.set_kpath_chains(0)  # automatic: 40 levels on 1500 x 300 are below the threshold (48 levels, n p >= 1e8)
        s.sequential_path(seq, ic_type=3)
        assert s.counters()["kpath_chunked_paths"] == 1
    with gpu.Session(X, y, is_warm_start=False) as s:
        s.sequential_path(seq, ic_type=3)
        assert s.counters()["kpath_chunked_paths"] == 0
    with gpu.Session(X, y) as s:
        s.sequential_path(seq[::-1].copy(), ic_type=3)  # descending levels
        s.sequential_path(seq[:12], ic_type=3)          # too short for 4 chunks of 8
        assert s.counters()["kpath_chunked_paths"] == 0
        s.sequential_path(seq, ic_type=3)
        assert s.counters()["kpath_chunked_paths"] == 1


def _glm_case(fam):
    if fam == "poisson":
        X, y, _, _ = synth.make_poisson(3000, 300, 12, seed=8)
        return X, y, dict(data_type=2, model_type=3)
    X, y, kw = _hard(fam, 2500 if fam == "cox" else 2000, 200)
    return X, y, kw


@pytest.mark.parametrize("fam", ["logistic", "poisson", "cox"])
@pytest.mark.parametrize("chains", [2, 3])
def test_chunk_chains_of_the_irls_and_newton_families(gpu, monkeypatch, fam, chains):
    """Logistic, Poisson, Cox: no Gram column cache to share -- the chunks start cold, side by side (one chain's IRLS /
    Newton steps beside another's pass over X), each on a context that owns everything a fit of the family writes, and
    are stitched.  Same candidates as the single chain: supports, iteration counts, criteria, coefficients."""
    X, y, kw = _glm_case(fam)
    seq = np.arange(1, 33)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)
        steps1 = s.submodel_steps() if hasattr(s, "submodel_steps") else None
    monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
    with gpu.Session(X, y, **kw) as s:
        got = s.sequential_path(seq, ic_type=3)
        again = s.sequential_path(seq, ic_type=3)
        one = s.fit(9)                               # the session's own state is still good for single fits
        cnt = s.counters()
    assert cnt["kpath_chunked_paths"] == 2 and cnt["kpath_stitch_refits"] >= 2 * (chains - 1)
    for out in (got, again):
        np.testing.assert_array_equal(out["cand_support"], single["cand_support"])
        np.testing.assert_array_equal(out["cand_iters"], single["cand_iters"])
        np.testing.assert_allclose(out["cand_ic"], single["cand_ic"], rtol=1e-10)
        np.testing.assert_allclose(out["cand_beta"], single["cand_beta"], rtol=1e-7, atol=1e-10)
        assert out["best_T0"] == single["best_T0"] and out["n_pdas_iters"] == single["n_pdas_iters"]
    assert len(one["support"]) == 9
    del steps1


def test_chunk_chains_with_weights_always_select_screening_and_the_drop_in_entry(gpu, monkeypatch):
    """Sessions with observation weights, always-selected columns and screening run chunked too (results in the
    caller's column numbering); the pywrap_bess drop-in creates a session per call: contexts and host threads come and
    go with it."""
    rng = np.random.default_rng(4)
    X, y, _, _ = synth.make_lm(2500, 900, 15, seed=21)
    w = rng.uniform(0.5, 2.0, 2500)
    seq = np.arange(1, 57)
    variants = [dict(weight=w), dict(always_select=[3, 700]), dict(is_screening=True, screening_size=500),
                dict(weight=w, always_select=[10], is_screening=True, screening_size=600, algorithm_type=5)]
    for kw in variants:
        outs = {}
        for chains in (1, 3):
            monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
            with gpu.Session(X, y, **kw) as s:
                outs[chains] = s.sequential_path(seq, [0.05] if kw.get("algorithm_type") == 5 else [0.0], ic_type=3)
                assert (s.counters()["kpath_chunked_paths"] > 0) == (chains == 3)
        _same_path(outs[3], outs[1])
    res = {}
    for chains in (1, 3):
        monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
        res[chains] = [gpu.pywrap_bess(X, y, 1, np.ones(2500), True, 1, 1, 20, 5, 1, True, 3, False, 5, np.arange(900), [0.0],
                                       seq, [0.0], 1, 20, 10, 1e-4, 0.01, 10.0, 10, False, 0, 1, [], 0.0, 900)
                       for _ in range(2)]
    for a, b in zip(res[3], res[1]):
        for u, v in zip(a, b):
            np.testing.assert_allclose(np.asarray(u, dtype=float), np.asarray(v, dtype=float), rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("fam", ["lm", "logistic"])
def test_links_of_a_longer_chain_run_as_chunk_chains_too(gpu, monkeypatch, fam):
    """bessx_session_sequential_path_chain -- a rank's chunk of a multi-GPU k-path -- qualifies when it only starts from a
    given model (no stop table): same link as one chain, same hand-over model; and the stitched 2-rank path built from
    such links is the single chain's."""
    from bess_amd import dist as bdist
    from helpers import run_ranks
    if fam == "lm":
        X, y, _, _ = synth.make_lm(2500, 700, 20, seed=11)
        kw = {}
    else:
        X, y, kw = _hard("logistic", 2000, 200)
    seq = np.arange(1, 65)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)
        head = s.sequential_path_chain(seq[:20], ic_type=3)
        tail1 = s.sequential_path_chain(seq[20:], ic_type=3, init_idx=head["last_idx"], init_val=head["last_val"],
                                        init_coef0=head["last_coef0"], keep_caches=True)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "3")
    with gpu.Session(X, y, **kw) as s:
        for keep in (True, False):
            h3 = s.sequential_path_chain(seq[:20], ic_type=3)
            t3 = s.sequential_path_chain(seq[20:], ic_type=3, init_idx=h3["last_idx"], init_val=h3["last_val"],
                                         init_coef0=h3["last_coef0"], keep_caches=keep)
            assert t3["stopped_at"] == -1
            np.testing.assert_array_equal(t3["cand_support"], tail1["cand_support"])
            np.testing.assert_array_equal(t3["cand_iters"], tail1["cand_iters"])
            np.testing.assert_allclose(t3["cand_ic"], tail1["cand_ic"], rtol=1e-9)
            np.testing.assert_array_equal(t3["last_idx"], tail1["last_idx"])
            np.testing.assert_allclose(t3["last_val"], tail1["last_val"], rtol=1e-8, atol=1e-12)
        assert s.counters()["kpath_chunked_paths"] >= 2
        # a refit with a stop table stays one chain
        before = s.counters()["kpath_chunked_paths"]
        r = s.sequential_path_chain(seq[20:], ic_type=3, init_idx=h3["last_idx"], init_val=h3["last_val"],
                                    init_coef0=h3["last_coef0"], keep_caches=True, stop_support=tail1["cand_support"],
                                    stop_beta=tail1["cand_beta"])
        assert r["stopped_at"] == 0 and s.counters()["kpath_chunked_paths"] == before

    def rank_fn(rank, comm):
        with gpu.Session(X, y, **kw) as sr:
            rep = bdist.StitchedKPath(sr, seq, 2, rank, ic_type=3, comm=comm).step()
            return rep, sr.counters()["kpath_chunked_paths"]

    for r, (rep, chunked) in enumerate(run_ranks(2, rank_fn)):
        a, b = bdist.partition(len(seq), 2, r)
        c = rep["chunk"]["cand_support"]
        np.testing.assert_array_equal(c, single["cand_support"][a:b, :c.shape[1]])
        np.testing.assert_allclose(rep["ic_curve"], single["cand_ic"], rtol=1e-9)
        assert chunked >= 1


def test_the_number_of_chains_can_change_between_paths(gpu, monkeypatch):
    """bessx_session_set_kpath_chains between paths of one session: more chains than before start more host threads
    (the pool is started again), fewer leave the extra ones idle; always the single chain's candidates."""
    monkeypatch.delenv("BESSX_KPATH_CHAINS", raising=False)
    X, y, _, _ = synth.make_lm(2500, 700, 20, seed=11)
    seq = np.arange(1, 65)
    with gpu.Session(X, y) as s:
        s.set_kpath_chains(1)
        single = s.sequential_path(seq, ic_type=3)
        for chains in (2, 5, 3, 8, 2, 1, 4):
            s.set_kpath_chains(chains)
            _same_path(s.sequential_path(seq, ic_type=3), single)
            if chains > 1:
                assert s.counters()["kpath_chains_last_path"] == chains
        with pytest.raises(Exception):
            s.set_kpath_chains(9)


def test_the_stitch_gives_up_where_the_chunks_do_not_merge(gpu, monkeypatch):
    """Correlated columns, weak signal, a long path: the chunks' own chains run on other trajectories than the warm chain.
    A refit that has not met its chunk within its budget ends the stitch -- the rest of the path is walked as one chain
    from the last settled model (still the single chain's candidates) -- and the session's automatic choice becomes one
    chain; an explicit bessx_session_set_kpath_chains tries again."""
    monkeypatch.delenv("BESSX_KPATH_CHAINS", raising=False)
    rng = np.random.default_rng(3)
    n, p = 4000, 2400
    Z = rng.standard_normal((n, p))
    X = Z.copy()
    for j in range(1, p):
        X[:, j] = 0.6 * X[:, j - 1] + 0.8 * Z[:, j]
    beta = np.zeros(p)
    beta[rng.choice(p, 60, replace=False)] = rng.uniform(0.2, 1.0, 60) * rng.choice([-1.0, 1.0], 60)
    y = X @ beta + 2.0 * rng.standard_normal(n)
    seq = np.arange(1, 161)
    with gpu.Session(X, y) as s:
        s.set_kpath_chains(1)
        single = s.sequential_path(seq, ic_type=3)
        s.set_kpath_chains(0)                      # automatic: chunk chains are tried ...
        first = s.sequential_path(seq, ic_type=3)
        c1 = s.counters()
        assert c1["kpath_chunked_paths"] == 1 and c1["kpath_stitch_giveups"] == 1
        second = s.sequential_path(seq, ic_type=3)  # ... once
        c2 = s.counters()
        assert c2["kpath_chunked_paths"] == 1
        s.set_kpath_chains(4)                      # an explicit choice is honoured (and gives up again, per path)
        third = s.sequential_path(seq, ic_type=3)
        c3 = s.counters()
        assert c3["kpath_chunked_paths"] == 2 and c3["kpath_stitch_giveups"] == 2
    for out in (first, second, third):
        _same_path(out, single)


@pytest.mark.parametrize("chains", [2, 4, 8])
def test_merged_launches_over_the_chunk_chains_return_the_single_chain(gpu, monkeypatch, chains):
    """Round 5 (bessx_dev.h, McChain; test hook kchunks_merged=1): the chunk phase as merged launches on ONE stream -- every
    chain a workgroup of the same launch, the candidate / iteration sequencing in device memory, parked chains served by
    one union fill -- gives the single chain's candidates like the stream-per-chain form does: consecutive levels, a
    path with gaps between the levels (no arg-max start), and a design whose chunks need fills of their own."""
    from helpers import hooks
    X, y, _, _ = synth.make_lm(2500, 700, 20, seed=11)
    seq = np.arange(1, 65)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y) as s:
        single = s.sequential_path(seq, ic_type=3)
        single_odd = s.sequential_path(seq[::2], ic_type=2)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
    hooks(monkeypatch, kchunks_merged=1)
    with gpu.Session(X, y) as s:
        first = s.sequential_path(seq, ic_type=3)
        cnt = s.counters()
        assert cnt["kpath_chunked_paths"] == 1 and cnt["kpath_merged_chunk_phases"] == 1
        again = s.sequential_path(seq, ic_type=3)
        odd = s.sequential_path(seq[::2], ic_type=2)
        assert s.counters()["kpath_merged_chunk_phases"] == 3
    _same_path(first, single)
    _same_path(again, single)
    _same_path(odd, single_odd)
    # a small cache-speculation width and many chunks: the chunks park for fills of their own (union fills)
    X2, y2, _, _ = synth.make_lm(3000, 2600, 60, seed=3)
    seq2 = np.arange(1, 161)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    hooks(monkeypatch, kchunks_merged=0)
    with gpu.Session(X2, y2) as s:
        want = s.sequential_path(seq2, ic_type=3)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
    hooks(monkeypatch, kchunks_merged=1)
    with gpu.Session(X2, y2) as s:
        got = s.sequential_path(seq2, ic_type=3)
        assert s.counters()["kpath_merged_chunk_phases"] == 1
    _same_path(got, want)


@pytest.mark.parametrize("chains", [2, 3])
def test_chunk_chains_of_the_streaming_lm_path(gpu, monkeypatch, chains):
    """Round 5: the STREAMING form of the LM score pass (score_mode = 1: every PDAS iteration reads X once -- the
    formulation north_star prescribes) as chunk chains, like the IRLS / Newton families: the chunks start cold side by
    side on contexts that own what enqueue_lm_slot writes (score sums, residual, column list, slab partials, the
    incremental Gram and its cache) and are stitched into the single chain."""
    X, y, _, _ = synth.make_lm(2500, 700, 20, seed=11)
    seq = np.arange(1, 65)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y, score_mode=1) as s:
        single = s.sequential_path(seq, ic_type=3)
        assert s.counters()["kpath_chunked_paths"] == 0
    monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
    with gpu.Session(X, y, score_mode=1) as s:
        first = s.sequential_path(seq, ic_type=3)
        cnt = s.counters()
        assert cnt["kpath_chunked_paths"] == 1 and cnt["kpath_chains_last_path"] == chains
        again = s.sequential_path(seq, ic_type=3)
        w = s.sequential_path(seq[:40], ic_type=2)
    _same_path(first, single)
    _same_path(again, single)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y, score_mode=1) as s:
        _same_path(w, s.sequential_path(seq[:40], ic_type=2))


@pytest.mark.parametrize("variant", ["rendezvous", "staged_own_stream", "staged_fill_stream", "pipeline", "late_stitch",
                                     "rendezvous_late_stitch"])
def test_fill_disciplines_of_the_chunk_chains_return_the_single_chain(gpu, monkeypatch, variant):
    """How a chunk chain's fill of the shared Gram column cache is kept from the other chains (bessx_sync.h:
    FillRendezvous): round 4's rendezvous -- every other chain stands still (kchunks_staged=0) -- and round 5's STAGED
    fills (default): the fill writes slots nobody can look up until its last launch publishes them (k_cov_publish_slots),
    on the filling chain's own stream (kchunks_reserve=0) or on the session's fill stream that leaves some compute units
    to the other chains' kernels; and the coarse chain BESIDE the chunks instead of in front of them (kchunks_pipeline=1:
    measured slower, kept selectable).  And when the first round of the stitch runs: early, on the thread of the chunk in
    front as soon as that chunk is walked (default), or after all chunks (kchunks_early_stitch=0).  Designs whose chunks need
    many fills of their own; the same candidates each way, path after path on one session."""
    from helpers import hooks
    hk = {"rendezvous": dict(kchunks_staged=0), "staged_own_stream": dict(kchunks_staged=1, kchunks_reserve=0),
          "staged_fill_stream": dict(kchunks_staged=1, kchunks_reserve=24), "pipeline": dict(kchunks_pipeline=1),
          "late_stitch": dict(kchunks_early_stitch=0), "rendezvous_late_stitch": dict(kchunks_staged=0, kchunks_early_stitch=0)}[variant]
    for (n, p, k, seed, top, chains) in ((3000, 2600, 60, 3, 160, 8), (2500, 700, 20, 11, 64, 4), (3000, 1500, 60, 5, 128, 3)):
        X, y, _, _ = synth.make_lm(n, p, k, seed=seed)
        seq = np.arange(1, top + 1)
        monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
        hooks(monkeypatch, kchunks_staged=1, kchunks_pipeline=0, kchunks_early_stitch=1)
        with gpu.Session(X, y) as s:
            want = s.sequential_path(seq, ic_type=3)
        monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
        hooks(monkeypatch, **hk)
        with gpu.Session(X, y) as s:
            for _ in range(3):
                _same_path(s.sequential_path(seq, ic_type=3), want)
            cnt = s.counters()
            assert cnt["kpath_chunked_paths"] == 3
            if variant == "pipeline":
                assert cnt["kpath_chunk_fills"] >= 3  # (the coarse chain's fills count: it runs on a context too)


@pytest.mark.parametrize("fam,chains", [("lm", 4), ("lm", 8), ("logistic", 6), ("poisson", 3), ("cox", 3), ("cox", 4)])
def test_shared_passes_return_the_path_of_the_chains_own_passes(gpu, monkeypatch, fam, chains):
    """Round 6: the chunk chains of the streaming forms hand their vector sets to ONE multi-chain launch per
    pass (k_xtv_mc / k_cox_score1p_mc) instead of streaming X once per chain.  The multi-chain kernels leave bitwise the
    sums of the single-chain kernels (tests/test_ops_gpu.py), so the same chains on passes of their own
    (kchunks_shared_pass=0, round 5) walk the same path -- and both return the single chain's candidates.  Also as two
    alternating groups.  (Coefficients are compared to 1e-8, not bitwise: which rows a chunk has stored when its
    predecessor's early stitch looks at them depends on timing, and a replaced row agrees with the chunk's own to 1e-9.)"""
    from helpers import hooks
    if fam == "lm":
        X, y, _, _ = synth.make_lm(3000, 900, 25, seed=21)
        kw, top = dict(score_mode=1), 96
    elif fam == "logistic":
        X, y, _, _ = synth.make_logistic(3000, 600, 12, seed=22)
        kw, top = dict(data_type=2, model_type=2), 60
    elif fam == "poisson":
        X, y, _, _ = synth.make_poisson(3000, 600, 12, seed=23)
        kw, top = dict(data_type=2, model_type=3), 48
    else:
        X, _, y, _, _ = synth.make_cox(3000, 600, 12, seed=24)
        kw, top = dict(data_type=3, model_type=4), 48
    seq = np.arange(1, top + 1)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y, **kw) as s:
        single = s.sequential_path(seq, ic_type=3)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", str(chains))
    outs = {}
    eq = 1000000000  # (equal chunk lengths in every variant: the same chunks start at the same levels)
    for name, hk in (("own", dict(kchunks_shared_pass=0, kchunks_len_div=eq)),
                     ("shared", dict(kchunks_shared_pass=1, kchunks_pass_groups=1, kchunks_len_div=eq)),
                     ("groups", dict(kchunks_shared_pass=1, kchunks_pass_groups=2, kchunks_len_div=eq))):
        hooks(monkeypatch, **hk)
        with gpu.Session(X, y, **kw) as s:
            s.enable_kernel_timing(True)
            outs[name] = s.sequential_path(seq, ic_type=3)
            again = s.sequential_path(seq, ic_type=3)
            cnt = s.counters()
            st = s.score_pass_stats()
        assert cnt["kpath_chunked_paths"] == 2 and cnt["kpath_chains_last_path"] == chains
        if name == "own":
            assert cnt["shared_pass_launches"] == 0
        else:
            # the chains' passes were shared launches that served more vector sets than there were launches (passes of
            # the session's own chain -- the tail after a stitch that gave up -- are counted in st, not here)
            assert st["launches"] > 0 and cnt["shared_pass_launches"] > 0 and cnt["shared_pass_partial_batches"] == 0
            assert cnt["shared_pass_chain_slots"] > cnt["shared_pass_launches"]
        _same_path(again, outs[name])
        _same_path(outs[name], outs["own"])
        _same_path(outs[name], single)


def test_fit_contexts_beyond_the_cap_of_own_queue_streams_fall_back_to_pool_streams(gpu, monkeypatch):
    """Streams with a hardware queue of their own (hipExtStreamCreateWithCUMask) are capped per process
    (bessx_session.cpp: OWN_QUEUE_CAP); beyond the cap a context gets an ordinary pool stream -- slower, same path."""
    from helpers import hooks
    X, y, _, _ = synth.make_lm(2500, 2304, 20, seed=11)
    seq = np.arange(1, 97)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "1")
    with gpu.Session(X, y) as s:
        want = s.sequential_path(seq, ic_type=3)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "4")
    for cap in (0, 2, 3):
        hooks(monkeypatch, ctx_streams_cap=cap)
        with gpu.Session(X, y) as s:
            _same_path(s.sequential_path(seq, ic_type=3), want)
            assert s.counters()["kpath_chains_last_path"] == 4


def test_streams_with_their_own_queue_are_recycled_across_sessions(gpu, monkeypatch):
    """A stream with a hardware queue of its own costs ~20 ms to create and destroy (tools/probe/cumask_stream_churn.hip);
    a session that ends leaves its streams idle for the next one of the process -- a drop-in call creates a session per
    call, and its 4-9 chain contexts would pay that every time."""
    X, y, _, _ = synth.make_lm(2500, 2304, 20, seed=11)
    seq = np.arange(1, 97)
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "4")
    created = []
    outs = []
    for _ in range(3):
        with gpu.Session(X, y) as s:
            outs.append(s.sequential_path(seq, ic_type=3))
            assert s.counters()["kpath_chains_last_path"] == 4
            created.append(s.counters()["own_queue_streams_created_by_the_process"])
    assert created[1] == created[0] and created[2] == created[0]  # sessions 2 and 3 created none
    _same_path(outs[1], outs[0])
    _same_path(outs[2], outs[0])
