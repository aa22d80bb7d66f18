"""Randomised differential test: GPU paths against the plain-C oracle (test infrastructure) on many small random
problems and option combinations -- families, weights, always_select, lambda grids, CV, golden section, groups,
both LM score-pass forms.  Every PDAS iteration's active set must match; coefficients to 1e-6.
  python tests/fuzz_parity.py [cases] [seed]        (FUZZ_SCALE=medium: n up to 30 000, p up to 4 000)
tests/test_fuzz_gpu.py runs a fixed slice of it in the GPU suite."""
import sys
import time

import numpy as np

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bess_amd import capi, synth  # noqa: E402
from helpers import assert_same_trace  # noqa: E402
from oracle import port_ctypes as P  # noqa: E402
from test_lm_gpu import run_gpu  # noqa: E402



def run(cases=100, seed=1, medium=False, verbose=True):
    """Returns the number of failing cases."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    fails = 0
    for c in range(cases):
        fam = rng.choice(["lm", "lm", "lm", "logit", "poisson", "cox"])
        if medium:  # FUZZ_SCALE=medium: shapes that take the multi-block / multi-slab / two-level paths
            n = int(rng.integers(3000, 30000))
            p = int(rng.integers(300, 4000))
        else:
            n = int(rng.integers(80, 1500))
            p = int(rng.integers(8, 400))
        kt = int(min(max(2, p // 6), rng.integers(2, 12)))
        seed = int(rng.integers(1, 1 << 30))
        kw = {}
        if fam == "lm":
            X, y, _, _ = synth.make_lm(n, p, kt, seed=seed)
            kw["score_mode"] = int(rng.integers(1, 3))
        elif fam == "logit":
            X, y, _, _ = synth.make_logistic(n, p, kt, seed=seed)
            kw.update(data_type=2, model_type=2)
        elif fam == "poisson":
            X = np.random.default_rng(seed).standard_normal((n, p))
            b = np.zeros(p)
            b[:kt] = np.random.default_rng(seed + 1).uniform(-0.4, 0.4, kt)
            y = np.random.default_rng(seed + 2).poisson(np.exp(np.clip(X @ b, -4, 3))).astype(float)
            kw.update(data_type=2, model_type=3)
        else:
            X, _, y, _, _ = synth.make_cox(n, p, kt, seed=seed)
            kw.update(data_type=3, model_type=4)
        kmax = int(min(p, n // 4, rng.integers(3, 60 if medium else 25)))
        if rng.random() < 0.25:
            kw["weight"] = rng.uniform(0.5, 2.0, n)
        if rng.random() < 0.2:
            kw["is_warm_start"] = False
        if rng.random() < 0.15:
            kw["max_iter"] = int(rng.integers(2, 6))
        if rng.random() < 0.2 and fam != "cox":
            kw["is_normal"] = bool(rng.random() < 0.5)
        mode = rng.choice(["seq", "seq", "gs", "lam", "cv", "grp", "powell", "scr"])
        if mode == "scr" and fam == "poisson":
            mode = "seq"  # (Poisson screening is refused: undefined behaviour in the reference)
        if mode == "gs":
            kw.update(path_type=2, s_min=1, s_max=kmax)
        elif mode == "lam":
            kw.update(sequence=np.arange(1, max(3, kmax // 2)),
                      lambda_seq=sorted(rng.uniform(0, 0.1 if fam == "cox" else 0.2, int(rng.integers(2, 4)))))
        elif mode == "cv":
            K = int(rng.integers(2, 6))
            kw.update(is_cv=True, K=K, cv_fold_id=synth.make_cv_folds(n, K, seed=seed))
            if rng.random() < 0.5:
                kw["sequence"] = np.arange(1, max(3, kmax // 2))
            else:
                kw.update(path_type=2, s_min=1, s_max=max(3, kmax))
        elif mode == "grp" and p >= 12 and not (medium and fam == "cox"):  # (the oracle's Cox group branch is O(n^2))
            cuts = np.sort(rng.choice(np.arange(1, p), min(p - 1, int(rng.integers(3, max(4, p // 3)))), replace=False))
            gi = np.concatenate([[0], cuts]).astype(np.int32)
            # (groups wider than 16 columns take the tiled moment / Cholesky-score kernels; Cox keeps whole groups in
            # a 256-column panel)
            gs_ = np.sort(np.diff(np.append(gi, p)))[::-1]
            if gs_[0] > (64 if fam == "cox" else 200) or int(np.sum(gs_[:5])) > n // 3:
                # (also: the widest selectable groups together must stay well below n.  A restricted fit with more
                # columns than independent rows has no reproducible reference value -- the reference's pivoted QR and
                # the oracle's LDL^T both return quotients of rounding errors there -- so a DIFFERENTIAL run cannot use
                # such draws; the GPU path returns the basic solution of its pivoted solve instead of an error:
                # tests/test_rank_deficient_gpu.py)
                gi = np.arange(0, p, 3).astype(np.int32)
            if rng.random() < 0.4:
                # groups of ONE width: untraced LM fits then expand the selected groups on the device (k_group_expand)
                gs_u = int(rng.integers(2, 6))
                p = (p // gs_u) * gs_u
                X = np.ascontiguousarray(X[:, :p])
                gi = np.arange(0, p, gs_u).astype(np.int32)
            kw.update(algorithm_type=2, g_index=gi, sequence=np.arange(1, min(len(gi), 6)))
            kw.pop("score_mode", None)
        elif mode == "powell":
            # (Cox: the reference adds 2 lambda with the sign that SUBTRACTS from the information matrix; beyond
            # lambda ~ 0.1 the Newton systems turn indefinite and even the oracle and the compiled reference part ways)
            kw.update(algorithm_type=5, path_type=3, s_min=1, s_max=max(2, kmax), lambda_min=0.001,
                      lambda_max=float(rng.uniform(0.02, 0.1) if fam == "cox" else rng.uniform(0.05, 0.5)),
                      nlambda=int(rng.integers(4, 10)),
                      powell_path=int(rng.integers(1, 3)))
            kw.pop("is_normal", None)
        elif mode == "scr" and p >= 20 and max(kmax + 2, 8) < p:
            kw["screening_size"] = int(rng.integers(max(kmax + 2, 8), p))
            kw["sequence"] = np.arange(1, kmax + 1)
            kw.pop("is_normal", None)
        else:
            kw["sequence"] = np.arange(1, kmax + 1)
        if mode not in ("grp", "powell", "scr") and rng.random() < 0.2 and kmax >= 3:
            al = sorted(rng.choice(p, 2, replace=False).tolist())
            kw["always_select"] = al
            if "sequence" in kw:
                kw["sequence"] = np.arange(3, max(4, kmax))
            else:
                kw["s_min"] = 3
                kw["s_max"] = max(4, kmax)
        if mode in ("seq", "gs", "lam", "grp", "powell", "scr") and rng.random() < 0.2:
            # cross-validation on top of any path type (folds of a random size; the fold fits of LM covariance-form
            # sessions run side by side when untraced)
            K = int(rng.integers(2, 7))
            kw.update(is_cv=True, K=K, cv_fold_id=synth.make_cv_folds(n, K, seed=seed + 1))
        kw["ic_type"] = int(rng.integers(1, 5))
        okw = {k: v for k, v in kw.items() if k != "score_mode"}
        if medium:
            print("case %d: %s n=%d p=%d mode=%s kmax=%d ..." % (c, fam, n, p, mode, kmax), flush=True)
        try:
            if "screening_size" in kw:
                okw.pop("screening_size")
                want = P.trace_screened(X, y, kw["screening_size"], **okw)
                want["beta"] = want["beta_screened"]  # the traces are in the screened numbering on both sides
            else:
                want = P.trace(X, y, **okw)
            got = run_gpu(capi, X, y, kw)
            if "screening_size" in kw:
                assert np.array_equal(got["screening_A"], want["screening_A"]), "kept columns differ"
            # LM information criteria are n log(loss) + ...: a loss that agrees to 1e-10 relative moves them by 1e-10 n
            # (Cox with a ridge that can outweigh the information matrix: an indefinite Newton system, coefficients to 1e-4)
            loose = fam == "cox" and mode in ("lam", "powell")
            if loose:
                assert len(got["trace"]["fits"]) == len(want["fits"])
                for a, b in zip(got["trace"]["fits"], want["fits"]):
                    assert len(a["iters"]) == len(b["iters"]) and all(np.array_equal(u, v) for u, v in zip(a["iters"], b["iters"]))
                    for u, v in zip(a["betas"], b["betas"]):
                        assert np.max(np.abs(u - v)) <= 1e-4 * max(np.max(np.abs(v)), 1e-300)
            elif mode == "powell" and (fam != "lm" or kw.get("is_cv")):
                # a Powell line search over a plateau of (nearly) equal criteria can take another route when the
                # IRLS-converged losses -- or, under CV, the test losses of two evaluations of the SAME candidate, which
                # golden_section_search compares with `<` (src/path.cpp:735-750) -- differ in their last digits: then
                # only the end result is compared
                try:
                    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="case %d" % c, ic_atol=1e-9 * n)
                except AssertionError:
                    sup = np.nonzero(want["beta"])[0]
                    if not np.array_equal(np.nonzero(got["beta"])[0], sup):
                        # the other side of an exact tie (the same candidate evaluated twice, criteria equal to 1e-15):
                        # golden_section_search then returns either the fresh model or -- its stale-model quirk,
                        # src/path.cpp:762-766 -- the one it stored for a point it has since moved.  Same criterion.
                        np.testing.assert_allclose(got["ic"], want["ic"], rtol=1e-12, err_msg="Powell end result: support")
                    elif not kw.get("is_cv"):
                        # (under CV the returned coefficients are the LAST FOLD's fit of whichever of two evaluations
                        # of the best candidate won a `<` between criteria equal to 1e-15, src/path.cpp:314-319 -- fold
                        # fits from different warm starts need not share a support)
                        np.testing.assert_allclose(got["beta"][sup], want["beta"][sup], rtol=1e-4, atol=1e-8)
                    np.testing.assert_allclose(got["ic"], want["ic"], rtol=1e-6)
            else:
                assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="case %d" % c, ic_atol=1e-9 * n)
            if not (mode == "powell" and (fam != "lm" or kw.get("is_cv"))):
                # the same call without the trace: the fast paths (fits chained on the device, fused selection + solve
                # launches, the fold fits of a CV evaluation side by side) must walk the path the traced run walked
                fast = run_gpu(capi, X, y, kw, trace=False)
                assert fast["n_fits"] == got["n_fits"] and fast["n_pdas_iters"] == got["n_pdas_iters"], "untraced: fit counts"
                assert np.array_equal(fast["cand_support"], got["cand_support"]), "untraced: supports"
                assert np.array_equal(fast["cand_iters"], got["cand_iters"]), "untraced: iterations"
                tol = 1e-4 if loose else 1e-8
                np.testing.assert_allclose(fast["cand_ic"], got["cand_ic"], rtol=tol, atol=1e-9 * n)
                np.testing.assert_allclose(fast["beta"], got["beta"], rtol=max(tol, 1e-7), atol=1e-12)
            if mode == "seq" and not kw.get("is_cv") and "always_select" not in kw and len(kw["sequence"]) >= 4:
                # the same path as two LINKS of one warm-start chain (bessx_session_sequential_path_chain): the second
                # link starts from the model the first hands over, on its caches or cold
                cut = int(rng.integers(1, len(kw["sequence"]) - 1))
                with capi.Session(X, y, weight=kw.get("weight"), data_type=kw.get("data_type", 1),
                                  is_normal=kw.get("is_normal", True), model_type=kw.get("model_type", 1),
                                  max_iter=kw.get("max_iter", 20), is_warm_start=kw.get("is_warm_start", True),
                                  score_mode=kw.get("score_mode", 0)) as sc:
                    head = sc.sequential_path_chain(kw["sequence"][:cut], ic_type=kw["ic_type"])
                    tail = sc.sequential_path_chain(kw["sequence"][cut:], ic_type=kw["ic_type"], init_idx=head["last_idx"],
                                                    init_val=head["last_val"], init_coef0=head["last_coef0"],
                                                    keep_caches=bool(rng.random() < 0.5))
                w = fast["cand_support"].shape[1]
                both = np.full((len(kw["sequence"]), w), -1, dtype=np.int32)
                both[:cut, :head["cand_support"].shape[1]] = head["cand_support"]
                both[cut:, :tail["cand_support"].shape[1]] = tail["cand_support"]
                assert np.array_equal(both, fast["cand_support"]), "chain links: supports"
                np.testing.assert_allclose(np.concatenate([head["cand_ic"], tail["cand_ic"]]), fast["cand_ic"],
                                           rtol=1e-4 if loose else 1e-8, atol=1e-9 * n, err_msg="chain links: criteria")
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("CASE %d FAILED: fam=%s n=%d p=%d seed=%d mode=%s kw=%r\n  %s" % (
                c, fam, n, p, seed, mode, {k: (v if np.size(v) < 8 else "...") for k, v in kw.items()}, str(e)[:300]), flush=True)
        if c % (5 if medium else 25) == (4 if medium else 24):
            print("%d cases, %d failures, %.0f s" % (c + 1, fails, time.time() - t0), flush=True)

    if verbose:
        print("fuzz done: %d cases, %d failures" % (cases, fails))
    return fails


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(n_cases, sd, os.environ.get("FUZZ_SCALE") == "medium") else 0)
