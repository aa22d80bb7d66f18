"""Shared test helpers: comparison of PDAS traces (every iteration's active set of every fit)."""
import numpy as np


def hooks(monkeypatch, **kw):
    """Add test hooks of libbessx (BESSX_TEST_HOOKS = "name=value,..."; read when a session is created): they force the
    fallback paths the library keeps anyway, so that a test can compare them with the defaults."""
    import os
    cur = dict(kv.split("=", 1) for kv in os.environ.get("BESSX_TEST_HOOKS", "").split(",") if "=" in kv)
    cur.update({k: str(v) for k, v in kw.items()})
    monkeypatch.setenv("BESSX_TEST_HOOKS", ",".join("%s=%s" % kv for kv in cur.items()))


def rel_err(u, v):
    u, v = np.asarray(u, dtype=float), np.asarray(v, dtype=float)
    assert u.shape == v.shape, (u.shape, v.shape)
    if u.size == 0:
        return 0.0
    return float(np.max(np.abs(u - v) / np.maximum(np.abs(u), 1e-12)))


def assert_same_trace(got, want, beta_rtol=1e-6, what="", ic_atol=1e-9):
    """got / want: dicts with 'fits' (list of {T0, train_n, iters, betas, coef0s}), 'ic_calls', 'loss_calls'.
    Active sets must be identical (bit-exact indices) at EVERY PDAS iteration; coefficients within rtol."""
    assert len(got["fits"]) == len(want["fits"]), "%s: number of fits %d != %d" % (what, len(got["fits"]), len(want["fits"]))
    for fi, (a, b) in enumerate(zip(got["fits"], want["fits"])):
        assert a["T0"] == b["T0"] and a["train_n"] == b["train_n"], "%s: fit %d header" % (what, fi)
        assert len(a["iters"]) == len(b["iters"]), "%s: fit %d (T0=%d) took %d PDAS iterations, expected %d" % (
            what, fi, a["T0"], len(a["iters"]), len(b["iters"]))
        for it, (x, y) in enumerate(zip(a["iters"], b["iters"])):
            assert np.array_equal(x, y), "%s: fit %d (T0=%d) iteration %d support differs:\n%s\n%s" % (
                what, fi, a["T0"], it + 1, x, y)
        for it, (x, y) in enumerate(zip(a["betas"], b["betas"])):
            scale = max(np.max(np.abs(y)), 1e-300)
            assert np.max(np.abs(x - y)) <= beta_rtol * scale, "%s: fit %d iteration %d beta differs by %g" % (
                what, fi, it + 1, np.max(np.abs(x - y)) / scale)
        # intercepts: relative to their own size or to the size of the fit's coefficients, whichever is larger
        bscale = max([float(np.max(np.abs(v))) for v in b["betas"] if len(v)] + [1e-3])
        np.testing.assert_allclose(a["coef0s"], b["coef0s"], rtol=beta_rtol, atol=beta_rtol * bscale)
    np.testing.assert_allclose(got["ic_calls"], want["ic_calls"], rtol=1e-9, atol=ic_atol, err_msg=what + " ic values")
    np.testing.assert_allclose(got["loss_calls"], want["loss_calls"], rtol=1e-9, atol=1e-12, err_msg=what + " loss values")


def golden_final_models(g):
    """Per fit of a golden file of tests/golden/make_fullsize_ref.py: (support, coefficients, intercept) of its LAST
    PDAS iteration -- the candidate's model, in the reference's internal (normalised) scale."""
    out, off, it_off = [], 0, 0
    for it, t in zip(g["fit_iters"], g["fit_T0"]):
        last = off + (int(it) - 1) * int(t)
        out.append((g["A_flat"][last:last + t], g["beta_flat"][last:last + t], float(g["coef0_flat"][it_off + int(it) - 1])))
        off += int(it) * int(t)
        it_off += int(it)
    return out


def assert_untraced_path_matches_golden(out, g, X, data_type, what, beta_rtol=1e-6, metric_rtol=1e-9):
    """An UNTRACED path (what the bench times: chained fits, chunk chains, fused launches -- tracing switches those off)
    against the compiled reference's golden: every candidate's final support bit-exact, its PDAS iteration count,
    coefficients to beta_rtol, loss / criterion to metric_rtol, and the selected model.  The golden's coefficients are
    the reference's internal ones; the path returns them de-normalised (src/path.cpp:76-110), so the golden is brought
    to the caller's scale with a NumPy restatement of Normalize (src/normalize.cpp:20-85) on the columns involved."""
    models = golden_final_models(g)
    nfit = len(models)
    truncated = int(g["truncated"]) if "truncated" in g.files else 0
    label = "%s (%s%d candidates of the compiled reference)" % (what, "first " if truncated else "", nfit)
    assert out["n_candidates"] >= nfit, label
    assert list(out["cand_T0"][:nfit]) == list(g["fit_T0"]), label + ": order of sparsity levels"
    assert list(out["cand_iters"][:nfit]) == list(g["fit_iters"]), label + ": PDAS iterations per candidate"
    n = X.shape[0]
    cols = np.unique(np.concatenate([m[0] for m in models]))
    Xc = np.asarray(X[:, cols], dtype=np.float64)
    mean = Xc.mean(axis=0) if data_type in (1, 2) else np.zeros(cols.size)
    Xc = Xc - mean
    norm = np.sqrt((Xc * Xc).sum(axis=0))
    pos = {int(c): i for i, c in enumerate(cols)}
    bscale = max(float(np.max(np.abs(m[1]))) for m in models)
    for i, (A, b, c0) in enumerate(models):
        t = len(A)
        assert np.array_equal(out["cand_support"][i][:t], A), "%s: candidate %d (T0=%d) support" % (label, i, t)
        assert np.all(out["cand_support"][i][t:] == -1)
        ix = np.array([pos[int(c)] for c in A])
        got_internal = out["cand_beta"][i][:t] * norm[ix] / np.sqrt(n)
        np.testing.assert_allclose(got_internal, b, rtol=beta_rtol, atol=beta_rtol * bscale,
                                   err_msg="%s: candidate %d coefficients" % (label, i))
        if data_type == 2:  # coef0 - beta . x_mean
            want_c0 = c0 - float(np.dot(out["cand_beta"][i][:t], mean[ix]))
            np.testing.assert_allclose(out["cand_coef0"][i], want_c0, rtol=beta_rtol, atol=beta_rtol * max(bscale, 1.0),
                                       err_msg="%s: candidate %d intercept" % (label, i))
    np.testing.assert_allclose(out["cand_ic"][:nfit], g["ic_calls"][:nfit], rtol=metric_rtol, err_msg=label + " ic")
    np.testing.assert_allclose(out["cand_train_loss"][:nfit], g["loss_calls"][:nfit], rtol=metric_rtol,
                               err_msg=label + " loss")
    if not truncated:
        assert np.array_equal(np.nonzero(out["beta"])[0], g["best_beta_idx"]), label + ": selected model"
        np.testing.assert_allclose(out["beta"][g["best_beta_idx"]], g["best_beta_val"], rtol=beta_rtol)
    return nfit


class NumpyLmSession:
    """CPU stand-in for bess_amd.capi.Session in the multi-rank HOST-LOGIC tests (gloo): the LM fit primitive
    (Algorithm::fit with GroupPdasLm::get_A / primary_model_fit, src/Algorithm.h:113-171, :1097-1135, singleton
    groups, unit weights) in NumPy on the normalised data (Normalize, src/normalize.cpp:20-46).  Test
    infrastructure only; the results of the paths built on it are checked against the pinned plain-C oracle."""

    def __init__(self, X, y, fold_id, K, max_iter=20):
        X = np.array(X, dtype=np.float64)
        self.n, self.p = X.shape
        self.x_mean = X.mean(axis=0)
        self.y_mean = float(np.mean(y))
        X = X - self.x_mean
        self.x_norm = np.sqrt((X * X).sum(axis=0))
        self.X = np.sqrt(self.n) * X / self.x_norm
        self.y = np.asarray(y, dtype=np.float64) - self.y_mean
        self.fold_id, self.K, self.max_iter = np.asarray(fold_id), K, max_iter

    def normalization(self):
        return self.x_mean, self.x_norm, self.y_mean

    def fit(self, T0, lam=0.0, fold=-1, init_idx=(), init_val=(), init_coef0=0.0):
        train = np.ones(self.n, bool) if fold < 0 else self.fold_id != fold
        Xt, yt = self.X[train], self.y[train]
        nt = Xt.shape[0]
        phi = np.sqrt(2 * lam + (Xt * Xt).sum(axis=0) / nt)
        beta = np.zeros(self.p)
        beta[np.asarray(init_idx, dtype=int)] = init_val
        seen = [np.zeros(T0, dtype=np.int64)]
        for l in range(1, self.max_iter + 1):
            d = Xt.T @ (yt - Xt @ beta - init_coef0) / nt - 2 * lam * beta
            bd = (phi * beta + d * (1.0 / phi)) ** 2
            A = np.sort(np.lexsort((np.arange(self.p), -bd))[:T0])
            XA = Xt[:, A]
            bA = np.linalg.solve(XA.T @ XA + lam * np.eye(T0), XA.T @ yt)
            beta = np.zeros(self.p)
            beta[A] = bA
            stop = any(np.array_equal(A, a) for a in seen)
            seen.append(A)
            if stop:
                break
        res = self.y - self.X @ beta
        test = ~train
        return {"support": A.astype(np.int32), "beta": bA, "coef0": float(init_coef0), "iters": l,
                "train_loss": float(res @ res) / self.n,
                "test_loss": float(res[test] @ res[test]) / (2 * max(int(test.sum()), 1)) if fold >= 0 else 0.0}

    def sequential_path_chain(self, sequence, ic_type=3, init_idx=(), init_val=(), init_coef0=0.0, keep_caches=False,
                              stop_support=None, stop_beta=None, stop_rtol=1e-9, lead_levels=()):
        """Stand-in for capi.Session.sequential_path_chain (one link of a warm-start chain, src/path.cpp:60-64), GIC."""
        seq = [int(v) for v in sequence]
        W = max(seq)
        bi, bv = np.asarray(init_idx, dtype=np.int32), np.asarray(init_val, dtype=np.float64)
        for T0 in lead_levels:  # bessx_path_chain.lead_levels: a coarse warm-start chain in front of the link
            r = self.fit(int(T0), 0.0, -1, bi, bv, init_coef0)
            bi, bv = r["support"], r["beta"]
        out = {"cand_T0": [], "cand_iters": [], "cand_train_loss": [], "cand_ic": [], "cand_coef0": [],
               "cand_support": [], "cand_beta": []}
        stopped = -1
        for i, T0 in enumerate(seq):
            r = self.fit(T0, 0.0, -1, bi, bv, init_coef0)
            bi, bv = r["support"], r["beta"]
            sup = np.full(W, -1, dtype=np.int32)
            sup[:T0] = r["support"]
            b = np.zeros(W)
            b[:T0] = np.sqrt(self.n) * r["beta"] / self.x_norm[r["support"]]
            out["cand_T0"].append(T0)
            out["cand_iters"].append(r["iters"])
            out["cand_train_loss"].append(r["train_loss"])
            out["cand_ic"].append(self.n * np.log(r["train_loss"]) + np.log(self.p) * np.log(np.log(self.n)) * T0)
            out["cand_coef0"].append(self.y_mean - float(b[:T0] @ self.x_mean[r["support"]]))
            out["cand_support"].append(sup)
            out["cand_beta"].append(b)
            if stop_support is not None and i < len(stop_support):
                want = np.asarray(stop_support[i])
                want = want[want >= 0]
                if np.array_equal(want, r["support"]) and (stop_beta is None or np.allclose(
                        np.asarray(stop_beta[i])[:T0], b[:T0], rtol=stop_rtol, atol=0)):
                    stopped = i
                    break
        res = {k: np.asarray(v) for k, v in out.items()}
        res.update({"n_candidates": len(out["cand_T0"]), "stopped_at": stopped, "last_idx": bi.copy(),
                    "last_val": bv.copy(), "last_coef0": float(init_coef0)})
        return res


class ThreadComm:
    """In-process stand-in for torch.distributed in tests that run N 'ranks' as N threads of one process (each with a
    session of its own on the one GPU; ctypes releases the GIL inside the library): all_gather over a barrier."""

    def __init__(self, world):
        import threading
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)

    def view(self, rank):
        parent = self

        class _View:
            def all_gather(self, mine, world):
                parent.slots[rank] = np.array(mine, dtype=np.float64, copy=True)
                parent.barrier.wait()
                got = [parent.slots[r].copy() for r in range(parent.world)]
                parent.barrier.wait()
                return got
        return _View()


def run_ranks(world, fn):
    """fn(rank, comm) on `world` threads; returns the list of results, re-raises the first exception."""
    import threading
    tc = ThreadComm(world)
    res, err = [None] * world, [None] * world

    def body(r):
        try:
            res[r] = fn(r, tc.view(r))
        except BaseException as e:  # noqa: BLE001 (a failed rank must not leave the others in the barrier for ever)
            err[r] = e
            tc.barrier.abort()
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, __import__("threading").BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return res
