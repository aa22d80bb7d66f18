"""Shared test helpers: comparison of PDAS traces (every iteration's active set of every fit)."""
import numpy as np


def rel_err(u, v):
    u, v = np.asarray(u, dtype=float), np.asarray(v, dtype=float)
    assert u.shape == v.shape, (u.shape, v.shape)
    if u.size == 0:
        return 0.0
    return float(np.max(np.abs(u - v) / np.maximum(np.abs(u), 1e-12)))


def assert_same_trace(got, want, beta_rtol=1e-6, what=""):
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
        np.testing.assert_allclose(a["coef0s"], b["coef0s"], rtol=beta_rtol, atol=beta_rtol * 1e-3)
    np.testing.assert_allclose(got["ic_calls"], want["ic_calls"], rtol=1e-9, atol=1e-9, err_msg=what + " ic values")
    np.testing.assert_allclose(got["loss_calls"], want["loss_calls"], rtol=1e-9, atol=1e-12, err_msg=what + " loss values")
