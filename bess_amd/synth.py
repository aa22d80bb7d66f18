"""Synthetic inputs for the BASELINE configs.

The coefficient recipe follows the reference's own generator
(/root/reference python/bess/gen_data.py:36-39 gaussian, :47-49 binomial, :78-100 cox):
m = 5*sqrt(2*log(p)/n); gaussian coefficients U(m, 100m), the other families U(2m, 10m);
random signs are added here so that the columns are not all positively associated.
Everything is drawn from numpy.random.Generator(PCG64(seed)) in a fixed order
(X, support, magnitudes, signs, noise/response) so that the oracle, the compiled
reference and the GPU path all see bit-identical inputs on any machine.
"""
import numpy as np

SEED_LM = 20200308       # configs[1] and configs[3]
SEED_LOGISTIC = 20200309  # configs[2]
SEED_CV = 20200310       # configs[3] fold permutation
SEED_COX = 20200311      # configs[4]
SEED_POISSON = 20200312  # the Poisson family at the logistic shape


def _design(rng, n, p):
    return rng.standard_normal((n, p))


def _coef(rng, n, p, k_true, lo, hi):
    support = np.sort(rng.choice(p, k_true, replace=False))
    m = 5.0 * np.sqrt(2.0 * np.log(p) / n)
    mag = rng.uniform(lo * m, hi * m, k_true)
    sign = rng.choice(np.array([-1.0, 1.0]), k_true)
    beta = np.zeros(p)
    beta[support] = mag * sign
    return support, beta


def make_lm(n=50000, p=10000, k_true=100, seed=SEED_LM):
    """configs[1]: LM, y = X beta + N(0,1).  Returns X (n x p, C order), y, support, beta."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = _design(rng, n, p)
    support, beta = _coef(rng, n, p, k_true, 1.0, 100.0)
    y = X[:, support] @ beta[support] + rng.standard_normal(n)
    return X, y, support, beta


def make_logistic(n=100000, p=5000, k_true=50, seed=SEED_LOGISTIC):
    """configs[2]: y ~ Bernoulli(sigmoid(clip(X beta, -30, 30)))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = _design(rng, n, p)
    support, beta = _coef(rng, n, p, k_true, 2.0, 10.0)
    eta = np.clip(X[:, support] @ beta[support], -30.0, 30.0)
    pr = np.exp(eta) / (1.0 + np.exp(eta))
    y = (rng.uniform(0.0, 1.0, n) < pr).astype(np.float64)
    return X, y, support, beta


def make_poisson(n=100000, p=5000, k_true=50, seed=SEED_POISSON):
    """Poisson at the shape of configs[2]: y ~ Poisson(exp(clip(X beta / 16, -30, 30))) -- the reference's
    generator scales the design by 1/16 for this family (python/bess/gen_data.py:60-73); here the design stays
    N(0,1) and the scale sits in the linear predictor, which is the same model after column normalisation."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = _design(rng, n, p)
    support, beta = _coef(rng, n, p, k_true, 2.0, 10.0)
    beta = beta / 16.0
    eta = np.clip(X[:, support] @ beta[support], -30.0, 30.0)
    y = rng.poisson(np.exp(eta)).astype(np.float64)
    return X, y, support, beta


def make_cv_folds(n, K=5, seed=SEED_CV):
    """configs[3]: fold id per row, same shape as Metric::set_cv_train_test_mask builds
    (/root/reference src/Metric.h:66-78): a permutation cut into K contiguous chunks of
    floor(n/K) rows, the last chunk taking the remainder."""
    perm = np.random.Generator(np.random.PCG64(seed)).permutation(n)
    size = n // K
    fold = np.empty(n, dtype=np.int32)
    for k in range(K):
        chunk = perm[k * size:(k + 1) * size] if k < K - 1 else perm[(K - 1) * size:]
        fold[chunk] = k
    return fold


def make_cox(n=200000, p=20000, k_true=75, seed=SEED_COX, scal=10.0):
    """configs[4]: time = (-log U / exp(X beta))^(1/scal), censoring time c*U with c chosen
    for about 50% events; rows are returned sorted by time (as bess_base.fit does,
    python/bess/linear.py:257-263) together with the status vector."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = _design(rng, n, p)
    support, beta = _coef(rng, n, p, k_true, 2.0, 10.0)
    eta = X[:, support] @ beta[support]
    time = np.power(-np.log(rng.uniform(0.0, 1.0, n)) / np.exp(eta), 1.0 / scal)
    u = rng.uniform(0.0, 1.0, n)
    # c such that P(time < c*U) is about one half: bisection on the empirical rate
    lo, hi = 0.0, 10.0 * float(np.max(time))
    for _ in range(60):
        c = 0.5 * (lo + hi)
        if np.mean(time < c * u) < 0.5:
            lo = c
        else:
            hi = c
    ctime = c * u
    status = (time < ctime).astype(np.float64)
    obs = np.minimum(time, ctime)
    order = np.argsort(obs, kind="stable")
    return np.ascontiguousarray(X[order]), obs[order], status[order], support, beta
