"""Estimator classes with the reference's Python surface (names, keyword arguments, defaults,
error messages and result attributes of /root/reference python/bess/linear.py), implemented on
top of libbessx.so.  The argument marshalling that bess_base.fit performs before it calls
pywrap_bess (python/bess/linear.py:204-387) is restated here; all numerical work happens in
the HIP library through bess_amd.capi.pywrap_bess -- there is no NumPy solver in this file.
"""
import math

import numpy as np

from . import capi

_ALGORITHM_CODE = {"Pdas": 1, "GroupPdas": 2, "L0L2": 5}          # linear.py:144-152
_MODEL_CODE = {"Lm": 1, "Logistic": 2, "Poisson": 3, "Cox": 4}     # linear.py:154-164
_PATH_CODE = {"seq": 1, "pgs": 2}                                  # linear.py:166-190
_IC_CODE = {"aic": 1, "bic": 2, "gic": 3, "ebic": 4}               # linear.py:192-202
_DATA_TYPE = {"Lm": 1, "Logistic": 2, "Poisson": 2, "Cox": 3}      # linear.py:475,517,559,597


class bess_base:
    """Base estimator.  Parameters follow python/bess/linear.py:87-91:

    max_iter=20, exchange_num=0, is_warm_start=True, sequence=None, lambda_sequence=None, s_min=None,
    s_max=None, K_max=None, epsilon=0.0001, lambda_min=0, lambda_max=0, ic_type="ebic", is_cv=False, K=5,
    is_screening=False, screening_size=None, powell_path=1, always_select=[], tao=0.

    Attributes after fit(): beta, coef0, train_loss, ic.
    """

    def __init__(self, algorithm_type, model_type, path_type, max_iter=20, exchange_num=0, is_warm_start=True,
                 sequence=None, lambda_sequence=None, s_min=None, s_max=None, K_max=None, epsilon=0.0001,
                 lambda_min=0, lambda_max=0, ic_type="ebic", is_cv=False, K=5, is_screening=False,
                 screening_size=None, powell_path=1, always_select=[], tao=0.):
        self.algorithm_type, self.model_type, self.path_type = algorithm_type, model_type, path_type
        self.max_iter, self.exchange_num, self.is_warm_start = max_iter, exchange_num, is_warm_start
        self.sequence, self.lambda_sequence = sequence, lambda_sequence
        self.s_min, self.s_max, self.K_max, self.epsilon = s_min, s_max, K_max, epsilon
        self.lambda_min, self.lambda_max, self.n_lambda = lambda_min, lambda_max, 100
        self.ic_type, self.is_cv, self.K = ic_type, is_cv, K
        self.is_screening, self.screening_size, self.powell_path = is_screening, screening_size, powell_path
        self.always_select, self.tao = always_select, tao
        self.path_len = self.p = None
        self.data_type = _DATA_TYPE.get(model_type)
        self.beta = self.coef0 = self.train_loss = self.ic = None
        self._arg_check()

    def _arg_check(self):
        if self.algorithm_type not in _ALGORITHM_CODE:
            raise ValueError("algorithm_type should not be " + str(self.algorithm_type))
        if self.model_type not in _MODEL_CODE:
            raise ValueError("model_type should not be " + str(self.model_type))
        if self.path_type not in _PATH_CODE:
            raise ValueError("path_type should be \'seq\' or \'pgs\'")
        if self.ic_type not in _IC_CODE:
            raise ValueError("ic_type should be \"aic\", \"bic\", \"ebic\" or \"gic\"")
        self.algorithm_type_int = _ALGORITHM_CODE[self.algorithm_type]
        self.model_type_int = _MODEL_CODE[self.model_type]
        self.path_type_int = _PATH_CODE[self.path_type]
        self.ic_type_int = _IC_CODE[self.ic_type]

    @staticmethod
    def _group_starts(group, p):
        # python/bess/linear.py:238-253: first column of every (sorted) group label
        if group is None:
            raise ValueError("When you choose GroupPdas algorithm, the group information should be given")
        if len(group) != p:
            raise ValueError("The length of group should be equal to the number of variables")
        group = np.sort(np.asarray(group))
        return [int(np.argmax(group == g)) for g in sorted(set(group.tolist()))]

    def fit(self, X, y, is_weight=False, is_normal=True, weight=None, state=None, group=None):
        X, y = np.asarray(X), np.asarray(y)
        if np.isnan(X).any():
            raise ValueError("There is NAN value in X")
        if np.isnan(y).any():
            raise ValueError("There is NAN value in y")
        n, p = X.shape
        self.p = p
        g_index = self._group_starts(group, p) if self.algorithm_type_int == 2 else range(p)
        if self.model_type_int == 4:
            # Cox: rows by ascending time, response becomes the status column (linear.py:257-263)
            order = y[:, 0].argsort()
            X, y = X[order], y[order][:, 1].reshape(-1)
        if n != y.size:
            raise ValueError("X.shape(0) should be equal to y.size")
        if is_weight:
            if weight is None:
                raise ValueError("When you choose is_weight is True, the parameter weight should be given")
            if n != np.asarray(weight).size:
                raise ValueError("X.shape(0) should be equal to weight.size")
        else:
            weight = np.ones(n)
        if state is None:
            state = np.ones(n)
        if self.path_type_int == 1:
            if self.sequence is None:
                self.sequence = [i + 1 for i in range(min(p, int(n / np.log(n))))]
            if self.lambda_sequence is None:
                self.lambda_sequence = [0]
            self.s_min = self.s_max = self.K_max = 0
            self.lambda_min = self.lambda_max = 0
            self.path_len = int(len(self.sequence))
        else:
            self.sequence, self.lambda_sequence = [1], [0]
            self.s_min = 1 if self.s_min is None else self.s_min
            self.s_max = p if self.s_max is None else self.s_max
            if self.K_max is None:
                self.K_max = int(math.log(p, 2 / (math.sqrt(5) - 1)))
            self.lambda_min = 0 if self.lambda_min is None else self.lambda_min
            self.lambda_max = 0 if self.lambda_max is None else self.lambda_max
            self.path_len = self.K_max + 2
        if self.is_screening:
            if self.screening_size:
                if self.screening_size < max(self.sequence):
                    raise ValueError("screening size should be more than max(sequence).")
            else:
                self.screening_size = max(p, int(n / np.log(n)))
        else:
            self.screening_size = 1
        # libbessx holds the k x k work space of a session in HBM for at most 16382 active columns
        # (include/bessx.h: bessx_problem.max_sparsity); the reference has no such bound (any T0 <= p)
        top = max(self.sequence) if self.path_type_int == 1 else self.s_max
        gsz = int(np.max(np.diff(list(g_index) + [p])))
        if min(p, top * gsz) > capi.MAX_SPARSITY:
            raise ValueError("bess_amd: sparsity levels up to %d active columns are supported, this path asks for %d "
                             "(shorten `sequence` / lower `s_max`)" % (capi.MAX_SPARSITY, min(p, top * gsz)))
        result = capi.pywrap_bess(X, y, self.data_type, weight, is_normal, self.algorithm_type_int,
                                  self.model_type_int, self.max_iter, self.exchange_num, self.path_type_int,
                                  self.is_warm_start, self.ic_type_int, self.is_cv, self.K, g_index, state,
                                  self.sequence, self.lambda_sequence, self.s_min, self.s_max, self.K_max,
                                  self.epsilon, self.lambda_min, self.lambda_max, self.n_lambda, self.is_screening,
                                  self.screening_size, self.powell_path, self.always_select, self.tao, p, 1, 1, 1, 1,
                                  1, 1, p)
        self.beta, self.coef0, self.train_loss, self.ic = result[0], result[1], result[2], result[3]

    def predict(self, X):
        X = np.asarray(X)
        if X.shape[1] != self.p:
            raise ValueError("X.shape[1] should be " + str(self.p))
        eta = np.dot(X, self.beta) + np.ones(X.shape[0]) * self.coef0
        if self.model_type_int == 1:
            return eta
        if self.model_type_int == 2:
            label = np.zeros(eta.size)
            label[eta > 0] = 1
            e = np.exp(np.clip(eta, -25, 25))
            return {"Y": label, "pr": e / (e + 1)}
        if self.model_type_int == 3:
            return {"lam": np.exp(eta)}
        return None


def _make(name, algorithm_type, model_type):
    def __init__(self, max_iter=20, exchange_num=0, path_type="seq", is_warm_start=True, sequence=None,
                 lambda_sequence=None, s_min=None, s_max=None, K_max=None, epsilon=0.0001, lambda_min=None,
                 lambda_max=None, ic_type="ebic", is_cv=False, K=5, is_screening=False, screening_size=None,
                 powell_path=1, always_select=[], tao=0.):
        bess_base.__init__(self, algorithm_type=algorithm_type, model_type=model_type, path_type=path_type,
                           max_iter=max_iter, exchange_num=exchange_num, is_warm_start=is_warm_start,
                           sequence=sequence, lambda_sequence=lambda_sequence, s_min=s_min, s_max=s_max,
                           K_max=K_max, epsilon=epsilon, lambda_min=lambda_min, lambda_max=lambda_max,
                           ic_type=ic_type, is_cv=is_cv, K=K, is_screening=is_screening,
                           screening_size=screening_size, powell_path=powell_path, always_select=always_select,
                           tao=tao)

    doc = ("%s: %s best-subset selection for the %s model (counterpart of bess.linear.%s, "
           "python/bess/linear.py:433-921).\n" % (name, algorithm_type, model_type, name)) + bess_base.__doc__
    return type(name, (bess_base,), {"__init__": __init__, "__doc__": doc})


PdasLm = _make("PdasLm", "Pdas", "Lm")
PdasLogistic = _make("PdasLogistic", "Pdas", "Logistic")
PdasPoisson = _make("PdasPoisson", "Pdas", "Poisson")
PdasCox = _make("PdasCox", "Pdas", "Cox")
L0L2Lm = _make("L0L2Lm", "L0L2", "Lm")
L0L2Logistic = _make("L0L2Logistic", "L0L2", "Logistic")
L0L2Poisson = _make("L0L2Poisson", "L0L2", "Poisson")
L0L2Cox = _make("L0L2Cox", "L0L2", "Cox")
GroupPdasLm = _make("GroupPdasLm", "GroupPdas", "Lm")
GroupPdasLogistic = _make("GroupPdasLogistic", "GroupPdas", "Logistic")
GroupPdasPoisson = _make("GroupPdasPoisson", "GroupPdas", "Poisson")
GroupPdasCox = _make("GroupPdasCox", "GroupPdas", "Cox")
