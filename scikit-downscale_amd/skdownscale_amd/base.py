"""Input validation shared by the estimators (mirrors base.py:12-136 of the reference).

Semantics kept: finite check (ValueError with sklearn's wording), fabricated monthly index + the
``UserWarning`` for index-less input (base.py:21-24, 32-35), ``n_features_in_``.
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
from sklearn.base import BaseEstimator


def _finite_error(name, a):
    a = np.asarray(a, dtype=np.float64)
    if np.isnan(a).any():
        return ValueError(f"Input {name} contains NaN.")
    return ValueError(f"Input {name} contains infinity or a value too large for dtype('float64').")


def _to_2d(a, name):
    a = np.asarray(a.values if isinstance(a, (pd.DataFrame, pd.Series)) else a, dtype=np.float64)
    if a.ndim == 1:
        if name == "X":
            raise ValueError(
                f"Expected 2D array, got 1D array instead:\narray={a}.\nReshape your data either using "
                "array.reshape(-1, 1) if your data has a single feature or array.reshape(1, -1) if it contains a single sample.")
        a = a.reshape(-1, 1)
    if a.ndim != 2:
        raise ValueError(f"Found array with dim {a.ndim}. Expected <= 2.")
    return a


class TimeSynchronousDownscaler(BaseEstimator):
    _timestep = "M"

    def _index_of(self, obj, n, fit):
        if isinstance(obj, (pd.DataFrame, pd.Series)):
            return obj.index, False
        # base.py:21-24 / 32-35: index-less input -> fabricated monthly index from 1950
        freq = "MS" if fit else self._timestep
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", FutureWarning)
            return pd.date_range(periods=n, start="1950", freq=freq), True

    def _check_X_y(self, X, y):
        """base.py:13-25.  Returns (X2d, y2d, index)."""
        both_df = isinstance(X, pd.DataFrame) and isinstance(y, pd.DataFrame)
        if both_df:
            pd.testing.assert_index_equal(X.index, y.index)
        X2, y2 = _to_2d(X, "X"), _to_2d(y, "y")
        if len(X2) != len(y2):
            raise ValueError(f"Found input variables with inconsistent numbers of samples: [{len(X2)}, {len(y2)}]")
        if not np.isfinite(X2).all():
            raise _finite_error("X", X2)
        if not np.isfinite(y2).all():
            raise _finite_error("y", y2)
        if both_df:
            index = X.index
        else:
            warnings.warn("X and y do not have pandas DateTimeIndexes, making one up...")
            index, _ = self._index_of(None, len(X2), fit=True)
        self.n_features_in_ = X2.shape[1]
        return X2, y2, index

    def _check_array(self, X, reset_features=False):
        """base.py:27-36.  Returns (X2d, index)."""
        X2 = _to_2d(X, "X")
        if not np.isfinite(X2).all():
            raise _finite_error("X", X2)
        if isinstance(X, pd.DataFrame):
            index = X.index
        else:
            warnings.warn("array does not have a pandas DateTimeIndex, making one up...")
            index, _ = self._index_of(None, len(X2), fit=False)
        if reset_features:
            self.n_features_in_ = X2.shape[1]
        return X2, index

# sklearn keyword arguments the reference forwards to LinearRegression / LogisticRegression (gard.py:136-150, 257-271, 389-402;
# trend.py:40-42).  The engine computes ONE model -- ordinary least squares with an intercept; the exact minimiser of the
# L2-penalised logistic objective with C = 1 --, so a value that asks for that model, or an option that does not change the
# fitted model (threads, copies, verbosity, solver, stopping rule: the engine's minimiser is the converged one), is accepted
# and anything else refused.
LINEAR_NEUTRAL = {"fit_intercept": (True,), "positive": (False,), "copy_X": None, "n_jobs": None, "tol": None}
LOGISTIC_NEUTRAL = {"penalty": ("l2",), "C": (1, 1.0), "fit_intercept": (True,), "dual": (False,), "intercept_scaling": (1, 1.0),
                    "class_weight": (None,), "l1_ratio": (None,), "multi_class": ("auto", "deprecated"), "solver": None, "tol": None,
                    "max_iter": None, "n_jobs": None, "verbose": None, "warm_start": None, "random_state": None}


def _same_value(v, a):
    if a is None or isinstance(a, (bool, str)):
        return v is a or (isinstance(a, str) and isinstance(v, str) and v == a)
    return isinstance(v, (int, float)) and not isinstance(v, bool) and v == a  # (C=1 and C=1.0 are the same request)


def check_sklearn_kwargs(kwargs, neutral, what, model):
    """raise NotImplementedError for the first entry of ``kwargs`` that would change the fitted model"""
    for k, v in (kwargs or {}).items():
        if k not in neutral:
            raise NotImplementedError(f"{what}={{{k!r}: {v!r}}} is not supported on the HIP engine ({model})")
        allowed = neutral[k]
        if allowed is not None and not any(_same_value(v, a) for a in allowed):
            raise NotImplementedError(f"{what}={{{k!r}: {v!r}}} is not supported on the HIP engine ({model})")
