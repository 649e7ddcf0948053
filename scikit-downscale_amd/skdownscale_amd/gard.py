"""GARD analog estimators with the reference's surface, computed by the HIP engine.

Mirrors ``skdownscale/pointwise_models/gard.py``: ``AnalogBase`` (55-98), ``AnalogRegression``
(101-224), ``PureAnalog`` (227-364) and ``PureRegression`` (367-504), with and without ``thresh``.  The KD-tree of the reference is
replaced by batched exact nearest-neighbour search in ``csrc/sd_analog.hip`` (neighbours ordered by
(squared distance, training index); identical to ``KDTree.query`` on tie-free data).
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
from sklearn.base import BaseEstimator, RegressorMixin
from sklearn.exceptions import NotFittedError

from . import _lib
from .base import LINEAR_NEUTRAL, LOGISTIC_NEUTRAL, _finite_error, check_sklearn_kwargs
from .engine import default_context

KIND_CODES = {"best_analog": _lib.ANALOG_BEST, "sample_analogs": _lib.ANALOG_SAMPLE,
              "weight_analogs": _lib.ANALOG_WEIGHT, "mean_analogs": _lib.ANALOG_MEAN}
OUTPUT_NAMES = ["pred", "exceedance_prob", "prediction_error"]


class AnalogGridModel:
    """Batched analog model over the cell axis: X [T,F,C], y [T,C], Xq [Tq,F,C] (numpy or DeviceArray)."""

    def __init__(self, n_analogs, ctx=None):
        self.n_analogs = int(n_analogs)
        self.ctx = ctx or default_context()
        self.state = None

    def fit(self, X, y):
        T = X.shape[0]
        if T >= self.n_analogs:  # gard.py:75-79
            self.k_ = self.n_analogs
        else:
            warnings.warn("length of X is less than n_analogs, setting n_analogs = len(X)")
            self.k_ = T
        self.state = self.ctx.analog_fit(X, y)
        return self

    def predict_pure(self, Xq, kind, thresh=None, sample_inds=None, want_neighbors=False, out=None):
        if kind == "best_analog" or self.n_analogs == 1:  # gard.py:291-296
            k, kind = 1, "best_analog"
        else:
            k = self.k_
        if kind not in KIND_CODES:
            raise ValueError(f"got unexpected kind {kind}")  # gard.py:336
        Tq, C = Xq.shape[0], Xq.shape[-1]
        if kind == "sample_analogs" and sample_inds is None:
            # gard.py:315 draws from the global NumPy RNG; one draw per cell, in cell order
            sample_inds = np.stack([np.random.randint(low=0, high=k, size=Tq) for _ in range(C)], axis=1)
        return self.ctx.analog_predict(self.state, Xq, k, KIND_CODES[kind], thresh, sample_inds, want_neighbors, out=out)

    def predict_regression(self, Xq, thresh=None, out=None):
        return self.ctx.analogreg_predict(self.state, Xq, self.k_, thresh, out=out)


ONE_CLASS_MESSAGE = ("This solver needs samples of at least 2 classes in the data, but the data contains only one class: "
                     "np.int8(0)")  # sklearn's LogisticRegression.fit on the labels of gard.py:204-207


def check_tree_kwargs(model):
    """kdtree_kwargs / query_kwargs (gard.py:82, 194, 299) that would change the neighbours are refused: the engine searches
    exactly, Euclidean; performance-only options of the KD-tree are ignored."""
    harmless = {"leaf_size", "dualtree", "breadth_first", "sort_results"}
    for name in ("kdtree_kwargs", "query_kwargs"):
        kw = getattr(model, name, None) or {}
        for k, v in kw.items():
            if k == "metric" and v in ("euclidean", "minkowski", "l2"):
                continue
            if k == "p" and v == 2:
                continue
            if k == "sort_results" and not v:
                raise NotImplementedError(f"{name}={{'sort_results': False}}: the engine returns neighbours in distance order")
            if k not in harmless:
                raise NotImplementedError(f"{name}={{{k!r}: {v!r}}} is not supported on the HIP engine (exact Euclidean search)")


def _as_2d(X, name="X"):
    a = np.asarray(X.values if isinstance(X, (pd.DataFrame, pd.Series)) else X, dtype=np.float64)
    if a.ndim == 1 and name == "X":
        raise ValueError(
            f"Expected 2D array, got 1D array instead:\narray={a}.\nReshape your data either using array.reshape(-1, 1) "
            "if your data has a single feature or array.reshape(1, -1) if it contains a single sample.")
    if not np.isfinite(a).all():
        raise _finite_error(name, a)
    return a


class AnalogBase(RegressorMixin, BaseEstimator):
    _fit_attributes = ["kdtree_", "X_", "y_", "k_"]
    n_outputs = 3
    output_names = OUTPUT_NAMES

    def fit(self, X, y):
        """Fit the analog model (gard.py:58-87): stores the training set on the device."""
        X2 = _as_2d(X, "X")
        y1 = _as_2d(y, "y")
        if y1.ndim == 2:
            if y1.shape[1] != 1:
                raise ValueError(f"y should be a 1d array, got an array of shape {y1.shape} instead.")
            y1 = y1[:, 0]
        if len(X2) != len(y1):
            raise ValueError(f"Found input variables with inconsistent numbers of samples: [{len(X2)}, {len(y1)}]")
        self.n_features_in_ = X2.shape[1]
        check_tree_kwargs(self)
        grid = AnalogGridModel(self.n_analogs)
        grid.fit(X2[:, :, None], y1[:, None])
        self._grid = grid
        self.k_ = grid.k_
        self.X_ = pd.DataFrame(X2, columns=X.columns) if isinstance(X, pd.DataFrame) else X2  # gard.py:49-50,84
        self.y_ = y1
        self.kdtree_ = grid.state  # opaque device handle standing in for the KDTree
        return self

    def _query(self, X):
        if not hasattr(self, "k_"):
            raise NotFittedError(
                f"This {type(self).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before "
                "using this estimator.")
        X2 = _as_2d(X, "X")
        if X2.shape[1] != self.n_features_in_:
            raise ValueError(f"X has {X2.shape[1]} features, but {type(self).__name__} is expecting "
                             f"{self.n_features_in_} features as input.")
        if not hasattr(self, "_grid"):
            self._grid = AnalogGridModel(self.n_analogs).fit(np.asarray(self.X_, dtype=np.float64)[:, :, None], self.y_[:, None])
        return X2

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_grid", None)
        d.pop("kdtree_", None)
        return d

    def __sklearn_tags__(self):
        from dataclasses import replace

        tags = super().__sklearn_tags__()
        return replace(tags, _skip_test="GARD models output 3 columns pandas dataframe instead of one during predict")


class AnalogRegression(AnalogBase):
    """AnalogRegression (gard.py:101-224): per query, a linear regression on its ``n_analogs`` nearest training samples.
    With ``thresh`` the exceedance probability comes from a logistic regression on the same analogs (the exact minimiser of
    sklearn's default L2-penalised objective; sklearn's own L-BFGS stops within ~1e-3 of it) and the linear model uses the
    analogs above the threshold."""

    def __init__(self, n_analogs=200, thresh=None, kdtree_kwargs=None, query_kwargs=None, logistic_kwargs=None,
                 lr_kwargs=None):
        self.n_analogs = n_analogs
        self.thresh = thresh
        self.kdtree_kwargs = kdtree_kwargs
        self.query_kwargs = query_kwargs
        self.logistic_kwargs = logistic_kwargs
        self.lr_kwargs = lr_kwargs

    def _check(self):
        check_sklearn_kwargs(self.lr_kwargs, LINEAR_NEUTRAL, "lr_kwargs", "plain OLS with intercept")
        check_sklearn_kwargs(self.logistic_kwargs, LOGISTIC_NEUTRAL, "logistic_kwargs", "LogisticRegression defaults: L2, C=1")
        check_tree_kwargs(self)

    def predict(self, X):
        X2 = self._query(X)
        self._check()
        out, status = self._grid.predict_regression(X2[:, :, None], self.thresh)
        if status[0] == _lib.CELL_ONE_CLASS:
            raise ValueError(ONE_CLASS_MESSAGE)
        out = out[:, :, 0]
        return pd.DataFrame(out, columns=self.output_names) if isinstance(X, pd.DataFrame) else out


class PureAnalog(AnalogBase):
    """PureAnalog (gard.py:227-364)."""

    def __init__(self, n_analogs=200, kind="best_analog", thresh=None, kdtree_kwargs=None, query_kwargs=None):
        self.n_analogs = n_analogs
        self.kind = kind
        self.thresh = thresh
        self.kdtree_kwargs = kdtree_kwargs
        self.query_kwargs = query_kwargs

    def predict(self, X):
        X2 = self._query(X)
        out, _ = self._grid.predict_pure(X2[:, :, None], self.kind, self.thresh)
        out = out[:, :, 0]
        if isinstance(X, pd.DataFrame):
            return pd.DataFrame(out, columns=self.output_names)  # fresh RangeIndex (gard.py:350-358, note N6)
        return out

    def kneighbors(self, X):
        """(dist, inds) of the k_ nearest training rows -- what ``kdtree_.query`` returns (gard.py:299)."""
        X2 = self._query(X)
        k = 1 if (self.kind == "best_analog" or self.n_analogs == 1) else self.k_
        _, _, inds, dist = self._grid.ctx.analog_predict(self._grid.state, X2[:, :, None], k, _lib.ANALOG_MEAN, None, None, True)
        return dist[:, :, 0], inds[:, :, 0]


class RegressionGridModel:
    """Batched PureRegression over the cell axis: X [T,F,C], y [T,C], Xq [Tq,F,C] (numpy or DeviceArray)."""

    def __init__(self, ctx=None, thresh=None):
        self.ctx = ctx or default_context()
        self.thresh = thresh
        self.state = None

    def fit(self, X, y):
        self.state = self.ctx.linreg_fit(X, y, self.thresh)
        return self

    def predict(self, Xq, out=None):
        if self.state is None:
            raise NotFittedError("This regression grid model is not fitted yet.")
        return self.ctx.linreg_predict(self.state, Xq, out=out)

    def export(self):
        return self.state.export()


class _FittedLinearModel:
    """``linear_model_`` stand-in: the numbers of the fitted sklearn LinearRegression (gard.py:439)."""

    def __init__(self, coef, intercept):
        self.coef_ = coef
        self.intercept_ = intercept


class _FittedLogisticModel:
    """``logistic_model_`` stand-in: the numbers of the fitted sklearn LogisticRegression (gard.py:420)."""

    def __init__(self, coef, intercept):
        self.coef_ = np.asarray(coef, dtype=np.float64).reshape(1, -1)
        self.intercept_ = np.asarray([intercept], dtype=np.float64)
        self.classes_ = np.array([0, 1], dtype=np.int8)


NO_SAMPLES_MESSAGE = "Found array with 0 sample(s) (shape=(0, {F})) while a minimum of 1 is required by LinearRegression."


class PureRegression(RegressorMixin, BaseEstimator):
    """PureRegression (gard.py:367-504): ordinary least squares of y on the features, the RMSE of the fit as prediction
    error.  With ``thresh``: the exceedance probability from a logistic regression of ``y > thresh`` on the features (the
    exact minimiser of sklearn's default L2-penalised objective; sklearn's own L-BFGS stops within ~1e-3 of it), the
    linear model on the exceeding samples; one class only: the threshold is dropped with the reference's warning.
    ``logistic_kwargs`` / ``linear_kwargs`` other than the defaults raise NotImplementedError."""

    _fit_attributes = ["logistic_model_", "linear_model_", "fit_error_"]
    n_outputs = 3
    output_names = OUTPUT_NAMES

    def __init__(self, thresh=None, logistic_kwargs=None, linear_kwargs=None):
        self.thresh = thresh
        self.logistic_kwargs = logistic_kwargs
        self.linear_kwargs = linear_kwargs

    def _check(self):
        check_sklearn_kwargs(self.linear_kwargs, LINEAR_NEUTRAL, "linear_kwargs", "plain OLS with intercept")
        check_sklearn_kwargs(self.logistic_kwargs, LOGISTIC_NEUTRAL, "logistic_kwargs", "LogisticRegression defaults: L2, C=1")

    def _adopt(self, e, c):
        """fitted attributes of cell ``c`` of an exported state (gard.py:420-443)"""
        self.linear_model_ = _FittedLinearModel(e["coef"][:, c].copy(), float(e["intercept"][c]))
        self.fit_error_ = float(e["fit_error"][c])
        if "logistic_coef" in e and not e["thresh_dropped"][c]:
            self.logistic_model_ = _FittedLogisticModel(e["logistic_coef"][:, c], float(e["logistic_intercept"][c]))

    def fit(self, X, y):
        self._check()
        X2 = _as_2d(X, "X")
        y1 = _as_2d(y, "y")
        if y1.ndim == 2:
            if y1.shape[1] != 1:
                raise ValueError(f"y should be a 1d array, got an array of shape {y1.shape} instead.")
            y1 = y1[:, 0]
        if len(X2) != len(y1):
            raise ValueError(f"Found input variables with inconsistent numbers of samples: [{len(X2)}, {len(y1)}]")
        self.n_features_in_ = X2.shape[1]
        self._grid = RegressionGridModel(thresh=self.thresh).fit(X2[:, :, None], y1[:, None])
        e = self._grid.export()
        self._n_fit = len(X2)
        if self.thresh is not None:
            if e["status"][0] == _lib.CELL_ONE_CLASS:  # no sample above the threshold: the linear model gets an empty set
                warnings.warn("Found only one class while attempting logistic regression. Mutating attribute thresh")
                self.thresh = None
                raise ValueError(NO_SAMPLES_MESSAGE.format(F=X2.shape[1]))
            if e["thresh_dropped"][0]:  # every sample above it (gard.py:426-437)
                warnings.warn("Found only one class while attempting logistic regression. Mutating attribute thresh")
                self.thresh = None
        self._adopt(e, 0)
        return self

    def predict(self, X):
        if not hasattr(self, "linear_model_"):
            raise NotFittedError(
                f"This {type(self).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before "
                "using this estimator.")
        X2 = _as_2d(X, "X")
        if X2.shape[1] != self.n_features_in_:
            raise ValueError(f"X has {X2.shape[1]} features, but {type(self).__name__} is expecting "
                             f"{self.n_features_in_} features as input.")
        if getattr(self, "_grid", None) is None:  # unpickled: the device state is rebuilt from the fitted numbers
            e = dict(coef=self.linear_model_.coef_.reshape(-1, 1), intercept=np.array([self.linear_model_.intercept_]),
                     fit_error=np.array([self.fit_error_]), status=np.zeros(1, np.int32), T=getattr(self, "_n_fit", 1))
            if self.thresh is not None:
                e.update(logistic_coef=self.logistic_model_.coef_.reshape(-1, 1), logistic_intercept=self.logistic_model_.intercept_,
                         thresh_dropped=np.zeros(1, np.int32))
            self._grid = RegressionGridModel(thresh=self.thresh)
            self._grid.state = self._grid.ctx.linreg_import(e)
        out, _ = self._grid.predict(X2[:, :, None])
        out = out[:, :, 0]
        return pd.DataFrame(out, columns=self.output_names) if isinstance(X, pd.DataFrame) else out  # gard.py:467-489

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_grid", None)
        return d

    def __sklearn_tags__(self):
        from dataclasses import replace

        tags = super().__sklearn_tags__()
        return replace(tags, _skip_test="GARD models output 3 columns pandas dataframe instead of one during predict")
