"""LinearTrendTransformer with the reference's surface (skdownscale/pointwise_models/trend.py:14-91), computed by the HIP
engine: the least-squares line of every column over the sample index 0 .. n-1 is one cell of a batched ordinary least
squares (``sd_linreg_fit`` with the index as the single feature); ``transform`` / ``inverse_transform`` subtract / add
the line evaluated by ``sd_linreg_predict``.  ``QuantileMapper(detrend=True)`` does not go through this class: the BCSD
kernels remove and restore the lines of their segments on chip (csrc/sd_bcsd_rs.hip).
"""
from __future__ import annotations

import collections

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.exceptions import NotFittedError
from sklearn.utils import check_array

from .base import LINEAR_NEUTRAL, check_sklearn_kwargs
from .engine import default_context

FittedLine = collections.namedtuple("FittedLine", ["coef_", "intercept_"])  # the lr_model_ attributes trend.py:50-51 leaves behind
FittedTrend = collections.namedtuple("FittedTrend", ["lr_model_"])


def index_feature(n, cells):
    """np.arange(n) as the single feature of every cell: [n, 1, cells] (trend.py:51,83)."""
    return np.ascontiguousarray(np.broadcast_to(np.arange(n, dtype=np.float64).reshape(n, 1, 1), (n, 1, cells)))


class LinearTrendTransformer(TransformerMixin, BaseEstimator):
    """Transform features by removing linear trends (trend.py:14-91).

    Parameters
    ----------
    lr_kwargs : dict -- the reference forwards these to sklearn's LinearRegression; only its defaults run on the engine

    Attributes
    ----------
    lr_model_ : ``coef_`` [n_columns, 1] and ``intercept_`` [n_columns] of the fitted lines
    """

    def __init__(self, lr_kwargs=None):
        self.lr_kwargs = lr_kwargs

    def fit(self, X, y=None):
        check_sklearn_kwargs(self.lr_kwargs, LINEAR_NEUTRAL, "LinearTrendTransformer(lr_kwargs)", "only the LinearRegression defaults run")
        X = check_array(X, dtype="numeric", ensure_2d=True)
        Xv = np.asarray(X, dtype=np.float64)
        n, cells = Xv.shape
        self._state = default_context().linreg_fit(index_feature(n, cells), Xv)
        e = self._state.export()
        bad = np.flatnonzero(np.asarray(e["status"]) != 0)
        if bad.size:  # (check_array has refused NaN / inf already: anything else the engine flags must not pass silently)
            raise ValueError(f"LinearTrendTransformer.fit: the engine reported status {int(e['status'][bad[0]])} for column {int(bad[0])}")
        self.lr_model_ = FittedLine(e["coef"].T.copy(), e["intercept"].copy())  # sklearn: coef_ [n_targets, 1], intercept_ [n_targets]
        self._export = e
        self.n_features_in_ = cells
        return self

    def _require_fitted(self):
        if not hasattr(self, "lr_model_"):
            raise NotFittedError(
                f"This {type(self).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before using this estimator.")

    def trendline(self, X):
        """The fitted lines over the sample index of ``X`` (trend.py:80-83)."""
        self._require_fitted()
        X = check_array(X, dtype="numeric", ensure_2d=True)
        n, cells = X.shape
        if cells != self.n_features_in_:
            raise ValueError(f"X has {cells} features, but LinearTrendTransformer is expecting {self.n_features_in_} features as input.")
        ctx = default_context()
        if getattr(self, "_state", None) is None:  # unpickled: rebuild the device state from the plain arrays
            self._state = ctx.linreg_import(self._export)
        out, _ = ctx.linreg_predict(self._state, index_feature(n, cells))
        return out[:, 0, :]

    def transform(self, X):
        """trend.py:54-65."""
        line = self.trendline(X)
        return np.asarray(X, dtype=np.float64).reshape(line.shape) - line

    def inverse_transform(self, X):
        """trend.py:67-78."""
        line = self.trendline(X)
        return np.asarray(X, dtype=np.float64).reshape(line.shape) + line

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_state", None)
        return d

    def __sklearn_tags__(self):
        from dataclasses import replace

        tags = super().__sklearn_tags__()
        return replace(tags, _skip_test="LinearTrendTransformer depends on the temporal order of the samples")
