"""Quantile-mapping regressors with the reference's surface, computed by the HIP engine.

Mirrors ``skdownscale/pointwise_models/quantile.py``: ``QuantileMappingReressor`` (160-395) and
``EquidistantCdfMatcher`` (556-636), every ``extrapolate`` mode.  The arithmetic runs in ``csrc/sd_qm.hip`` through the
C ABI.  With ``'min'`` / ``'max'`` / ``'both'`` the reference interpolates samples *beyond the fitted range* across
synthetic end points at +-1e20 (quantile.py:17-18, 338-346): a cancellation that leaves ~1e5 of absolute rounding noise
in its own outputs; the engine evaluates the same expression (same noise level, not the same noise).  Samples inside the
fitted range are unaffected by the mode.
"""
from __future__ import annotations

import collections

import numpy as np
from sklearn.base import BaseEstimator, RegressorMixin, TransformerMixin
from sklearn.exceptions import NotFittedError
from sklearn.utils import check_array

from . import _lib
from .base import LINEAR_NEUTRAL, check_sklearn_kwargs
from .engine import default_context
from .trend import FittedLine, FittedTrend

Cdf = collections.namedtuple("Cdf", ["pp", "vals"])  # quantile.py:20


def plotting_positions(n, alpha=0.4, beta=0.4):
    """quantile.py:23-43."""
    return (np.arange(1, n + 1) - alpha) / (n + 1.0 - alpha - beta)


def check_max_features(array, n=1):
    """utils.py:10-25."""
    if array.ndim == 2 and array.shape[1] > n:
        raise ValueError(f"Found array with {array.shape[1]} features (shape={array.shape}) while a maximum of {n} is required")
    if array.ndim > 2:
        raise ValueError(f"Found array with {array.ndim} dimensions")
    return array


def check_extrapolate(extrapolate):
    if extrapolate not in (None, "1to1", "min", "max", "both"):
        raise ValueError(f"unknown value for extrapolate: {extrapolate}")  # quantile.py:348-349


FittedCunnane = collections.namedtuple("FittedCunnane", ["cdf_"])


class QuantileMapper(TransformerMixin, BaseEstimator):
    """Transform features using quantile mapping (quantile.py:46-157), default ``qt_kwargs`` / ``lt_kwargs``:
    ``transform(X)`` ranks X within itself (Cunnane plotting positions of its own sorted values, np.interp exact-hit
    rule) and maps the positions through the CDF of the data seen in ``fit`` (10-point OLS tails when X is longer).
    ``detrend=True`` (quantile.py:95-98, 128-145): both series lose their least-squares line over the sample index first;
    the line of the transformed series comes back afterwards, re-based on the fitted intercept (``x_trend_fit_``).
    This is the mapping BCSD applies per month (bcsd.py:59-79); here the whole series is one group."""

    _fit_attributes = ["x_cdf_fit_"]

    def __init__(self, detrend=False, lt_kwargs=None, qt_kwargs=None):
        self.detrend = detrend
        self.lt_kwargs = lt_kwargs
        self.qt_kwargs = qt_kwargs

    def _check(self):
        if self.detrend and self.lt_kwargs:
            extra = set(self.lt_kwargs) - {"lr_kwargs"}
            if extra:  # (LinearTrendTransformer takes lr_kwargs only: trend.py:40)
                raise TypeError(f"LinearTrendTransformer.__init__() got an unexpected keyword argument {sorted(extra)[0]!r}")
            check_sklearn_kwargs(self.lt_kwargs.get("lr_kwargs"), LINEAR_NEUTRAL, "QuantileMapper(lt_kwargs={'lr_kwargs': ...})",
                                 "only the LinearRegression defaults run")
        self._tails()

    def _tails(self):
        """(extrapolate, n_endpoints) of ``qt_kwargs`` (quantile.py:92, 136: the CunnaneTransformer of both directions gets them):
        which tails of the fitted inverse CDF continue along the least-squares line through their ``n_endpoints`` outermost
        points.  ``alpha`` / ``beta`` are accepted and, as in the reference, without effect (quantile.py:462)."""
        qt = self.qt_kwargs or {}
        for k in qt:
            if k not in ("alpha", "beta", "extrapolate", "n_endpoints"):
                raise TypeError(f"CunnaneTransformer.__init__() got an unexpected keyword argument {k!r}")
        extrapolate = qt.get("extrapolate", "both")
        if extrapolate not in ("min", "max", "both", "1to1", None):
            extrapolate = None  # (the reference only tests membership, quantile.py:527-528: anything else is np.interp's end values)
        n_endpoints = qt.get("n_endpoints", 10)
        if not isinstance(n_endpoints, (int, np.integer)) or n_endpoints < 1:
            raise NotImplementedError(f"CunnaneTransformer(n_endpoints={n_endpoints!r}): a positive integer is needed on the HIP engine")
        return extrapolate, int(n_endpoints)

    def fit(self, X, y=None):
        self._check()
        X = check_array(X, dtype="numeric", ensure_2d=True)
        X = check_max_features(X, n=1)
        Xv = np.asarray(X, dtype=np.float64).reshape(-1, 1)
        ctx = default_context()
        self._state = ctx.bcsd_fit(_lib.BCSD_PR, None, Xv, np.zeros(len(Xv), dtype=np.int32), 1, False, detrend=bool(self.detrend))
        self._state.set_tails(*self._tails())
        e = self._state.export()
        vals = e["y_sorted"][0]
        self.x_cdf_fit_ = FittedCunnane(Cdf(plotting_positions(len(vals)), vals))
        if self.detrend:
            slope, icpt = e["y_trend"][0, 0]
            self.x_trend_fit_ = FittedTrend(FittedLine(np.array([[slope]]), np.array([icpt])))
        self.n_features_in_ = 1
        return self

    def _rebuild_state(self, ctx):
        """device state from the fitted attributes (after unpickling)"""
        vals = np.asarray(self.x_cdf_fit_.cdf_.vals, dtype=np.float64)
        n = len(vals)
        exported = dict(info=dict(kind=_lib.BCSD_PR, G=1, T=n, C=1, return_anoms=False, detrend=bool(self.detrend)),
                        y_sorted=vals.reshape(1, n), x_climo=np.zeros((1, 1)), y_climo=np.array([[vals.mean()]]),
                        status=np.zeros(1, np.int32), group_offsets=np.array([0, n], dtype=np.int64))
        if self.detrend:
            line = self.x_trend_fit_.lr_model_
            exported["y_trend"] = np.array([float(np.ravel(line.coef_)[0]), float(np.ravel(line.intercept_)[0])]).reshape(1, 1, 2)
        state = ctx.bcsd_import(exported)
        state.set_tails(*self._tails())
        return state

    def transform(self, X):
        if not hasattr(self, "x_cdf_fit_"):
            raise NotFittedError(
                f"This {type(self).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before using this estimator.")
        X = check_array(X, dtype="numeric", ensure_2d=True)
        Xv = np.asarray(X, dtype=np.float64)[:, :1]
        ctx = default_context()
        if getattr(self, "_state", None) is None:  # unpickled: rebuild the device state from the fitted CDF
            self._state = self._rebuild_state(ctx)
        out, _ = ctx.bcsd_predict(self._state, np.ascontiguousarray(Xv), np.zeros(len(Xv), dtype=np.int32))
        return out

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_state", None)
        return d

    def __sklearn_tags__(self):
        from dataclasses import replace

        tags = super().__sklearn_tags__()
        return replace(tags, _skip_test="QuantileMapper only supports 1 feature and has temporal dependencies")


class CunnaneTransformer(TransformerMixin, BaseEstimator):
    """Quantile transform using Cunnane plotting positions with optional extrapolation (quantile.py:398-553).

    ``fit`` keeps the sorted sample and its plotting positions (``cdf_``); ``transform`` maps values to positions with
    ``np.interp`` semantics, ``inverse_transform`` maps positions to values, with the tails selected by ``extrapolate``
    extended by a least-squares line through the ``n_endpoints`` outermost points.  All three run on the HIP engine
    (``sd_qm_fit`` / ``sd_qm_cunnane``)."""

    _fit_attributes = ["cdf_"]

    def __init__(self, *, alpha=0.4, beta=0.4, extrapolate="both", n_endpoints=10):
        self.alpha = alpha
        self.beta = beta
        self.extrapolate = extrapolate
        self.n_endpoints = n_endpoints

    def _check(self):
        # (alpha / beta are accepted and, as in the reference, unused: fit calls plotting_positions(len(X)) with its
        # defaults, quantile.py:462)
        if self.extrapolate not in _lib.EXTRAP_CODES:
            raise ValueError(f"unknown value for extrapolate: {self.extrapolate}")

    def fit(self, X, y=None):
        self._check()
        X = check_array(X, ensure_2d=True)
        if X.shape[1] > 1:
            raise ValueError("CunnaneTransformer.fit() only supports a single feature")
        Xv = np.ascontiguousarray(X[:, :1], dtype=np.float64)
        if len(Xv) < 2:
            raise ValueError("CunnaneTransformer.fit() needs at least 2 samples on the HIP engine")
        self._state = default_context().qm_fit(Xv)
        self.cdf_ = Cdf(plotting_positions(len(Xv)), self._state.export(with_y=False)["x_sorted"][0])
        self.n_features_in_ = 1
        return self

    def _apply(self, direction, X):
        if not hasattr(self, "cdf_"):
            raise NotFittedError(
                f"This {type(self).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before using this estimator.")
        self._check()
        Xv = np.ascontiguousarray(X[:, :1], dtype=np.float64)
        ctx = default_context()
        if getattr(self, "_state", None) is None:  # unpickled: rebuild the device state from the fitted CDF
            self._state = ctx.qm_fit(np.asarray(self.cdf_.vals, dtype=np.float64).reshape(-1, 1))
        out, _ = ctx.qm_cunnane(self._state, direction, Xv, self.extrapolate, self.n_endpoints)
        return out

    def transform(self, X):
        X = check_array(X, ensure_2d=True)
        if X.shape[1] > 1:
            raise ValueError("CunnaneTransformer.transform() only supports a single feature")
        pps = self._apply(_lib.CUNNANE_FORWARD, X)
        if np.isinf(pps).any():
            # values beyond an extended tail: the reference fails here too (quantile.py:497/501 call ``.values`` on the
            # ndarray that check_array returned)
            raise AttributeError("'numpy.ndarray' object has no attribute 'values' (CunnaneTransformer.transform of values outside "
                                 f"the fitted range with extrapolate={self.extrapolate!r}: quantile.py:497)")
        return pps

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)

    def inverse_transform(self, X):
        X = check_array(X, ensure_2d=True)
        return self._apply(_lib.CUNNANE_INVERSE, X)

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_state", None)
        return d

    def __sklearn_tags__(self):
        from dataclasses import replace

        tags = super().__sklearn_tags__()
        return replace(tags, _skip_test="CunnaneTransformer only supports 1 feature")


class CunnaneGridModel:
    """Batched CunnaneTransformer over the cell axis: X [T, C] (numpy or DeviceArray), one CDF per cell."""

    def __init__(self, extrapolate="both", n_endpoints=10, ctx=None):
        if extrapolate not in _lib.EXTRAP_CODES:
            raise ValueError(f"unknown value for extrapolate: {extrapolate}")
        self.extrapolate = extrapolate
        self.n_endpoints = int(n_endpoints)
        self.ctx = ctx or default_context()
        self.state = None

    def fit(self, X):
        self.state = self.ctx.qm_fit(X)
        return self

    def _apply(self, direction, X, out):
        if self.state is None:
            raise NotFittedError("This Cunnane grid model is not fitted yet.")
        return self.ctx.qm_cunnane(self.state, direction, X, self.extrapolate, self.n_endpoints, out=out)

    def transform(self, X, out=None):
        """positions [Tp, C]; values beyond an extended tail come back as -inf / +inf (see CunnaneTransformer.transform)"""
        return self._apply(_lib.CUNNANE_FORWARD, X, out)

    def inverse_transform(self, P, out=None):
        return self._apply(_lib.CUNNANE_INVERSE, P, out)


class QuantileMapperGridModel:
    """Batched QuantileMapper over the cell axis: the BCSD kernels with the whole series of a cell as one group
    (quantile.py:81-147 per cell).  X [T, C] numpy or DeviceArray."""

    def __init__(self, ctx=None, detrend=False):
        self.ctx = ctx or default_context()
        self.detrend = bool(detrend)
        self.state = None

    def fit(self, X):
        T = X.shape[0]
        self.state = self.ctx.bcsd_fit(_lib.BCSD_PR, None, X, np.zeros(T, dtype=np.int32), 1, False, detrend=self.detrend)
        self.status_ = self.state.status()
        return self

    def transform(self, X, out=None):
        if self.state is None:
            raise NotFittedError("This QuantileMapper grid model is not fitted yet.")
        return self.ctx.bcsd_predict(self.state, X, np.zeros(X.shape[0], dtype=np.int32), out=out)


class QmGridModel:
    """Batched quantile-mapping regressor over the cell axis: X, y [T, C], Xp [Tp, C] (numpy or DeviceArray)."""

    def __init__(self, model, extrapolate=None, ctx=None, n_endpoints=10):
        check_extrapolate(extrapolate)
        self.model = int(model)
        self.extrapolate = extrapolate
        self.n_endpoints = int(n_endpoints)
        self.ctx = ctx or default_context()
        self.state = None

    def fit(self, X, y):
        self.state = self.ctx.qm_fit(X, y)
        return self

    def predict(self, Xp, out=None):
        if self.state is None:
            raise NotFittedError("This quantile-mapping grid model is not fitted yet.")
        return self.ctx.qm_predict(self.state, self.model, Xp, self.extrapolate, self.n_endpoints, out=out)


class QuantileMappingReressor(RegressorMixin, BaseEstimator):
    """Transform features using quantile mapping (quantile.py:160-395; the class name is the reference's spelling).

    Parameters
    ----------
    extrapolate : {None, 'min', 'max', 'both', '1to1'} -- how the CDFs are extended at the tails
    n_endpoints : int, points of the least-squares lines behind the synthetic end points of 'min' / 'max' / 'both'
        (and the reference's minimum-sample rule: 2 * n_endpoints + 1 samples to fit)
    """

    _fit_attributes = ["_X_cdf", "_y_cdf"]
    _engine_model = _lib.QM_REGRESSOR

    def __init__(self, extrapolate=None, n_endpoints=10):
        self.extrapolate = extrapolate
        self.n_endpoints = n_endpoints
        if self.n_endpoints < 2:
            raise ValueError("Invalid number of n_endpoints, must be >= 2")  # quantile.py:189-190

    def _engine_code(self):
        return self._engine_model

    def fit(self, X, y, **kwargs):
        X = check_array(X, dtype="numeric", ensure_min_samples=2 * self.n_endpoints + 1, ensure_2d=True)
        y = check_array(y, dtype="numeric", ensure_min_samples=2 * self.n_endpoints + 1, ensure_2d=False)
        X = check_max_features(X, n=1)
        check_extrapolate(self.extrapolate)
        self._grid = QmGridModel(self._engine_code(), self.extrapolate, n_endpoints=self.n_endpoints)
        self._grid.fit(np.asarray(X, dtype=np.float64).reshape(-1, 1), np.asarray(y, dtype=np.float64).reshape(-1, 1))
        e = self._grid.state.export()
        self._X_cdf = self._extended(e["x_sorted"][0])
        self._y_cdf = self._extended(e["y_sorted"][0])
        return self

    def _extended(self, vals):
        """The fitted attribute of quantile.py:312-387: n + 2 (position, value) pairs.  None / '1to1' duplicate the end
        points; 'min' / 'max' / 'both' put synthetic ones at -+1e20 on the least-squares line through the n_endpoints
        outermost points (attribute only: predictions compute their own in the kernel)."""
        pp = plotting_positions(len(vals))
        pp = np.concatenate([pp[:1], pp, pp[-1:]])
        vals = np.concatenate([vals[:1], vals, vals[-1:]])

        def line_at(s, x0):
            xm, ym = pp[s].mean(), vals[s].mean()
            slope = np.sum((pp[s] - xm) * (vals[s] - ym)) / np.sum((pp[s] - xm) ** 2)
            return (ym - slope * xm) + slope * x0

        if self.extrapolate in ("min", "both"):
            pp[0] = -1e20
            vals[0] = line_at(slice(1, self.n_endpoints + 1), pp[0])
        if self.extrapolate in ("max", "both"):
            pp[-1] = 1e20
            vals[-1] = line_at(slice(-self.n_endpoints - 1, -1), pp[-1])
        return Cdf(pp, vals)

    def predict(self, X, **kwargs):
        if not hasattr(self, "_X_cdf"):
            raise NotFittedError(
                f"This {type(self).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before using this estimator.")
        X = check_array(X, ensure_2d=True)
        if not hasattr(self, "_grid"):
            self._grid = QmGridModel(self._engine_code(), self.extrapolate, n_endpoints=self.n_endpoints)
            n = len(self._X_cdf.vals) - 2
            self._grid.fit(self._X_cdf.vals[1:-1].reshape(n, 1), self._y_cdf.vals[1:-1].reshape(n, 1))
        out, _ = self._grid.predict(np.asarray(X[:, :1], dtype=np.float64))
        return out[:, 0]

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_grid", None)  # device handle: rebuilt on demand from the fitted CDFs
        return d

    def __sklearn_tags__(self):
        from dataclasses import replace

        tags = super().__sklearn_tags__()
        return replace(tags, _skip_test="QuantileMappingReressor only supports 1 feature")


class EquidistantCdfMatcher(QuantileMappingReressor):
    """Equidistant CDF matching (quantile.py:556-636): quantile mapping that preserves the difference or the ratio
    between the new and the training X at equal plotting positions."""

    def __init__(self, kind="difference", extrapolate=None, n_endpoints=10, max_ratio=None):
        if kind not in ["difference", "ratio"]:
            raise NotImplementedError("kind must be either difference or ratio")
        self.kind = kind
        self.extrapolate = extrapolate
        self.n_endpoints = n_endpoints
        self.max_ratio = max_ratio
        if self.n_endpoints < 2:
            raise ValueError("Invalid number of n_endpoints, must be >= 2")

    def _engine_code(self):
        if self.max_ratio is not None:  # the reference calls np.min(ratio, max_ratio) (quantile.py:621-622) and fails
            raise NotImplementedError("EquidistantCdfMatcher(max_ratio=...) is not supported")
        return _lib.QM_EDCDF_DIFFERENCE if self.kind == "difference" else _lib.QM_EDCDF_RATIO


class TrendAwareQuantileMappingRegressor(RegressorMixin, BaseEstimator):
    """Experimental meta estimator for trend-aware quantile mapping (quantile.py:639-716): X and y lose their least-squares
    lines (``LinearTrendTransformer``, one batched line fit on the engine each), ``qm_estimator`` maps the detrended series,
    and the prediction gets the centred trend line of the new X plus the change of its mean back.

    Parameters
    ----------
    qm_estimator : a quantile-mapping regressor of this package (``QuantileMappingReressor``, ``EquidistantCdfMatcher``)
    trend_transformer : ``LinearTrendTransformer`` or None (default: ``LinearTrendTransformer()``)

    The reference only sets ``trend_transformer`` when None is passed (quantile.py:655-656: any other argument leaves the
    attribute unset and ``fit`` raises AttributeError); here a given transformer is kept.
    """

    def __init__(self, qm_estimator=None, trend_transformer=None):
        from .trend import LinearTrendTransformer

        self.qm_estimator = qm_estimator
        self.trend_transformer = LinearTrendTransformer() if trend_transformer is None else trend_transformer

    @staticmethod
    def _column(a):
        a = np.asarray(getattr(a, "values", a), dtype=np.float64)
        return a.reshape(len(a), -1)

    def fit(self, X, y):
        import copy

        if self.qm_estimator is None:
            raise AttributeError("'NoneType' object has no attribute 'fit'")  # what the reference does with its default
        Xv, yv = self._column(X), self._column(y)
        self._X_mean_fit = Xv.mean(axis=0)  # quantile.py:673-674
        self._y_mean_fit = yv.mean(axis=0)
        y_detrend = copy.deepcopy(self.trend_transformer).fit_transform(yv)  # quantile.py:676-680
        x_detrend = copy.deepcopy(self.trend_transformer).fit_transform(Xv)
        self.qm_estimator.fit(x_detrend, y_detrend)  # quantile.py:682
        return self

    def predict(self, X):
        import copy

        Xv = self._column(X)
        X_trend = copy.deepcopy(self.trend_transformer)
        x_detrend = X_trend.fit_transform(Xv)  # quantile.py:700-701
        y_hat = np.asarray(self.qm_estimator.predict(x_detrend), dtype=np.float64).reshape(-1, 1)  # quantile.py:703
        delta = (Xv.mean(axis=0) - self._X_mean_fit) + self._y_mean_fit  # quantile.py:707: projected change + observed mean
        trendline = X_trend.trendline(Xv)  # quantile.py:711-712
        trendline = trendline - trendline.mean()
        return y_hat + trendline + delta  # quantile.py:715
