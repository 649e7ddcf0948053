"""BCSD estimators with the reference's sklearn-style surface, computed by the HIP engine.

Mirrors ``skdownscale/pointwise_models/bcsd.py`` of the reference: ``BcsdBase`` (14-93),
``BcsdPrecipitation`` (96-193), ``BcsdTemperature`` (196-289).  Constructor parameters, fitted
attributes (``y_climo_``, ``_x_climo``, ``quantile_mappers_``, ``n_features_in_``) and error
strings follow the reference; the arithmetic runs in ``csrc/sd_bcsd.hip`` through the C ABI.

A single estimator instance handles one cell (like the reference).  ``PointWiseDownscaler``
recognises these classes and fits/predicts *all cells of a grid in one launch* through
``BcsdGridModel``.
"""
from __future__ import annotations

import collections

import numpy as np
import pandas as pd
from sklearn.exceptions import NotFittedError

from . import _lib
from .base import TimeSynchronousDownscaler
from .engine import DeviceArray, default_context
from .groupers import DAY_GROUPER, MONTH_GROUPER, group_keys

Cdf = collections.namedtuple("Cdf", ["pp", "vals"])  # quantile.py:20
FittedCunnane = collections.namedtuple("FittedCunnane", ["cdf_"])

_QT_DEFAULTS = dict(alpha=0.4, beta=0.4, extrapolate="both", n_endpoints=10)  # quantile.py:419-426


def plotting_positions(n, alpha=0.4, beta=0.4):
    """quantile.py:23-43."""
    return (np.arange(1, n + 1) - alpha) / (n + 1.0 - alpha - beta)


class _FittedQuantileMapper:
    """Read-only stand-in for the reference's per-group ``QuantileMapper`` (quantile.py:46-157):
    exposes ``x_cdf_fit_.cdf_`` = (plotting positions, sorted values)."""

    def __init__(self, vals):
        self.x_cdf_fit_ = FittedCunnane(Cdf(plotting_positions(len(vals)), vals))


def check_supported(model):
    """Raise NotImplementedError for configurations outside the engine's hot path."""
    if not callable(model.time_grouper) or isinstance(model.time_grouper, type):
        raise NotImplementedError(
            f"time_grouper={model.time_grouper!r}: only callable group-key functions (e.g. MONTH_GROUPER) run on the "
            "HIP engine; 'daily_nasa-nex' / pandas frequency strings are not supported yet")
    if model.climate_trend is not model.time_grouper:
        raise NotImplementedError("climate_trend must be the same grouper as time_grouper on the HIP engine")
    qm = model.qm_kwargs or {}
    extra = set(qm) - {"detrend", "lt_kwargs", "qt_kwargs"}
    if extra:
        raise TypeError(f"QuantileMapper.__init__() got an unexpected keyword argument {sorted(extra)[0]!r}")
    if qm.get("detrend", False):
        raise NotImplementedError("QuantileMapper(detrend=True) is not supported on the HIP engine")
    qt = qm.get("qt_kwargs") or {}
    for k, v in qt.items():
        if k not in _QT_DEFAULTS:
            raise TypeError(f"CunnaneTransformer.__init__() got an unexpected keyword argument {k!r}")
        if v != _QT_DEFAULTS[k]:
            raise NotImplementedError(f"CunnaneTransformer({k}={v!r}): only the default {_QT_DEFAULTS[k]!r} runs on the HIP engine")


class BcsdGridModel:
    """Batched BCSD over the cell axis: fields are [T, C] (cells fastest), numpy or DeviceArray."""

    def __init__(self, kind, return_anoms=True, grouper=MONTH_GROUPER, ctx=None):
        self.kind = kind
        self.return_anoms = bool(return_anoms)
        self.grouper = grouper
        self.ctx = ctx or default_context()
        self.state = None
        self.keys = None

    def group_ids_fit(self, index):
        keys = group_keys(index, self.grouper)
        self.keys, gid = np.unique(keys, return_inverse=True)
        return gid.astype(np.int32)

    def group_ids_predict(self, index):
        keys = group_keys(index, self.grouper)
        pos = np.searchsorted(self.keys, keys)
        pos = np.clip(pos, 0, len(self.keys) - 1)
        bad = self.keys[pos] != keys
        if bad.any():
            raise KeyError(keys[bad][0])  # the reference fails on quantile_mappers_[key] (bcsd.py:77)
        return pos.astype(np.int32)

    def fit(self, X, y, index):
        gid = self.group_ids_fit(index)
        self.state = self.ctx.bcsd_fit(self.kind, X, y, gid, len(self.keys), self.return_anoms)
        self.status_ = self.state.status()
        return self

    def predict(self, Xp, index_p, out=None):
        if self.state is None:
            raise NotFittedError("This BCSD grid model is not fitted yet.")
        return self.ctx.bcsd_predict(self.state, Xp, self.group_ids_predict(index_p), out=out)

    def export(self):
        e = self.state.export()
        e["keys"] = self.keys
        return e


class BcsdBase(TimeSynchronousDownscaler):
    """Base class for BCSD model (bcsd.py:14-93)."""

    _fit_attributes = ["y_climo_", "quantile_mappers_"]
    _timestep = "M"
    _kind = None

    def __init__(self, time_grouper=MONTH_GROUPER, climate_trend_grouper=DAY_GROUPER, climate_trend=MONTH_GROUPER,
                 return_anoms=True, qm_kwargs=None):
        self.time_grouper = time_grouper
        self.climate_trend_grouper = climate_trend_grouper
        self.climate_trend = climate_trend
        self.return_anoms = return_anoms
        self.qm_kwargs = qm_kwargs

    # ---- helpers -------------------------------------------------------------------------------
    def _fit_engine(self, X2, y2, index):
        check_supported(self)
        grid = BcsdGridModel(self._kind, self.return_anoms, self.time_grouper)
        grid.fit(X2, y2, index)
        self._grid = grid
        self._adopt(grid.export(), 0)
        return grid

    def _adopt(self, exported, c):
        """Populate the reference's fitted attributes for cell ``c`` of an exported state."""
        keys = exported["keys"]
        off = exported["group_offsets"]
        self.y_climo_ = pd.DataFrame(exported["y_climo"][c].reshape(-1, 1), index=keys)  # bcsd.py:223
        if self._kind == _lib.BCSD_TAS:
            self._x_climo = pd.DataFrame(exported["x_climo"][c].reshape(-1, 1), index=keys)  # bcsd.py:222
        ys = exported["y_sorted"][c]
        self.quantile_mappers_ = {k: _FittedQuantileMapper(ys[off[g]:off[g + 1]]) for g, k in enumerate(keys)}

    def _require_fitted(self):
        if not hasattr(self, "y_climo_"):
            raise NotFittedError(
                f"This {type(self).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before "
                "using this estimator.")

    def _predict_engine(self, X2, index, columns):
        if not hasattr(self, "_grid"):  # unpickled / adopted from a grid fit: rebuild the device state lazily
            self._grid = self._rebuild_grid()
        out, status = self._grid.predict(X2, index)
        if status[0] == _lib.CELL_NONFINITE:
            raise ValueError("Input X contains NaN.")
        return pd.DataFrame(out, index=index, columns=columns)  # bcsd.py:78-79 keeps the time index

    def _rebuild_grid(self):
        keys = np.asarray(self.y_climo_.index)
        vals = [self.quantile_mappers_[k].x_cdf_fit_.cdf_.vals for k in keys]
        off = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.int64)
        T = int(off[-1])
        info = dict(kind=self._kind, G=len(keys), T=T, C=1, return_anoms=bool(self.return_anoms))
        xc = self._x_climo.values.reshape(1, -1) if self._kind == _lib.BCSD_TAS else np.zeros((1, len(keys)))
        exported = dict(info=info, y_sorted=np.concatenate(vals).reshape(1, T), x_climo=xc,
                        y_climo=self.y_climo_.values.reshape(1, -1), status=np.zeros(1, np.int32), group_offsets=off)
        grid = BcsdGridModel(self._kind, self.return_anoms, self.time_grouper)
        grid.keys = keys
        grid.state = grid.ctx.bcsd_import(exported)
        return grid

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_grid", None)  # device handle: rebuilt on demand from the plain-array attributes
        return d

    def __sklearn_tags__(self):
        from dataclasses import replace

        tags = super().__sklearn_tags__()
        tags = replace(tags, target_tags=replace(tags.target_tags, required=False))
        return replace(tags, _skip_test="BCSD only supports 1 feature and temporal order matters")


class BcsdPrecipitation(BcsdBase):
    """Classic BCSD model for precipitation (bcsd.py:96-193)."""

    _kind = _lib.BCSD_PR

    def fit(self, X, y):
        X2, y2, index = self._check_X_y(X, y)
        if self.n_features_in_ != 1:
            raise ValueError(f"BCSD only supports 1 feature, found {self.n_features_in_}")  # bcsd.py:133-134
        grid = self._fit_engine(X2, y2[:, :1], index)
        if grid.status_[0] == _lib.CELL_BAD_CLIMO:
            for a in ("y_climo_", "quantile_mappers_", "_grid"):
                self.__dict__.pop(a, None)
            raise ValueError("Invalid value in target climatology")  # bcsd.py:140-141
        return self

    def predict(self, X):
        self._require_fitted()
        X2, index = self._check_array(X, reset_features=True)  # bcsd.py:163 resets n_features_in_ (note N5)
        cols = X.columns if isinstance(X, pd.DataFrame) else None
        return self._predict_engine(X2[:, :1], index, cols)


class BcsdTemperature(BcsdBase):
    """BCSD model for temperature (bcsd.py:196-289)."""

    _kind = _lib.BCSD_TAS

    def fit(self, X, y):
        X2, y2, index = self._check_X_y(X, y)
        if self.n_features_in_ != 1:
            raise ValueError(f"BCSD only supports up to 4 features, found {self.n_features_in_}")  # bcsd.py:215-216 (sic)
        self._fit_engine(X2, y2[:, :1], index)
        return self

    def predict(self, X):
        self._require_fitted()
        X2, index = self._check_array(X)
        cols = X.columns if isinstance(X, pd.DataFrame) else None
        return self._predict_engine(X2[:, :1], index, cols)


def is_device(a):
    return isinstance(a, DeviceArray)
