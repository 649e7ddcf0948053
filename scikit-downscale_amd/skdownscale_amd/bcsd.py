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
from .base import LINEAR_NEUTRAL, TimeSynchronousDownscaler, check_sklearn_kwargs
from .engine import DeviceArray, default_context
from .groupers import DAY_GROUPER, MONTH_GROUPER, PaddedDOYGrouper, group_keys, padded_doy_table
from .trend import FittedLine, FittedTrend

Cdf = collections.namedtuple("Cdf", ["pp", "vals"])  # quantile.py:20
FittedCunnane = collections.namedtuple("FittedCunnane", ["cdf_"])

_QT_DEFAULTS = dict(alpha=0.4, beta=0.4, extrapolate="both", n_endpoints=10)  # quantile.py:419-426


def plotting_positions(n, alpha=0.4, beta=0.4):
    """quantile.py:23-43."""
    return (np.arange(1, n + 1) - alpha) / (n + 1.0 - alpha - beta)


class _FittedQuantileMapper:
    """Read-only stand-in for the reference's per-group ``QuantileMapper`` (quantile.py:46-157):
    exposes ``x_cdf_fit_.cdf_`` = (plotting positions, sorted values)."""

    def __init__(self, vals, trend=None):
        self.x_cdf_fit_ = FittedCunnane(Cdf(plotting_positions(len(vals)), vals))
        self.detrend = trend is not None
        if trend is not None:  # quantile.py:97: the fitted line of the (group's) series over its sample index
            self.x_trend_fit_ = FittedTrend(FittedLine(np.array([[trend[0]]]), np.array([trend[1]])))


def check_supported(model):
    """Raise NotImplementedError for configurations outside the engine's hot path."""
    tg = model.time_grouper
    if not (tg is PaddedDOYGrouper or (callable(tg) and not isinstance(tg, type))):
        raise NotImplementedError(
            f"time_grouper={tg!r}: callable group-key functions (e.g. MONTH_GROUPER) and 'daily_nasa-nex' run on the HIP engine")
    if not callable(model.climate_trend) or isinstance(model.climate_trend, type):
        raise NotImplementedError(f"climate_trend={model.climate_trend!r}: only callable group-key functions run on the HIP engine")
    qm = model.qm_kwargs or {}
    extra = set(qm) - {"detrend", "lt_kwargs", "qt_kwargs"}
    if extra:
        raise TypeError(f"QuantileMapper.__init__() got an unexpected keyword argument {sorted(extra)[0]!r}")
    if qm.get("detrend", False):
        check_sklearn_kwargs((qm.get("lt_kwargs") or {}).get("lr_kwargs"), LINEAR_NEUTRAL, "LinearTrendTransformer(lr_kwargs)",
                             "only the LinearRegression defaults run")
    qt_settings(model)


def qt_settings(model):
    """(extrapolate, n_endpoints) of ``qm_kwargs['qt_kwargs']`` (bcsd.py:59-67 -> quantile.py:92, 136: every group's
    CunnaneTransformer gets them).  ``alpha`` / ``beta`` are accepted and without effect, as in the reference: its fit calls
    ``plotting_positions(len(X))`` with the defaults (quantile.py:462)."""
    qt = (model.qm_kwargs or {}).get("qt_kwargs") or {}
    for k in qt:
        if k not in _QT_DEFAULTS:
            raise TypeError(f"CunnaneTransformer.__init__() got an unexpected keyword argument {k!r}")
    extrapolate = qt.get("extrapolate", _QT_DEFAULTS["extrapolate"])
    if extrapolate not in ("min", "max", "both", "1to1", None):
        # the reference takes any other value as "no extrapolation" (quantile.py:527-528 only tests membership): so does np.interp here
        extrapolate = None
    n_endpoints = qt.get("n_endpoints", _QT_DEFAULTS["n_endpoints"])
    if not isinstance(n_endpoints, (int, np.integer)) or n_endpoints < 1:
        raise NotImplementedError(f"CunnaneTransformer(n_endpoints={n_endpoints!r}): a positive integer is needed on the HIP engine")
    return extrapolate, int(n_endpoints)


class BcsdGridModel:
    """Batched BCSD over the cell axis: fields are [T, C] (cells fastest), numpy or DeviceArray.

    ``grouper``: the time grouper (group-key function -> one group per key, ``timestep='monthly'`` in the reference's
    terms) or ``PaddedDOYGrouper`` (``timestep='daily'``: 366 overlapping +-15-day day-of-year groups in fit,
    bcsd.py:50-55).  ``trend_grouper`` (``climate_trend``): groups of the 9-sample rolling mean in BcsdTemperature.predict
    (bcsd.py:247-250).  ``day_grouper`` (``climate_trend_grouper``): what the daily time step groups by in predict
    (bcsd.py:51-53) -- its keys select fitted day-of-year groups, which is the reference's behaviour."""

    def __init__(self, kind, return_anoms=True, grouper=MONTH_GROUPER, ctx=None, trend_grouper=None, day_grouper=DAY_GROUPER,
                 detrend=False, extrapolate="both", n_endpoints=10):
        self.kind = kind
        self.tails = (extrapolate, int(n_endpoints))  # qt_kwargs: tail handling of the fitted inverse CDFs
        self.return_anoms = bool(return_anoms)
        self.detrend = bool(detrend)  # qm_kwargs={'detrend': True}: quantile.py:95-98, 128-145 per group
        self.grouper = grouper
        self.trend_grouper = grouper if trend_grouper is None else trend_grouper
        self.day_grouper = day_grouper
        self.daily = grouper is PaddedDOYGrouper
        self.ctx = ctx or default_context()
        self.state = None
        self.keys = None

    def group_ids_fit(self, index):
        keys = group_keys(index, self.grouper)
        self.keys, gid = np.unique(keys, return_inverse=True)
        return gid.astype(np.int32)

    def group_ids_predict(self, index):
        keys = group_keys(index, self.day_grouper if self.daily else self.grouper)
        pos = np.searchsorted(self.keys, keys)
        pos = np.clip(pos, 0, len(self.keys) - 1)
        bad = self.keys[pos] != keys
        if bad.any():
            raise KeyError(keys[bad][0])  # the reference fails on quantile_mappers_[key] (bcsd.py:77)
        return pos.astype(np.int32)

    def fit(self, X, y, index):
        if self.daily:
            order, offsets = padded_doy_table(index)
            if (np.diff(offsets) == 0).any():  # QuantileMapper.fit on an empty group (bcsd.py:66-67 -> sklearn check_array)
                raise ValueError("Found array with 0 sample(s) (shape=(0, 1)) while a minimum of 1 is required by QuantileMapper.")
            self.keys = np.arange(1, 367)
            self.state = self.ctx.bcsd_fit_groups(self.kind, X, y, order, offsets, self.return_anoms, detrend=self.detrend)
        else:
            gid = self.group_ids_fit(index)
            self.state = self.ctx.bcsd_fit(self.kind, X, y, gid, len(self.keys), self.return_anoms, detrend=self.detrend)
        self._apply_tails()
        self.status_ = self.state.status()
        return self

    def _apply_tails(self):
        if self.tails != ("both", 10):
            self.state.set_tails(*self.tails)

    def predict(self, Xp, index_p, out=None, out_dtype=None):
        if self.state is None:
            raise NotFittedError("This BCSD grid model is not fitted yet.")
        if self.daily and self.return_anoms:
            # the reference divides / subtracts the climatology over the overlapping day-of-year groups of the result and
            # then finds 31x too many rows (bcsd.py:170-185, 266-267 with 271-281): mirrored, not repaired
            if self.kind == _lib.BCSD_TAS:
                raise ValueError("shape of climo is not equal to input array")
            n_rows = int(padded_doy_table(index_p)[1][-1])
            raise ValueError(f"Result shape ({n_rows}, 1) does not match input shape ({len(index_p)}, 1)")
        gid_p = self.group_ids_predict(index_p)
        if self.kind == _lib.BCSD_TAS and (self.daily or self.trend_grouper is not self.grouper):
            tkeys, gid_t = np.unique(group_keys(index_p, self.trend_grouper), return_inverse=True)
            return self.ctx.bcsd_predict_trend(self.state, Xp, gid_p, gid_t.astype(np.int32), len(tkeys), out=out)
        return self.ctx.bcsd_predict(self.state, Xp, gid_p, out=out, out_dtype=out_dtype)

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
    def _pre_fit(self):
        """bcsd.py:34-44: 'daily_nasa-nex' swaps the grouper class in (and stays swapped, as in the reference)."""
        if isinstance(self.time_grouper, str):
            if self.time_grouper == "daily_nasa-nex":
                self.time_grouper = PaddedDOYGrouper
                self.timestep = "daily"
            else:
                raise KeyError(self.time_grouper)  # the reference's df.groupby(<frequency string>) looks for a column of that name
        else:
            self.time_grouper_ = self.time_grouper
            self.timestep = "monthly"

    def _new_grid(self):
        extrapolate, n_endpoints = qt_settings(self)
        return BcsdGridModel(self._kind, self.return_anoms, self.time_grouper, trend_grouper=self.climate_trend,
                             day_grouper=self.climate_trend_grouper, detrend=bool((self.qm_kwargs or {}).get("detrend", False)),
                             extrapolate=extrapolate, n_endpoints=n_endpoints)

    def _fit_engine(self, X2, y2, index):
        self._pre_fit()
        check_supported(self)
        grid = self._new_grid()
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
        trend = exported.get("y_trend")
        self.quantile_mappers_ = {k: _FittedQuantileMapper(ys[off[g]:off[g + 1]], None if trend is None else trend[c, g])
                                  for g, k in enumerate(keys)}

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
        detrend = bool((self.qm_kwargs or {}).get("detrend", False))
        info = dict(kind=self._kind, G=len(keys), T=T, C=1, return_anoms=bool(self.return_anoms), detrend=detrend)
        xc = self._x_climo.values.reshape(1, -1) if self._kind == _lib.BCSD_TAS else np.zeros((1, len(keys)))
        exported = dict(info=info, y_sorted=np.concatenate(vals).reshape(1, T), x_climo=xc,
                        y_climo=self.y_climo_.values.reshape(1, -1), status=np.zeros(1, np.int32), group_offsets=off)
        if detrend:
            lines = [self.quantile_mappers_[k].x_trend_fit_.lr_model_ for k in keys]
            exported["y_trend"] = np.array([[float(np.ravel(l.coef_)[0]), float(np.ravel(l.intercept_)[0])] for l in lines]).reshape(1, -1, 2)
        grid = self._new_grid()
        grid.keys = keys
        grid.state = grid.ctx.bcsd_import(exported)
        grid._apply_tails()
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
