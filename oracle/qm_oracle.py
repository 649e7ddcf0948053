"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's quantile-mapping regressors.

Follows ``/root/reference/skdownscale/pointwise_models/quantile.py``:
``QuantileMappingReressor`` (160-395), ``EquidistantCdfMatcher`` (556-636) and ``CunnaneTransformer``
(398-553), one cell at a time, plus a grid driver over the cell axis like ``core.py:86-96,137-141``.
Pinned by ``tests/golden/g9_qm.npz`` and ``g10_cunnane.npz`` (generated from the real reference by
``tests/golden/make_golden.py``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU baseline may import this module.
"""
from __future__ import annotations

import numpy as np

SYNTHETIC_MIN = -1e20  # quantile.py:17
SYNTHETIC_MAX = 1e20   # quantile.py:18
EXTRAPOLATE = (None, "min", "max", "both", "1to1")


def plotting_positions(n, alpha=0.4, beta=0.4):
    """quantile.py:23-43."""
    return (np.arange(1, n + 1) - alpha) / (n + 1.0 - alpha - beta)


def ols_predict(x, y, x0):
    """sklearn LinearRegression().fit(x, y).predict(x0): centred least squares (quantile.py:366-385, 259-265)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xm, ym = x.mean(), y.mean()
    sxx = np.sum((x - xm) * (x - xm))
    slope = np.sum((x - xm) * (y - ym)) / sxx if sxx > 0 else 0.0
    return (ym - slope * xm) + slope * np.asarray(x0, dtype=np.float64)


def extended_cdf(data_sorted, extrapolate, n_endpoints):
    """quantile.py:312-387 (``_calc_extrapolated_cdf`` on already sorted data): (pp[n+2], vals[n+2])."""
    data = np.asarray(data_sorted, dtype=np.float64)
    n = len(data)
    pp = np.empty(n + 2)
    pp[1:-1] = plotting_positions(n)
    vals = np.empty(n + 2)
    vals[1:-1] = data
    vals[0], vals[-1] = data[0], data[-1]
    if extrapolate in (None, "1to1"):
        pp[0], pp[-1] = pp[1], pp[-2]
    elif extrapolate == "both":
        pp[0], pp[-1] = SYNTHETIC_MIN, SYNTHETIC_MAX
    elif extrapolate == "max":
        pp[0], pp[-1] = pp[1], SYNTHETIC_MAX
    elif extrapolate == "min":
        pp[0], pp[-1] = SYNTHETIC_MIN, pp[-2]
    else:
        raise ValueError(f"unknown value for extrapolate: {extrapolate}")
    if extrapolate in ("min", "both"):
        s = slice(1, n_endpoints + 1)
        vals[0] = ols_predict(pp[s], vals[s], pp[0])
    if extrapolate in ("max", "both"):
        s = slice(-n_endpoints - 1, -1)
        vals[-1] = ols_predict(pp[s], vals[s], pp[-1])
    return pp, vals


def qm_fit(X, y, extrapolate=None, n_endpoints=10):
    """quantile.py:197-223: the two extended CDFs."""
    return extended_cdf(np.sort(np.asarray(X, dtype=np.float64)), extrapolate, n_endpoints), \
        extended_cdf(np.sort(np.asarray(y, dtype=np.float64)), extrapolate, n_endpoints)


def _extrapolate_1to1(X, y_hat, x_cdf, y_cdf):
    """quantile.py:277-310; fit X and y have the same length here (one cell of a grid)."""
    x_min, x_max = x_cdf[1][0], x_cdf[1][-1]
    y_min, y_max = y_cdf[1][0], y_cdf[1][-1]
    hi = X > x_max
    y_hat[hi] = y_max + (X[hi] - x_max)
    lo = X < x_min
    y_hat[lo] = y_min + (X[lo] - x_min)
    return y_hat


def qmr_predict(state, Xp, extrapolate=None, n_endpoints=10):
    """QuantileMappingReressor.predict (quantile.py:225-275)."""
    x_cdf, y_cdf = state
    X = np.asarray(Xp, dtype=np.float64)
    order = np.argsort(X, kind="stable")
    pp, vals = extended_cdf(X[order], extrapolate, n_endpoints)
    left = -np.inf if extrapolate in ("min", "both") else None
    right = np.inf if extrapolate in ("max", "both") else None
    pp = np.interp(vals, x_cdf[1], x_cdf[0], left=left, right=right)  # quantile.py:247-249
    if np.isinf(pp).any():  # quantile.py:253-265 (regresses vals on pp, then "predicts" with vals -- as written there)
        lower = np.nonzero(pp == -np.inf)[0]
        upper = np.nonzero(pp == np.inf)[0]
        if len(lower):
            s = slice(lower[-1] + 1, lower[-1] + 1 + n_endpoints)
            pp[lower] = ols_predict(pp[s], vals[s], vals[lower])
        if len(upper):
            s = slice(upper[0] - n_endpoints, upper[0])
            pp[upper] = ols_predict(pp[s], vals[s], vals[upper])
    y_hat = np.full_like(X, np.nan)
    y_hat[order] = np.interp(pp, y_cdf[0], y_cdf[1])[1:-1]  # quantile.py:268-269
    if extrapolate == "1to1":
        y_hat = _extrapolate_1to1(X, y_hat, x_cdf, y_cdf)
    return y_hat


def ecm_predict(state, Xp, kind="difference", extrapolate=None, n_endpoints=10):
    """EquidistantCdfMatcher.predict (quantile.py:595-636), max_ratio=None."""
    x_cdf, y_cdf = state
    X = np.asarray(Xp, dtype=np.float64)
    order = np.argsort(X, kind="stable")  # the reference's argsort is unstable: results for tied X are unpinned
    pp, vals = extended_cdf(X[order], extrapolate, n_endpoints)
    x_train = np.interp(pp, x_cdf[0], x_cdf[1])  # quantile.py:613
    y_map = np.interp(pp, y_cdf[0], y_cdf[1])
    if kind == "difference":
        sorted_y = y_map + (vals - x_train)  # quantile.py:616-618
    elif kind == "ratio":
        sorted_y = y_map * (vals / x_train)  # quantile.py:619-623
    else:
        raise NotImplementedError("kind must be either difference or ratio")
    y_hat = np.full_like(X, np.nan)
    y_hat[order] = sorted_y[1:-1]
    if extrapolate == "1to1":
        y_hat = _extrapolate_1to1(X, y_hat, x_cdf, y_cdf)
    return y_hat


def pointwise_qm(model, X, y, Xp, extrapolate=None, n_endpoints=10, kind="difference"):
    """Grid driver: X, y [T, C], Xp [Tp, C] -> out [Tp, C]; masked cells (NaN at t=0, core.py:35-37) stay NaN."""
    T, C = y.shape
    out = np.full((Xp.shape[0], C), np.nan)
    for c in range(C):
        if np.isnan(X[0, c]):
            continue
        st = qm_fit(X[:, c], y[:, c], extrapolate, n_endpoints)
        if model == "qmr":
            out[:, c] = qmr_predict(st, Xp[:, c], extrapolate, n_endpoints)
        else:
            out[:, c] = ecm_predict(st, Xp[:, c], kind, extrapolate, n_endpoints)
    return out


# ---- CunnaneTransformer (quantile.py:398-553) -------------------------------------------------------

def _np_interp(x, xp, fp, left, right):
    """``np.interp`` spelled out (numpy ``arr_interp``): last node <= x, exact hit returns the node value."""
    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    n = len(xp)
    for i, v in enumerate(x):
        if v != v:
            out[i] = v
        elif v < xp[0]:
            out[i] = left
        elif v > xp[-1]:
            out[i] = right
        elif v == xp[-1]:
            out[i] = fp[-1]
        else:
            j = int(np.searchsorted(xp, v, side="right")) - 1
            if xp[j] == v or j == n - 1:
                out[i] = fp[j]
            else:
                slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j])
                out[i] = slope * (v - xp[j]) + fp[j]
    return out


def cunnane_fit(X):
    """quantile.py:438-463: (plotting positions, sorted sample)."""
    vals = np.sort(np.asarray(X, dtype=np.float64).ravel())
    return plotting_positions(len(vals)), vals


def cunnane_transform(cdf, X, extrapolate="both"):
    """quantile.py:465-503 for the samples it can serve: a value beyond an extended tail comes back as -inf / +inf
    (the reference's branch for those, 490-501, raises AttributeError on the ndarray ``check_array`` returned)."""
    pp, vals = cdf
    left = -np.inf if extrapolate in ("min", "both") else pp[0]
    right = np.inf if extrapolate in ("max", "both") else pp[-1]
    return _np_interp(X, vals, pp, left, right)


def cunnane_inverse(cdf, P, extrapolate="both", n_endpoints=10):
    """quantile.py:523-545: np.interp on the position grid, least-squares tails through the outermost points."""
    pp, vals = cdf
    P = np.asarray(P, dtype=np.float64)
    lo_ext, hi_ext = extrapolate in ("min", "both"), extrapolate in ("max", "both")
    out = _np_interp(P, pp, vals, -np.inf if lo_ext else vals[0], np.inf if hi_ext else vals[-1])
    lower, upper = out == -np.inf, out == np.inf
    if lower.any():
        s = slice(None, n_endpoints)
        out[lower] = ols_predict(pp[s], vals[s], P[lower])
    if upper.any():
        s = slice(-n_endpoints, None)
        out[upper] = ols_predict(pp[s], vals[s], P[upper])
    return out


# ---- TrendAwareQuantileMappingRegressor (quantile.py:639-716) ------------------------------------------

def _line(v):
    """least-squares line of a series over its sample index (trend.py:49-52): slope, intercept"""
    n = len(v)
    t = np.arange(n, dtype=np.float64)
    tb, vb = t.mean(), v.mean()
    slope = float(((t - tb) * (v - vb)).sum() / ((t - tb) ** 2).sum()) if n > 1 else 0.0
    return slope, float(vb - slope * tb)


def trend_aware_predict(model, X, y, Xp, extrapolate=None, n_endpoints=10, kind="difference"):
    """One cell: fit on (X, y), predict Xp.  X and y lose their lines (quantile.py:676-680), the regressor maps the detrended
    series (682, 700-703), the centred line of Xp and the change of the mean come back (707-715)."""
    X, y, Xp = (np.asarray(a, dtype=np.float64).reshape(-1) for a in (X, y, Xp))
    sx, ix = _line(X)
    sy, iy = _line(y)
    st = qm_fit(X - (np.arange(len(X)) * sx + ix), y - (np.arange(len(y)) * sy + iy), extrapolate, n_endpoints)
    sp, ip = _line(Xp)
    line = np.arange(len(Xp)) * sp + ip
    xd = Xp - line
    y_hat = qmr_predict(st, xd, extrapolate, n_endpoints) if model == "qmr" else ecm_predict(st, xd, kind, extrapolate, n_endpoints)
    delta = (Xp.mean() - X.mean()) + y.mean()
    return y_hat + (line - line.mean()) + delta
