"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy) of the reference's GARD analog hot path.

Checker for the HIP engine; imported only by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg.  Pinned against the real reference through
``tests/golden/`` (see ``tests/golden/make_golden.py``) and live in
``tests/test_oracle_vs_reference.py`` when ``/root/reference`` is present.

Citations are ``gard.py:line`` of ``/root/reference/skdownscale/pointwise_models/gard.py``.

Third-party arithmetic restated here: ``sklearn.neighbors.KDTree.query`` (gard.py:82,194,299;
scikit-learn 1.7.2, uv.lock:3033) returns, per query row, the ``k`` training rows with the smallest
*reduced distance* ``rdist = sum_j (x_j - t_j)**2`` accumulated sequentially over features
j = 0..F-1, ascending, and ``dist = sqrt(rdist)``.  Its order among exactly tied distances is
heap-defined (not index order), so KDTree tie order is **parity unpinned**; the restatement uses
(rdist, training index) lexicographic order, which is identical on tie-free data.
"""
from __future__ import annotations

import numpy as np

KIND_BEST = 0
KIND_SAMPLE = 1
KIND_WEIGHT = 2
KIND_MEAN = 3
KIND_NAMES = {"best_analog": KIND_BEST, "sample_analogs": KIND_SAMPLE, "weight_analogs": KIND_WEIGHT,
              "mean_analogs": KIND_MEAN}


def knn(X, Xq, k, chunk=512):
    """k nearest training rows for every query row.

    X [T,F], Xq [Tq,F] -> (dist [Tq,k] float64, inds [Tq,k] int64), ascending (rdist, index).
    """
    X = np.asarray(X, dtype=np.float64)
    Xq = np.asarray(Xq, dtype=np.float64)
    T, F = X.shape
    Tq = Xq.shape[0]
    dist = np.empty((Tq, k))
    inds = np.empty((Tq, k), dtype=np.int64)
    for s in range(0, Tq, chunk):
        q = Xq[s:s + chunk]
        d = np.zeros((q.shape[0], T))
        for j in range(F):  # sequential accumulation over features (sklearn euclidean rdist)
            diff = q[:, j][:, None] - X[:, j][None, :]
            d = d + diff * diff
        order = np.argsort(d, axis=1, kind="stable")[:, :k]  # stable => ties in index order
        inds[s:s + chunk] = order
        dist[s:s + chunk] = np.sqrt(np.take_along_axis(d, order, axis=1))
    return dist, inds


def analog_k(n_analogs, n_train, kind=None):
    """gard.py:73-79 (k_ = min(n_analogs, len(X))) and gard.py:291-296 (best_analog -> k = 1)."""
    k_ = min(n_analogs, n_train)
    if kind is not None and (kind == KIND_BEST or n_analogs == 1):
        return 1, KIND_BEST
    return k_, kind


def pure_analog_predict(X, y, Xq, n_analogs, kind, thresh=None, sample_inds=None):
    """gard.py:273-364 for one cell.  Returns (out [Tq,3], dist, inds).

    ``sample_inds`` replaces ``np.random.randint(0, k, size=Tq)`` (gard.py:315) for reproducibility.
    Output column order ['pred','exceedance_prob','prediction_error'] (gard.py:254-255).
    """
    y = np.asarray(y, dtype=np.float64)
    k, kind = analog_k(n_analogs, len(X), kind)
    dist, inds = knn(X, Xq, k)
    analogs = y[inds]  # gard.py:301
    if thresh is not None:
        mask = analogs > thresh  # gard.py:307
        masked = np.where(mask, analogs, np.nan)  # gard.py:308
    if kind == KIND_BEST:
        pred = analogs[:, 0]  # gard.py:311
    elif kind == KIND_SAMPLE:
        pred = analogs[np.arange(len(Xq)), np.asarray(sample_inds)]  # gard.py:313-317
    elif kind == KIND_WEIGHT:
        w = 1.0 / np.where(dist == 0, 1e-20, dist)  # gard.py:322-323
        src = masked if thresh is not None else analogs
        pred = np.sum(src * w, axis=1) / np.sum(w, axis=1)  # np.average (gard.py:324-327)
    elif kind == KIND_MEAN:
        src = masked if thresh is not None else analogs
        pred = src.mean(axis=1)  # gard.py:329-333
    else:
        raise ValueError(f"got unexpected kind {kind}")
    if thresh is not None:
        pred = np.nan_to_num(pred, nan=0.0)  # gard.py:341
        err = masked.std(axis=1)  # gard.py:342 (plain std: NaN if any analog <= thresh)
        prob = np.where(mask, 1, 0).mean(axis=1)  # gard.py:343
    else:
        err = analogs.std(axis=1)  # gard.py:345
        prob = np.ones(len(Xq))  # gard.py:346
    return np.column_stack([pred, prob, err]), dist, inds


def analog_regression_predict(X, y, Xq, n_analogs):
    """gard.py:152-224 with thresh=None: per query OLS on the k_ analogs.

    sklearn LinearRegression == centred least squares; solved by lstsq on centred data (min-norm
    for rank-deficient analog sets, like scipy gelsd).  Returns (out [Tq,3], inds).
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    Xq = np.asarray(Xq, dtype=np.float64)
    k = min(n_analogs, len(X))
    _, inds = knn(X, Xq, k)
    out = np.empty((len(Xq), 3))
    for i in range(len(Xq)):
        xa = X[inds[i]]
        ya = y[inds[i]]
        xm = xa.mean(axis=0)
        ym = ya.mean()
        coef, *_ = np.linalg.lstsq(xa - xm, ya - ym, rcond=None)
        icpt = ym - xm @ coef
        yhat = xa @ coef + icpt
        out[i, 0] = Xq[i] @ coef + icpt  # gard.py:221
        out[i, 1] = 1.0  # gard.py:209
        out[i, 2] = np.sqrt(np.mean((ya - yhat) ** 2))  # gard.py:218-219
    return out, inds


def logistic_fit(x, t, C=1.0, tol=1e-12, max_iter=100):
    """sklearn ``LogisticRegression()`` (penalty='l2', C=1.0, fit_intercept=True; gard.py:177, 204-212, 416-420) restated as
    the exact minimiser of  sum_i [log(1 + exp(z_i)) - t_i z_i] + ||w||^2 / (2 C),  z = x w + b  (the intercept is not
    penalised).  sklearn runs L-BFGS to gtol 1e-4 on the mean loss, so its coefficients sit within ~1e-3 of this optimum;
    here: damped Newton to machine precision.  Returns (w [F], b)."""
    x = np.asarray(x, dtype=np.float64).reshape(len(t), -1)
    t = np.asarray(t, dtype=np.float64)
    n, F = x.shape
    xa = np.column_stack([x, np.ones(n)])
    th = np.zeros(F + 1)
    reg = np.r_[np.full(F, 1.0 / C), 0.0]

    def fval(th):
        z = xa @ th
        return np.sum(np.logaddexp(0.0, z) - t * z) + 0.5 * np.sum(reg * th * th)

    f = fval(th)
    for _ in range(max_iter):
        z = xa @ th
        sg = 0.5 * (1.0 + np.tanh(0.5 * z))
        g = xa.T @ (sg - t) + reg * th
        if np.max(np.abs(g)) <= tol * max(1.0, n):
            break
        H = (xa * (sg * (1.0 - sg))[:, None]).T @ xa + np.diag(reg)
        H[np.diag_indices(F + 1)] += 1e-12
        d = np.linalg.solve(H, -g)
        a = 1.0
        while True:
            fn = fval(th + a * d)
            if fn <= f or a < 1e-10:
                break
            a *= 0.5
        th = th + a * d
        f = fn
    return th[:F], th[F]


def analog_regression_thresh_predict(X, y, Xq, n_analogs, thresh):
    """gard.py:152-224 with a threshold: per query, logistic regression of (analog value > thresh) on the k_ analogs'
    features and ``exceedance_prob = predict_proba(x)[0, 0]`` -- the probability of the *first* class, i.e. of NOT
    exceeding (gard.py:210, as written there); 1.0 when every analog exceeds; the linear model and its RMSE use the
    exceeding analogs only (gard.py:215-219).  A query whose analogs all stay at or below the threshold makes the
    reference raise (LogisticRegression needs two classes): reported as ``ValueError`` here too."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    Xq = np.asarray(Xq, dtype=np.float64)
    k = min(n_analogs, len(X))
    _, inds = knn(X, Xq, k)
    out = np.empty((len(Xq), 3))
    for i in range(len(Xq)):
        xa, ya = X[inds[i]], y[inds[i]]
        exc = ya > thresh
        if exc.all():
            prob = 1.0
        elif not exc.any():
            raise ValueError("This solver needs samples of at least 2 classes in the data, but the data contains only one class: 0")
        else:
            w, b = logistic_fit(xa, exc)
            prob = 1.0 - 1.0 / (1.0 + np.exp(-(Xq[i] @ w + b)))  # P(class 0)
        xs, ys = xa[exc], ya[exc]
        xm, ym = xs.mean(axis=0), ys.mean()
        coef, *_ = np.linalg.lstsq(xs - xm, ys - ym, rcond=None)
        icpt = ym - xm @ coef
        out[i] = [Xq[i] @ coef + icpt, prob, np.sqrt(np.mean((ys - (xs @ coef + icpt)) ** 2))]
    return out, inds


def pure_regression_thresh(X, y, Xq, thresh):
    """PureRegression(thresh).fit(X, y).predict(Xq) (gard.py:410-470): logistic regression of (y > thresh) on X over the
    whole series, ``exceedance_prob = predict_proba(Xq)[:, 1]``; linear model and fit_error_ on the exceeding samples.  One
    class only: the reference drops the threshold (warns, gard.py:426-437) and fits the plain linear model on ``exceed_ind``
    -- all samples, or none (then LinearRegression raises)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    Xq = np.asarray(Xq, dtype=np.float64)
    exc = y > thresh
    if exc.all() or not exc.any():
        if not exc.any():
            raise ValueError("Found array with 0 sample(s) (shape=(0, %d)) while a minimum of 1 is required by LinearRegression." % X.shape[1])
        prob = np.ones(len(Xq))
    else:
        w, b = logistic_fit(X, exc)
        prob = 1.0 / (1.0 + np.exp(-(Xq @ w + b)))
    xs, ys = X[exc], y[exc]
    xm, ym = xs.mean(axis=0), ys.mean()
    coef = np.linalg.lstsq(xs - xm, ys - ym, rcond=None)[0]
    icpt = ym - xm @ coef
    err = np.sqrt(np.mean((ys - (xs @ coef + icpt)) ** 2))
    return np.column_stack([Xq @ coef + icpt, prob, np.full(len(Xq), err)]), coef, icpt, err


def pointwise_analog(X, y, Xq, n_analogs, kind, thresh=None, sample_inds=None, regression=False):
    """Grid driver (core.py:69-143): X [T,F,C], y [T,C], Xq [Tq,F,C] -> out [Tq,3,C]."""
    T, F, C = X.shape
    Tq = Xq.shape[0]
    out = np.full((Tq, 3, C), np.nan)
    for c in range(C):
        if np.isnan(X[0, 0, c]):  # core.py:35-37
            continue
        if regression and thresh is not None:
            o, _ = analog_regression_thresh_predict(X[:, :, c], y[:, c], Xq[:, :, c], n_analogs, thresh)
        elif regression:
            o, _ = analog_regression_predict(X[:, :, c], y[:, c], Xq[:, :, c], n_analogs)
        else:
            o, _, _ = pure_analog_predict(X[:, :, c], y[:, c], Xq[:, :, c], n_analogs, kind, thresh,
                                          None if sample_inds is None else sample_inds[:, c])
        out[:, :, c] = o
    return out


def pure_regression(X, y, Xq):
    """PureRegression(thresh=None).fit(X, y).predict(Xq) (gard.py:414-470): sklearn LinearRegression = lstsq on the
    centred data (minimum-norm for collinear features), fit_error_ = RMSE of the fit.  Returns ([Tq,3], coef, intercept,
    fit_error)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    Xq = np.asarray(Xq, dtype=np.float64)
    xm, ym = X.mean(axis=0), y.mean()
    coef = np.linalg.lstsq(X - xm, y - ym, rcond=None)[0]
    icpt = ym - xm @ coef
    err = np.sqrt(np.mean((y - (X @ coef + icpt)) ** 2))
    out = np.column_stack([Xq @ coef + icpt, np.ones(len(Xq)), np.full(len(Xq), err)])
    return out, coef, icpt, err


def pointwise_pure_regression(X, y, Xq):
    """Grid driver over the cell axis: X [T,F,C], y [T,C], Xq [Tq,F,C] -> [Tq,3,C]."""
    C = X.shape[2]
    out = np.empty((Xq.shape[0], 3, C))
    for c in range(C):
        out[:, :, c] = pure_regression(X[:, :, c], y[:, c], Xq[:, :, c])[0]
    return out
