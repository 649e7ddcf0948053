"""GPU parity: PureRegression, with and without thresh (csrc/sd_linreg.hip through the C ABI) vs goldens from the reference and the oracle."""
import pickle

import numpy as np
import pandas as pd
import pytest

import analog_oracle as ao
from _cases import assert_close, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from skdownscale_amd.engine import default_context

    return default_context()


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_goldens_from_the_reference(ctx, case):
    """g11_pure_regression.npz: one / three features, two collinear features (minimum-norm coefficients like sklearn's
    lstsq), a noise-free target (fit error at rounding level)."""
    g = load("g11_pure_regression")
    X, y, Xq = g[f"X{case}"], g[f"y{case}"], g[f"Xq{case}"]
    st = ctx.linreg_fit(X[:, :, None], y[:, None])
    e = st.export()
    assert e["status"].tolist() == [0]
    assert_close(e["coef"][:, 0], g[f"coef{case}"], rtol=1e-9, what=f"coef {case}")
    assert abs(e["intercept"][0] - float(g[f"intercept{case}"])) <= 1e-9 * (1 + abs(e["intercept"][0]))
    scale = float(np.std(y))
    assert abs(e["fit_error"][0] - float(g[f"fit_error{case}"])) <= 1e-9 * scale
    out, status = ctx.linreg_predict(st, Xq[:, :, None])
    assert (status == 0).all()
    assert_close(out[:, 0, 0], g[f"out{case}"][:, 0], rtol=1e-9, what=f"pred {case}")
    assert np.array_equal(out[:, 1, 0], g[f"out{case}"][:, 1])
    assert np.abs(out[:, 2, 0] - g[f"out{case}"][:, 2]).max() <= 1e-9 * scale


@pytest.mark.parametrize("T,Tq,F,C", [(3, 5, 1, 1), (365, 200, 2, 67), (14600, 3000, 1, 130), (2000, 700, 8, 65), (50, 10, 4, 5)])
@pytest.mark.parametrize("resident", [False, True])
def test_grid_vs_oracle(ctx, T, Tq, F, C, resident):
    """Cell tiles of 64 with a ragged last tile, every feature count, resident fields, a masked and a NaN cell."""
    rng = np.random.default_rng(T + F + C)
    X = 280 + 10 * rng.standard_normal((T, F, C))
    w = rng.standard_normal((F, C))
    y = np.einsum("tfc,fc->tc", X, w) + 2.0 + rng.standard_normal((T, C))
    Xq = 280 + 12 * rng.standard_normal((Tq, F, C))
    if C > 4:
        X[0, 0, 2] = np.nan       # masked cell (core.py:35-37)
        y[T // 2, 4] = np.inf     # non-finite target
    if resident:
        st = ctx.linreg_fit(ctx.to_device(X), ctx.to_device(y))
        out, status = ctx.linreg_predict(st, ctx.to_device(Xq))
        out = out.to_host()
    else:
        st = ctx.linreg_fit(X, y)
        out, status = ctx.linreg_predict(st, Xq)
    ok = np.ones(C, bool)
    if C > 4:
        ok[[2, 4]] = False
        assert status[2] == 1 and status[4] == 2 and np.isnan(out[:, :, [2, 4]]).all()
    assert (status[ok] == 0).all()
    exp = ao.pointwise_pure_regression(X[:, :, ok], y[:, ok], Xq[:, :, ok])
    assert_close(out[:, 0, ok], exp[:, 0], rtol=1e-9, scale=float(np.std(y[:, ok])), what="pred")
    assert np.array_equal(out[:, 1, ok], exp[:, 1])
    assert_close(out[:, 2, ok], exp[:, 2], rtol=1e-9, scale=float(np.std(y[:, ok])), what="fit error")
    Xq2 = Xq.copy()
    Xq2[1, F - 1, 0] = np.nan
    out2, status2 = ctx.linreg_predict(st, Xq2)
    assert status2[0] == 2 and np.isnan(out2[1, :, 0]).all() and np.array_equal(out2[0, :, 0], out[0, :, 0])


def test_estimator_and_pointwise_surface():
    """test_pointwise_models.py:144-200 style: DataFrame in -> 3-column DataFrame out, ndarray in -> [n, 3] array; fitted
    attributes; pickling; the grid driver batches the cells; thresh is refused."""
    from sklearn.exceptions import NotFittedError

    from skdownscale_amd import GridArray, PointWiseDownscaler, PureRegression

    n = 365
    index = pd.date_range("2019-01-01", periods=n)
    rng = np.random.default_rng(0)
    X = pd.DataFrame({"foo": np.sin(np.linspace(-10 * np.pi, 10 * np.pi, n)) * 10, "bar": rng.random(n)}, index=index)
    y = X["foo"] + 2
    m = PureRegression()
    with pytest.raises(NotFittedError):
        m.predict(X)
    out = m.fit(X, y).predict(X)
    assert isinstance(out, pd.DataFrame) and list(out.columns) == ["pred", "exceedance_prob", "prediction_error"]
    exp, coef, icpt, err = ao.pure_regression(X.values, y.values, X.values)
    assert_close(out.values, exp, rtol=1e-9, what="estimator")
    assert_close(m.linear_model_.coef_, coef, rtol=1e-9, scale=1.0, what="coef_") and abs(m.fit_error_ - err) < 1e-9
    arr = m.predict(X.values)
    assert isinstance(arr, np.ndarray) and arr.shape == (n, 3) and np.array_equal(arr, out.values)
    assert_close(pickle.loads(pickle.dumps(m)).predict(X.values), arr, rtol=1e-12, what="unpickled")
    with pytest.raises(NotImplementedError):
        PureRegression(linear_kwargs={"fit_intercept": False}).fit(X, y)
    with pytest.raises(ValueError, match="NaN"):
        m.predict(X.values * np.nan)
    Xg = rng.standard_normal((120, 2, 3, 4))
    yg = Xg[:, 0] * 2.0 - Xg[:, 1] + 0.1 * rng.standard_normal((120, 3, 4))
    Xg[0, 0, 1, 2] = np.nan
    pw = PointWiseDownscaler(PureRegression())
    pw.fit(GridArray(Xg, ("time", "variable", "y", "x")), GridArray(yg, ("time", "y", "x")))
    assert pw._models.kind == "linreg"
    res = pw.predict(GridArray(Xg, ("time", "variable", "y", "x")))
    assert res.dims == ("time", "variable", "y", "x") and res.shape == (120, 3, 3, 4)
    assert list(res.coords["variable"]) == ["pred", "exceedance_prob", "prediction_error"]
    assert np.isnan(res.values[:, :, 1, 2]).all()
    exp = ao.pure_regression(Xg[:, :, 0, 0], yg[:, 0, 0], Xg[:, :, 0, 0])[0]
    assert_close(res.values[:, :, 0, 0], exp, rtol=1e-9, what="pointwise pure regression")


PROB_TIGHT = 1e-6    # exceedance probability vs the reference's objective solved tightly (logistic_kwargs tol=1e-12)
PROB_DEFAULT = 1e-3  # ... vs the reference's default LogisticRegression: its L-BFGS stops at tol=1e-4, within ~2e-4 of the optimum


@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("resident", [False, True])
def test_thresholded_goldens_from_the_reference(ctx, case, resident):
    """g14_thresholded_regressions.npz: PureRegression(thresh) (gard.py:416-470) -- logistic exceedance probability, linear
    model on the exceeding samples, the cell whose samples all exceed (threshold dropped: probability 1)."""
    g = load("g14_thresholded_regressions")
    X, y, Xq, thresh = g[f"pr_X{case}"], g[f"pr_y{case}"], g[f"pr_Xq{case}"], float(g[f"pr_thresh{case}"])
    st = ctx.linreg_fit(ctx.to_device(X), ctx.to_device(y), thresh) if resident else ctx.linreg_fit(X, y, thresh)
    e = st.export()
    assert (e["status"] == 0).all() and np.array_equal(e["thresh_dropped"], g[f"pr_dropped{case}"])
    live = ~e["thresh_dropped"]
    np.testing.assert_allclose(e["logistic_coef"][:, live], g[f"pr_lcoef{case}_tight"][:, live], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(e["logistic_intercept"][live], g[f"pr_licpt{case}_tight"][live], rtol=1e-5, atol=1e-7)
    out, status = ctx.linreg_predict(st, ctx.to_device(Xq) if resident else Xq)
    out = out.to_host() if resident else out
    assert (status == 0).all()
    for name, tol in (("tight", PROB_TIGHT), ("default", PROB_DEFAULT)):
        exp = g[f"pr_out{case}_{name}"]
        assert_close(out[:, 0], exp[:, 0], rtol=1e-9, what=f"pred case {case}")
        assert_close(out[:, 2], exp[:, 2], rtol=1e-9, what=f"fit error case {case}")
        assert np.abs(out[:, 1] - exp[:, 1]).max() <= tol, (case, name, np.abs(out[:, 1] - exp[:, 1]).max())
    st2 = ctx.linreg_import(e)  # pickling / checkpoint path
    out2, _ = ctx.linreg_predict(st2, Xq)
    assert np.array_equal(out2, out)


def test_mixed_scale_features(ctx):
    """Features in very different units (1e-5 next to 1e5, g14_thresholded_regressions.npz ms_*): the equilibrated normal
    equations resolve both coefficients like lstsq on the data matrix does."""
    g = load("g14_thresholded_regressions")
    st = ctx.linreg_fit(g["ms_X"][:, :, None], g["ms_y"][:, None])
    e = st.export()
    np.testing.assert_allclose(e["coef"][:, 0], g["ms_coef"], rtol=1e-7)
    out, _ = ctx.linreg_predict(st, g["ms_Xq"][:, :, None])
    assert_close(out[:, :, 0], g["ms_out"], what="mixed-scale features")


@pytest.mark.parametrize("T,Tq,F,C", [(40, 17, 1, 5), (3000, 500, 2, 70), (14600, 1000, 1, 130), (5000, 100, 5, 9)])
def test_thresholded_grid_vs_oracle(ctx, T, Tq, F, C):
    """thresholded PureRegression vs the oracle on grids: partial tiles, several features, masked cells, one-class cells."""
    rng = np.random.default_rng(T + F)
    X = rng.standard_normal((T, F, C))
    y = X.sum(axis=1) * 0.7 + rng.standard_normal((T, C))
    Xq = rng.standard_normal((Tq, F, C))
    thresh = 0.1
    if C > 3:
        X[0, 0, 1] = np.nan                       # masked cell (core.py:35-37)
        y[:, 2] = np.abs(y[:, 2]) + thresh + 0.5  # every sample exceeds: threshold dropped
        y[:, 3] = -np.abs(y[:, 3])                # none does: flagged (the reference's linear model gets 0 samples)
    st = ctx.linreg_fit(X, y, thresh)
    e = st.export()
    out, status = ctx.linreg_predict(st, Xq)
    for c in range(C):
        if C > 3 and c == 1:
            assert e["status"][c] == 1 and np.isnan(out[:, :, c]).all()
            continue
        if C > 3 and c == 3:
            assert e["status"][c] == 4
            continue
        exp = ao.pure_regression_thresh(X[:, :, c], y[:, c], Xq[:, :, c], thresh)[0]
        assert_close(out[:, [0, 2], c], exp[:, [0, 2]], rtol=1e-9, what=f"cell {c}")
        assert np.abs(out[:, 1, c] - exp[:, 1]).max() <= 1e-8, (c, np.abs(out[:, 1, c] - exp[:, 1]).max())
        assert bool(e["thresh_dropped"][c]) == (C > 3 and c == 2)


def test_thresholded_estimator_surface():
    """PureRegression(thresh) like the reference: fitted attributes, the warning and the mutated thresh for one class, the
    error for an empty linear sample, pickling, DataFrame output; the grid driver batches it."""
    import warnings

    from skdownscale_amd import PointWiseDownscaler, PureRegression
    from skdownscale_amd.core import GridArray

    g = load("g14_thresholded_regressions")
    X, y, Xq, thresh = g["pr_X0"], g["pr_y0"], g["pr_Xq0"], float(g["pr_thresh0"])
    m = PureRegression(thresh=thresh).fit(pd.DataFrame(X[:, :, 0]), pd.Series(y[:, 0]))
    assert m.thresh == thresh and m.logistic_model_.coef_.shape == (1, 1)
    np.testing.assert_allclose(m.logistic_model_.coef_[0], g["pr_lcoef0_tight"][:, 0], rtol=1e-5)
    out = m.predict(pd.DataFrame(Xq[:, :, 0]))
    assert list(out.columns) == ["pred", "exceedance_prob", "prediction_error"]
    assert np.abs(out.values[:, 1] - g["pr_out0_tight"][:, 1, 0]).max() <= PROB_TIGHT
    m2 = pickle.loads(pickle.dumps(m))
    assert np.array_equal(m2.predict(Xq[:, :, 0]), out.values)
    last = X.shape[2] - 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = PureRegression(thresh=thresh).fit(X[:, :, last], y[:, last])
    assert m.thresh is None and any("only one class" in str(x.message) for x in w)
    assert (m.predict(Xq[:, :, last])[:, 1] == 1.0).all()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(ValueError, match="0 sample"):
            PureRegression(thresh=1e9).fit(X[:, :, 0], y[:, 0])
        pw = PointWiseDownscaler(PureRegression(thresh=thresh))
        pw.fit(GridArray(X, ("time", "variable", "point")), GridArray(y, ("time", "point")))
        got = pw.predict(GridArray(Xq, ("time", "variable", "point")))
    assert got.values.shape == (Xq.shape[0], 3, X.shape[2])
    assert np.abs(got.values[:, 1] - g["pr_out0_tight"][:, 1]).max() <= PROB_TIGHT
