"""GPU parity: PureRegression(thresh=None) (csrc/sd_linreg.hip through the C ABI) vs goldens from the reference and the oracle."""
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
        PureRegression(thresh=0.0).fit(X, y)
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
