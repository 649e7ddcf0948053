"""GPU parity: quantile-mapping regressors (csrc/sd_qm.hip through the C ABI) vs goldens from the reference and the oracle."""
import pickle

import numpy as np
import pytest

import qm_oracle as qo
from _cases import assert_close, load

pytestmark = pytest.mark.gpu
MODELS = {"qmr": 0, "ecm_difference": 1, "ecm_ratio": 2}


@pytest.fixture(scope="module")
def ctx():
    from skdownscale_amd.engine import default_context

    return default_context()


def noise_floor(X, y, ex, ne):
    """Absolute rounding noise of the reference's (and the engine's) expression for samples *beyond the fitted range* under
    extrapolate 'min' / 'max' / 'both': positions and values are interpolated across synthetic nodes at +-1e20 whose values
    are ~1e20 * (slope of the tail line), so every such result carries a few ulps of the node value.  256 ulps of the
    largest synthetic y value of the cell; 0 for the other modes."""
    if ex not in ("min", "max", "both"):
        return np.zeros(X.shape[1])
    out = np.zeros(X.shape[1])
    for c in range(X.shape[1]):
        (_, vx), (_, vy) = qo.qm_fit(X[:, c], y[:, c], ex, ne)
        out[c] = 256 * np.finfo(float).eps * max(abs(vy[0]), abs(vy[-1]), abs(vx[0]), abs(vx[-1]))
    return out


def check_against(out, exp, X, Xp, floor, what):
    """samples inside the fitted X range: the north-star tolerance; beyond it: the noise floor of the expression"""
    inside = (Xp >= X.min(axis=0)) & (Xp <= X.max(axis=0))
    for c in range(X.shape[1]):
        assert_close(out[inside[:, c], c], exp[inside[:, c], c], scale=float(np.std(exp[inside[:, c], c])), what=f"{what} cell {c} (in range)")
        if floor[c] > 0:
            o, e = out[~inside[:, c], c], exp[~inside[:, c], c]
            # (the 'ratio' kind divides cancellation residues: 0/0 and x/0 turn up on either side)
            both = np.isfinite(o) & np.isfinite(e)
            err = np.abs(o[both] - e[both])
            assert (err <= floor[c]).all(), f"{what} cell {c}: beyond the fitted range, max err {err.max():.3e} > noise floor {floor[c]:.3e}"
        else:
            assert_close(out[~inside[:, c], c], exp[~inside[:, c], c], scale=float(np.std(exp[:, c])), what=f"{what} cell {c} (beyond the range)")


@pytest.mark.parametrize("case", [0, 1, 2])
@pytest.mark.parametrize("resident", [False, True])
def test_goldens_from_the_reference(ctx, case, resident):
    """g9_qm.npz: QuantileMappingReressor / EquidistantCdfMatcher outputs of the real reference for every extrapolate mode
    and n_endpoints 10 / 2; predict series equal / longer and shifted up / shorter and shifted down."""
    g = load("g9_qm")
    X, y, Xp = g[f"X{case}"], g[f"y{case}"], g[f"Xp{case}"]
    st = ctx.qm_fit(ctx.to_device(X), ctx.to_device(y)) if resident else ctx.qm_fit(X, y)
    for ex in (None, "1to1", "min", "max", "both"):
        for ne in (10, 2):
            floor = noise_floor(X, y, ex, ne)
            for name, code in MODELS.items():
                out, status = ctx.qm_predict(st, code, ctx.to_device(Xp) if resident else Xp, ex, ne)
                out = out.to_host() if resident else out
                assert (status == 0).all()
                check_against(out, g[f"out{case}_{name}_{ex}_{ne}"], X, Xp, floor, f"{name} case {case} extrapolate={ex} n_endpoints={ne}")


@pytest.mark.parametrize("T,Tp,C", [(21, 21, 1), (365, 400, 5), (3000, 2999, 7), (14600, 14600, 4), (14600, 9000, 3), (5000, 19000, 2)])
def test_vs_oracle_sizes_and_ties(ctx, T, Tp, C):
    """Every sort width (1 ... 19 samples per thread), ties in fit and predict series (quantized data: the stable
    (value, index) order defines the ranks of EquidistantCdfMatcher), values outside the fitted range, every
    extrapolate mode."""
    rng = np.random.default_rng(T + Tp)
    X = np.round(10 + 3 * rng.standard_normal((T, C)), 2)
    y = np.round(12 + 4 * rng.standard_normal((T, C)), 2) + 20.0
    Xp = np.round(11 + 4 * rng.standard_normal((Tp, C)), 2) + 20.0 * (np.arange(C) % 2)
    X = X + 20.0 * (np.arange(C) % 2)
    st = ctx.qm_fit(X, y)
    e = st.export()
    assert np.array_equal(e["x_sorted"], np.sort(X, axis=0).T) and np.array_equal(e["y_sorted"], np.sort(y, axis=0).T)
    ne = 10 if T >= 21 else 2
    for ex in (None, "1to1", "min", "max", "both"):
        floor = noise_floor(X, y, ex, ne)
        out, _ = ctx.qm_predict(st, 0, Xp, ex, ne)
        check_against(out, qo.pointwise_qm("qmr", X, y, Xp, ex, ne), X, Xp, floor, f"qmr {ex}")
        for kind, code in (("difference", 1), ("ratio", 2)):
            out, _ = ctx.qm_predict(st, code, Xp, ex, ne)
            check_against(out, qo.pointwise_qm("ecm", X, y, Xp, ex, ne, kind=kind), X, Xp, floor, f"ecm {kind} {ex}")


def test_linear_model_quantile_mapping_parametrisations():
    """The reference's smoke test (test_pointwise_models.py:111-141) for its ten quantile-mapping parametrisations, on the
    same 365-day sine; beyond the length check the outputs are compared with the oracle (every sample is inside the
    fitted range here, so all modes are well conditioned)."""
    import pandas as pd

    from skdownscale_amd import EquidistantCdfMatcher, QuantileMappingReressor

    n = 365
    index = pd.date_range("2019-01-01", periods=n)
    X = pd.DataFrame({"foo": np.sin(np.linspace(-10 * np.pi, 10 * np.pi, n)) * 10}, index=index)
    y = X + 2
    for ex in (None, "min", "max", "both", "1to1"):
        for model, name in ((QuantileMappingReressor(extrapolate=ex), "qmr"), (EquidistantCdfMatcher(extrapolate=ex), "ecm")):
            model.fit(X, y)
            y_hat = model.predict(X)
            assert len(y_hat) == len(X)
            exp = qo.pointwise_qm(name, X.values, y.values, X.values, ex)[:, 0]
            assert_close(y_hat, exp, what=f"{name} extrapolate={ex}")
            assert model._X_cdf.pp.shape == (n + 2,) and (model._X_cdf.pp[0] == -1e20) == (ex in ("min", "both"))
            np.testing.assert_allclose(model._y_cdf.vals, qo.qm_fit(X.values[:, 0], y.values[:, 0], ex, 10)[1][1], rtol=1e-12)


def test_masked_and_nonfinite_cells(ctx):
    rng = np.random.default_rng(2)
    X, y, Xp = (rng.standard_normal((300, 4)) for _ in range(3))
    X[0, 1] = np.nan          # masked cell (core.py:35-37)
    y[17, 2] = np.inf         # non-finite inside an active cell
    st = ctx.qm_fit(X, y)
    assert st.export()["status"].tolist() == [0, 1, 2, 0]
    Xp[5, 3] = np.nan
    out, status = ctx.qm_predict(st, 0, Xp)
    assert status.tolist() == [0, 1, 2, 2]
    assert np.isnan(out[:, 1:]).all() and np.isfinite(out[:, 0]).all()


def test_estimators_reference_surface():
    """test_pointwise_models.py:323-344 (exact), constructor / error behaviour, pickling, PointWiseDownscaler."""
    import pandas as pd

    from skdownscale_amd import EquidistantCdfMatcher, GridArray, PointWiseDownscaler, QuantileMappingReressor

    x = np.arange(1, 22)
    for kind, Xt, exp in (("difference", x + 2, x + 3 + 2), ("ratio", x * 2, (x + 3) * 2)):
        m = EquidistantCdfMatcher(kind=kind).fit(X=pd.DataFrame(x), y=pd.DataFrame(x + 3))
        assert (m.predict(pd.DataFrame(Xt)).reshape(-1, 1) == exp.reshape(-1, 1)).all()
        m2 = pickle.loads(pickle.dumps(m))
        assert np.array_equal(m2.predict(pd.DataFrame(Xt)), m.predict(pd.DataFrame(Xt)))
    with pytest.raises(ValueError, match="n_endpoints"):
        QuantileMappingReressor(n_endpoints=1)
    with pytest.raises(NotImplementedError):
        EquidistantCdfMatcher(kind="sum")
    with pytest.raises(ValueError, match="unknown value for extrapolate"):
        QuantileMappingReressor(extrapolate="sideways").fit(np.arange(30.0).reshape(-1, 1), np.arange(30.0))
    with pytest.raises(ValueError, match="minimum of 21"):
        QuantileMappingReressor().fit(np.arange(10.0).reshape(-1, 1), np.arange(10.0))
    rng = np.random.default_rng(4)
    X, y, Xp = (10 + rng.standard_normal((200, 1)) for _ in range(3))
    m = QuantileMappingReressor(extrapolate="1to1").fit(X, y[:, 0])
    assert_close(m.predict(Xp), qo.qmr_predict(qo.qm_fit(X[:, 0], y[:, 0], "1to1"), Xp[:, 0], "1to1"), what="estimator")
    assert len(m._X_cdf.pp) == 202 and m._X_cdf.vals[0] == X.min()
    # grid driver: (time, y, x) fields, one masked cell
    Xg, yg, Xpg = (10 + rng.standard_normal((120, 3, 4)) for _ in range(3))
    Xg[0, 1, 2] = np.nan
    pw = PointWiseDownscaler(EquidistantCdfMatcher(kind="difference"))
    dims = ("time", "y", "x")
    pw.fit(GridArray(Xg, dims), GridArray(yg, dims))
    out = pw.predict(GridArray(Xpg, dims))
    exp = qo.pointwise_qm("ecm", Xg.reshape(120, 12), yg.reshape(120, 12), Xpg.reshape(120, 12)).reshape(120, 3, 4)
    assert out.dims == dims and np.isnan(out.values[:, 1, 2]).all()
    assert_close(out.values, exp, what="PointWiseDownscaler ecm")


def test_quantile_mapper_qt_kwargs_golden():
    """Stand-alone QuantileMapper(qt_kwargs=...) against g20_qm_qt_kwargs.npz from the real reference (quantile.py:92, 136: the
    keywords reach the CunnaneTransformer of the fitted CDF): every `extrapolate`, several `n_endpoints`, `alpha` / `beta`
    without effect; a fit of 800 samples (fused kernels) and one of 3 000 (one group beyond 2 112 samples: the workgroup-sort
    kernels take the same tail settings); the fitted object survives pickling with them."""
    import ast
    import pickle

    from skdownscale_amd import QuantileMapper

    g = load("g20_qm_qt_kwargs")
    for tag in ("s", "l"):
        fit, new = g[f"{tag}_fit"], g[f"{tag}_new"]
        for i in range(int(g["n_variants"])):
            kw = dict(ast.literal_eval(str(g["variants"][i])))
            m = QuantileMapper(qt_kwargs=kw).fit(fit)
            assert_close(m.transform(new), g[f"{tag}{i}"], what=f"QuantileMapper(qt_kwargs={kw}) {tag}")
            if i in (2, 4):
                assert_close(pickle.loads(pickle.dumps(m)).transform(new), g[f"{tag}{i}"], what=f"unpickled {kw} {tag}")
    with pytest.raises(TypeError):
        QuantileMapper(qt_kwargs={"gamma": 1}).fit(g["s_fit"])


def test_quantile_mapper_transformer():
    """The reference's only numeric test of this path (test_pointwise_models.py:81-90) and its golden output; longer /
    shorter series than the fitted one (tail OLS and table paths) vs the BCSD oracle's mapping; series beyond the
    register-sort limit; pickling; use inside PointWiseDownscaler.transform."""
    import bcsd_oracle as bo
    from skdownscale_amd import GridArray, PointWiseDownscaler, QuantileMapper

    n = 100
    expected = (np.sin(np.linspace(-10 * np.pi, 10 * np.pi, n)) * 10).reshape(-1, 1)
    mapper = QuantileMapper().fit(expected)
    actual = mapper.transform(expected + 2)
    np.testing.assert_almost_equal(actual, expected)
    g = load("g8_reference_tests")
    assert_close(actual, g["qm_actual"], what="test_quantile_mapper golden")
    assert np.array_equal(mapper.x_cdf_fit_.cdf_.vals, np.sort(expected[:, 0]))
    rng = np.random.default_rng(8)
    for nfit, npred in ((500, 500), (500, 800), (800, 300), (3000, 3000), (14600, 14600), (14600, 9000)):
        a, b = rng.standard_normal((nfit, 1)), 0.5 + 1.5 * rng.standard_normal((npred, 1))
        m = QuantileMapper().fit(a)
        exp, _ = bo.pointwise_fit_predict(bo.PR, None, a, b, np.zeros(nfit, np.int32), np.zeros(npred, np.int32), G=1,
                                          return_anoms=False)
        assert_close(m.transform(b), exp, what=f"quantile mapper {nfit}->{npred}")
        assert np.array_equal(pickle.loads(pickle.dumps(m)).transform(b), m.transform(b))
    with pytest.raises(NotImplementedError):  # detrend=True itself runs on the engine (tests/test_gpu_detrend.py); non-default trend options do not
        QuantileMapper(detrend=True, lt_kwargs={"lr_kwargs": {"fit_intercept": False}}).fit(expected)
    X = rng.standard_normal((200, 2, 2))
    pw = PointWiseDownscaler(QuantileMapper())
    pw.fit(GridArray(X, ("time", "y", "x")))
    out = pw.transform(GridArray(X + 1.0, ("time", "y", "x")))
    assert_close(out.values[:, 0], X, what="pointwise quantile mapper")  # the bias is removed


# ---- CunnaneTransformer (quantile.py:398-553) through sd_qm_cunnane ----

@pytest.mark.parametrize("case", [0, 1, 2])
def test_cunnane_goldens_from_the_reference(ctx, case):
    """g10_cunnane.npz: transform (bit-exact positions: the same divisions as NumPy), fit_transform, inverse_transform
    with the least-squares tails, every extrapolate mode, n_endpoints 10 / 3, a fit shorter than n_endpoints."""
    g = load("g10_cunnane")
    x = g[f"x{case}"]
    st = ctx.qm_fit(x.reshape(-1, 1))
    assert np.array_equal(st.export(with_y=False)["x_sorted"][0], np.sort(x))
    for ex in qo.EXTRAPOLATE:
        for ne in (10, 3):
            fwd, status = ctx.qm_cunnane(st, 0, g[f"inside{case}"].reshape(-1, 1), ex, ne)
            assert (status == 0).all()
            assert_close(fwd[:, 0], g[f"fwd{case}_{ex}_{ne}"], rtol=1e-13, what=f"cunnane forward {case} {ex}")
            if ex in (None, "1to1"):
                out, _ = ctx.qm_cunnane(st, 0, g[f"outside{case}"].reshape(-1, 1), ex, ne)
                assert_close(out[:, 0], g[f"fwd_out{case}_{ex}_{ne}"], rtol=1e-13, what="cunnane forward clamped")
            inv, _ = ctx.qm_cunnane(st, 1, g[f"p{case}"].reshape(-1, 1), ex, ne)
            assert_close(inv[:, 0], g[f"inv{case}_{ex}_{ne}"], rtol=1e-9, what=f"cunnane inverse {case} {ex} {ne}")
    ft, _ = ctx.qm_cunnane(st, 0, x.reshape(-1, 1), "both", 10)
    assert_close(ft[:, 0], g[f"fit_transform{case}"], rtol=1e-13, what="fit_transform")


@pytest.mark.parametrize("T,Tp,C", [(365, 500, 5), (14600, 14600, 3), (19000, 700, 2)])
def test_cunnane_grid_vs_oracle(ctx, T, Tp, C):
    """Grid model over the cell axis, resident fields, quantized data (ties), values beyond the ends -> +-inf marks."""
    from skdownscale_amd import CunnaneGridModel

    rng = np.random.default_rng(T + C)
    X = np.round((10 + 3 * rng.standard_normal((T, C))) * 8) / 8
    Xn = 10 + 3.6 * rng.standard_normal((Tp, C))
    P = rng.uniform(-0.05, 1.05, (Tp, C))
    for ex, ne in (("both", 10), (None, 10), ("min", 4)):
        gm = CunnaneGridModel(ex, ne, ctx).fit(ctx.to_device(X))
        fwd, _ = gm.transform(ctx.to_device(Xn))
        inv, _ = gm.inverse_transform(ctx.to_device(P))
        fwd, inv = fwd.to_host(), inv.to_host()
        for c in range(C):
            cdf = qo.cunnane_fit(X[:, c])
            exp = qo.cunnane_transform(cdf, Xn[:, c], ex)
            fin = np.isfinite(exp)
            assert np.array_equal(np.isinf(fwd[:, c]), ~fin) and np.array_equal(fwd[~fin, c], exp[~fin])
            assert_close(fwd[fin, c], exp[fin], rtol=1e-13, what=f"grid forward {ex}")
            assert_close(inv[:, c], qo.cunnane_inverse(cdf, P[:, c], ex, ne), rtol=1e-9, what=f"grid inverse {ex} {ne}")


def test_cunnane_estimator_surface():
    from sklearn.exceptions import NotFittedError

    from skdownscale_amd import CunnaneTransformer

    g = load("g10_cunnane")
    x = g["x0"].reshape(-1, 1)
    t = CunnaneTransformer()
    with pytest.raises(NotFittedError):
        t.transform(x)
    pp = t.fit_transform(x)
    assert pp.shape == x.shape
    assert_close(pp[:, 0], g["fit_transform0"], rtol=1e-13, what="estimator fit_transform")
    assert np.array_equal(t.cdf_.vals, np.sort(x[:, 0])) and t.cdf_.pp.shape == (len(x),)
    back = t.inverse_transform(pp)
    assert_close(back[:, 0], x[:, 0], rtol=1e-12, what="inverse of transform")
    assert_close(t.inverse_transform(g["p0"].reshape(-1, 1))[:, 0], g["inv0_both_10"], rtol=1e-9, what="estimator inverse")
    with pytest.raises(AttributeError):  # the reference fails the same way beyond an extended tail (quantile.py:497)
        t.transform(np.array([[x.max() + 1.0]]))
    assert CunnaneTransformer(extrapolate=None).fit(x).transform(np.array([[x.max() + 1.0]]))[0, 0] == t.cdf_.pp[-1]
    t2 = pickle.loads(pickle.dumps(t))
    assert np.array_equal(t2.inverse_transform(pp), back)
    with pytest.raises(ValueError, match="single feature"):
        CunnaneTransformer().fit(np.zeros((5, 2)))
    with pytest.raises(ValueError, match="unknown value for extrapolate"):
        CunnaneTransformer(extrapolate="sideways").fit(x)


def test_pointwise_transformers_are_batched():
    """PointWiseDownscaler.transform / inverse_transform (core.py:340-403) with CunnaneTransformer / QuantileMapper: one
    engine launch for the grid, same numbers as one estimator per cell, masked cells NaN, the reference's failures."""
    from skdownscale_amd import CunnaneTransformer, GridArray, PointWiseDownscaler, QuantileMapper

    rng = np.random.default_rng(21)
    dims = ("time", "y", "x")
    X = 10 + 3 * rng.standard_normal((300, 3, 4))
    X[0, 2, 1] = np.nan  # masked cell (core.py:35-37)
    Xn = X + 0.25
    ok = np.ones((3, 4), bool)
    ok[2, 1] = False
    for model in (CunnaneTransformer(extrapolate=None), QuantileMapper()):
        pw = PointWiseDownscaler(model)
        pw.fit(GridArray(X, dims))
        assert pw._models.kind in ("cunnane", "qmapper")
        out = pw.transform(GridArray(Xn, dims))
        assert out.dims == ("time", "variable", "y", "x") and out.shape == (300, 1, 3, 4)
        assert np.isnan(out.values[:, 0, 2, 1]).all()
        for iy, ix in zip(*np.nonzero(ok)):
            one = type(model)(**model.get_params()).fit(X[:, iy, ix].reshape(-1, 1)).transform(Xn[:, iy, ix].reshape(-1, 1))
            assert np.array_equal(out.values[:, 0, iy, ix], one[:, 0])
    pw = PointWiseDownscaler(CunnaneTransformer())
    pw.fit(GridArray(X, dims))
    pp = pw.transform(GridArray(X, dims))
    back = pw.inverse_transform(pp)
    assert_close(back.values[:, 0][:, ok], X[:, ok], rtol=1e-12, what="grid inverse of transform")
    with pytest.raises(AttributeError):  # beyond an extended tail (quantile.py:497)
        pw.transform(GridArray(Xn + 20.0, dims))
    with pytest.raises(AttributeError):  # QuantileMapper has no inverse_transform
        p2 = PointWiseDownscaler(QuantileMapper())
        p2.fit(GridArray(X, dims))
        p2.inverse_transform(GridArray(X, dims))
    bad = Xn.copy()
    bad[5, 0, 0] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        pw.transform(GridArray(bad, dims))


@pytest.mark.parametrize("name", ["qmr", "qmr_both", "ecm"])
def test_trend_aware_quantile_mapping_regressor_matches_the_reference(name):
    """g16_trend_aware.npz: TrendAwareQuantileMappingRegressor of the real reference (quantile.py:639-716) around
    QuantileMappingReressor (extrapolate None / 'both') and EquidistantCdfMatcher, fit on 1 400 drifting samples, predict on
    1 900: detrending, mapping, trend line and mean change all through the engine's estimators."""
    import pandas as pd

    from skdownscale_amd import (EquidistantCdfMatcher, QuantileMappingReressor, TrendAwareQuantileMappingRegressor, synth)

    g = load("g16_trend_aware")
    T, Tp, C = int(g["T"]), int(g["Tp"]), int(g["C"])
    index = pd.date_range(str(g["start"]), periods=T)
    index_p = pd.date_range(str(g["pstart"]), periods=Tp)
    cells = np.arange(C)
    X = synth.tas_field("X_hist", 7 + 16, index, cells, 1000) + 3e-3 * np.arange(T)[:, None]
    y = synth.tas_field("y_obs", 7 + 16, index, cells, 1000) + 2e-3 * np.arange(T)[:, None]
    Xp = synth.tas_field("X_fut", 7 + 16, index_p, cells, 1000) + 4e-3 * np.arange(Tp)[:, None]
    make = {"qmr": lambda: QuantileMappingReressor(), "qmr_both": lambda: QuantileMappingReressor(extrapolate="both"),
            "ecm": lambda: EquidistantCdfMatcher()}[name]
    exp = g[f"out_{name}"]
    for c in range(C):
        m = TrendAwareQuantileMappingRegressor(make()).fit(X[:, c:c + 1], y[:, c:c + 1])
        out = m.predict(Xp[:, c:c + 1])
        assert out.shape == (Tp, 1)
        if name == "qmr_both":  # detrended samples beyond the fitted range carry the +-1e20 node noise of quantile.py:338-346
            err = np.abs(out[:, 0] - exp[:, c])
            assert np.median(err) <= 1e-6 * np.std(exp[:, c]) and (err <= 1e-3 * np.std(exp[:, c])).all(), err.max()
        else:
            assert_close(out[:, 0], exp[:, c], scale=float(np.std(exp[:, c])), what=f"trend-aware {name} cell {c}")
    # DataFrame inputs (what the reference's class expects) and a given trend transformer
    from skdownscale_amd import LinearTrendTransformer

    m = TrendAwareQuantileMappingRegressor(make(), LinearTrendTransformer()).fit(pd.DataFrame(X[:, :1], index=index), pd.DataFrame(y[:, :1], index=index))
    out = m.predict(pd.DataFrame(Xp[:, :1], index=index_p))
    if name != "qmr_both":
        assert_close(out[:, 0], exp[:, 0], scale=float(np.std(exp[:, 0])), what=f"trend-aware {name} (DataFrames)")


def test_pointwise_transformers_match_the_reference_loop():
    """g18_pointwise_transformers.npz: the reference's per-cell loop (core.py:146-197, 340-425) over a 2 x 3 grid with one
    masked cell -- CunnaneTransformer.transform / inverse_transform, QuantileMapper.transform, and the attributes
    get_attr stacks (CunnaneTransformer.cdf_, BcsdTemperature.y_climo_) -- against the batched PointWiseDownscaler."""
    import pandas as pd

    from skdownscale_amd import BcsdTemperature, CunnaneTransformer, GridArray, PointWiseDownscaler, QuantileMapper, synth

    g = load("g18_pointwise_transformers")
    ny, nx, T, Tz = (int(g[k]) for k in ("ny", "nx", "T", "Tz"))
    C = ny * nx
    cells = np.arange(C)
    index = pd.date_range(str(g["start"]), periods=T)
    X = synth.tas_field("X_hist", 7 + 19, index, cells, 1000)
    y = synth.tas_field("y_obs", 7 + 19, index, cells, 1000)
    Z = synth.tas_field("X_fut", 7 + 19, index[:Tz], cells, 1000)
    for a in (X, y, Z):
        a[:, 4] = np.nan  # the masked cell (first sample NaN: core.py:35-37)
    dims = ("time", "y", "x")
    grid = lambda a, idx: GridArray(a.reshape(len(a), ny, nx), dims, {"time": idx})  # noqa: E731
    masked = np.isnan(g["ct_fwd"][0])
    assert masked.sum() == 1 and masked[4]
    # CunnaneTransformer: transform (inside the fitted range) and inverse_transform of a probability ramp
    pw = PointWiseDownscaler(CunnaneTransformer())
    pw.fit(grid(X, index))
    Zc = np.clip(Z, np.nanmin(X, axis=0), np.nanmax(X, axis=0))
    Zc[:, 4] = np.nan
    fwd = pw.transform(grid(Zc, index[:Tz]))
    assert fwd.dims == ("time", "variable", "y", "x")
    assert_close(fwd.values[:, 0].reshape(Tz, C), g["ct_fwd"], scale=1.0, what="PointWiseDownscaler(CunnaneTransformer).transform")
    pp = np.repeat(g["pp"][:, None], C, axis=1)
    pp[:, 4] = np.nan
    inv = pw.inverse_transform(GridArray(pp.reshape(-1, ny, nx), dims))
    assert_close(inv.values[:, 0].reshape(-1, C), g["ct_inv"], what="PointWiseDownscaler(CunnaneTransformer).inverse_transform")
    # the fitted CDFs of the cells, as get_attr's per-cell estimators hold them (cdf_ is a (pp, vals) pair in the reference too)
    for c in range(C):
        est = pw._cell_model(c, {})
        if masked[c]:
            assert est is None
        else:
            assert_close(np.asarray(est.cdf_.vals).reshape(-1), g["ct_cdf"][:, c], what=f"cdf_.vals of cell {c}")
    # QuantileMapper.transform
    pq = PointWiseDownscaler(QuantileMapper())
    pq.fit(grid(X, index))
    out = pq.transform(grid(Z, index[:Tz]))
    assert_close(out.values[:, 0].reshape(Tz, C), g["qm_fwd"], what="PointWiseDownscaler(QuantileMapper).transform")
    # get_attr of a fitted BCSD grid: the monthly climatologies of every cell
    pb = PointWiseDownscaler(BcsdTemperature())
    pb.fit(grid(X, index), grid(y, index))
    climo = pb.get_attr("y_climo_")
    assert_close(np.asarray(climo.values).reshape(12, C), g["y_climo"], what="get_attr('y_climo_')")


@pytest.mark.parametrize("T,Tp", [(14600, 14600), (3000, 7001), (19000, 5000)])
def test_round6_paths_are_bit_identical_to_the_kernels_they_replace(dev_ctx, monkeypatch, T, Tp):
    """Round 6: the tile-shaped first stage of the fit (qm_tile_runs_kernel + qm_merge_runs_kernel; series of up to 17 408 samples)
    against the staging transpose + workgroup sort (SD_QM_NOTILE), and the plotting positions by the verified correction step
    against the division (SD_QM_DIVIDE) -- switches of the development library -- every model and extrapolate mode, ties, a
    masked and a non-finite cell."""
    rng = np.random.default_rng(T)
    C = 11
    X = np.round(3 * rng.standard_normal((T, C)), 3)
    y = np.round(4 * rng.standard_normal((T, C)), 3) + 2.0
    Xp = np.round(3.5 * rng.standard_normal((Tp, C)), 3) + 1.0
    X[0, 3] = np.nan
    y[17, 5] = np.inf

    def run():
        st = dev_ctx.qm_fit(X, y)
        e = st.export()
        outs = [dev_ctx.qm_predict(st, code, Xp, ex, 10) for code in (0, 1, 2) for ex in (None, "1to1", "min", "both")]
        st.close()
        return e, outs

    e0, o0 = run()
    ok = np.ones(C, bool)
    ok[[3, 5]] = False
    assert np.array_equal(e0["x_sorted"][ok], np.sort(X, axis=0).T[ok]) and np.array_equal(e0["y_sorted"][ok], np.sort(y, axis=0).T[ok])
    assert e0["status"][3] != 0 and e0["status"][5] != 0
    for env in ("SD_QM_NOTILE", "SD_QM_DIVIDE"):
        monkeypatch.setenv(env, "1")
        e1, o1 = run()
        monkeypatch.delenv(env)
        assert np.array_equal(e1["x_sorted"][ok], e0["x_sorted"][ok]) and np.array_equal(e1["y_sorted"][ok], e0["y_sorted"][ok])
        for (a, sa), (b, sb) in zip(o0, o1):
            assert np.array_equal(sa, sb) and np.array_equal(a, b, equal_nan=True), env
