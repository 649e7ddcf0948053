"""GPU parity of the detrended quantile mapping (QuantileMapper(detrend=True), qm_kwargs={'detrend': True} in BCSD,
LinearTrendTransformer) against g15_detrend.npz (from the reference) and the oracle."""
import pickle
import warnings

import numpy as np
import pandas as pd
import pytest

import bcsd_oracle as bo
from _cases import assert_close, detrend_inputs, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from skdownscale_amd.engine import default_context

    return default_context()


@pytest.fixture(scope="module")
def case():
    g = load("g15_detrend")
    return (g,) + detrend_inputs(g)


def test_quantile_mapper_detrend_golden(ctx, case):
    """quantile.py:95-98,128-145: whole series as one group; the reference's own test of this option is an xfail stub
    (test_pointwise_models.py: test_quantile_mapper_detrend), so the golden holds what the reference computes."""
    from skdownscale_amd import PointWiseDownscaler, QuantileMapper, QuantileMapperGridModel
    from skdownscale_amd.core import GridArray

    g, index, index_p, (X, y, Xp), _ = case
    for name, A, B in (("short", X, Xp), ("long", y, Xp)):
        A, B = A[:int(g[f"qmap_{name}_nfit"])], B[:int(g[f"qmap_{name}_n"])]
        gm = QuantileMapperGridModel(ctx, detrend=True).fit(A)
        e = gm.state.export()
        assert e["info"]["detrend"] and not e["info"]["return_anoms"]
        np.testing.assert_allclose(e["y_sorted"].T, g[f"qmap_{name}_cdf"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(e["y_trend"][:, 0, :], g[f"qmap_{name}_line"], rtol=1e-9)
        out, status = gm.transform(B)
        assert (status == 0).all()
        assert_close(out, g[f"qmap_{name}_out"], what=f"QuantileMapper(detrend) grid {name}")
        dout, _ = QuantileMapperGridModel(ctx, detrend=True).fit(ctx.to_device(A)).transform(ctx.to_device(B))
        assert np.array_equal(dout.to_host(), out)
        # the estimator, one cell; pickling keeps the fitted line
        m = QuantileMapper(detrend=True).fit(A[:, :1])
        res = m.transform(B[:, :1])
        assert_close(res[:, 0], g[f"qmap_{name}_out"][:, 0], what=f"QuantileMapper(detrend) estimator {name}")
        np.testing.assert_allclose(np.ravel(m.x_trend_fit_.lr_model_.coef_)[0], g[f"qmap_{name}_line"][0, 0], rtol=1e-9)
        np.testing.assert_allclose(np.ravel(m.x_trend_fit_.lr_model_.intercept_)[0], g[f"qmap_{name}_line"][0, 1], rtol=1e-9)
        m2 = pickle.loads(pickle.dumps(m))
        assert np.array_equal(m2.transform(B[:, :1]), res)
        # the grid driver
        T, Tb, C = len(A), len(B), A.shape[1]
        pw = PointWiseDownscaler(QuantileMapper(detrend=True))
        pw.fit(GridArray(A.reshape(T, C, 1), ("time", "y", "x"), {"time": np.arange(T)}))
        got = pw.transform(GridArray(B.reshape(Tb, C, 1), ("time", "y", "x"), {"time": np.arange(Tb)}))
        assert_close(np.asarray(got.values).reshape(Tb, C), g[f"qmap_{name}_out"], what="PointWiseDownscaler(QuantileMapper(detrend))")
    with pytest.raises(NotImplementedError):
        QuantileMapper(detrend=True, lt_kwargs={"lr_kwargs": {"fit_intercept": False}}).fit(X[:100, :1])


def test_linear_trend_transformer_golden(case):
    from skdownscale_amd import LinearTrendTransformer

    g, index, index_p, (X, y, Xp), _ = case
    lt = LinearTrendTransformer().fit(X[:1200])
    np.testing.assert_allclose(lt.lr_model_.coef_, g["lt_coef"], rtol=1e-9)
    np.testing.assert_allclose(lt.lr_model_.intercept_, g["lt_intercept"], rtol=1e-9)
    assert_close(lt.transform(Xp[:800]), g["lt_transform"], what="LinearTrendTransformer.transform")
    assert_close(lt.inverse_transform(Xp[:800]), g["lt_inverse"], what="LinearTrendTransformer.inverse_transform")
    np.testing.assert_allclose(lt.inverse_transform(lt.transform(Xp[:800])), Xp[:800], rtol=1e-12)
    lt2 = pickle.loads(pickle.dumps(lt))
    assert np.array_equal(lt2.transform(Xp[:800]), lt.transform(Xp[:800]))
    from sklearn.exceptions import NotFittedError

    with pytest.raises(NotFittedError):
        LinearTrendTransformer().transform(X[:10])
    with pytest.raises(ValueError, match="features"):
        lt.transform(X[:10, :2])


@pytest.mark.parametrize("return_anoms", [True, False])
def test_bcsd_detrend_golden(ctx, case, return_anoms):
    """bcsd.py:65-67 with qm_kwargs={'detrend': True}: engine (host fields, resident fields, fused fit+predict, state
    export / import), estimators (fit / predict / pickle), PointWiseDownscaler."""
    from skdownscale_amd import BcsdPrecipitation, BcsdTemperature, PointWiseDownscaler
    from skdownscale_amd.core import GridArray

    g, index, index_p, (X, y, Xp), (P, yP, Pp) = case
    gid, gid_p = bo.month_group_id(index).astype(np.int32), bo.month_group_id(index_p).astype(np.int32)
    tag = f"anoms{int(return_anoms)}"
    for kind, (A, b, Ap), key in ((0, (X, y, Xp), f"tas_out_{tag}"), (1, (P, yP, Pp), f"pr_out_{tag}")):
        st = ctx.bcsd_fit(kind, A, b, gid, 12, return_anoms, detrend=True)
        out, status = ctx.bcsd_predict(st, Ap, gid_p)
        assert (status == 0).all()
        assert_close(out, g[key], what=f"bcsd detrend kind={kind} (host fields)")
        dA, db, dAp = ctx.to_device(A), ctx.to_device(b), ctx.to_device(Ap)
        dout, _ = ctx.bcsd_predict(ctx.bcsd_fit(kind, dA, db, gid, 12, return_anoms, detrend=True), dAp, gid_p)
        assert np.array_equal(dout.to_host(), out)
        fout, _ = ctx.bcsd_fit_predict(kind, dA, db, gid, 12, dAp, gid_p, return_anoms, detrend=True)
        assert_close(fout.to_host(), g[key], what=f"bcsd detrend kind={kind} (fit+predict in one call)")
        e = st.export()
        st2 = ctx.bcsd_import(e)
        out2, _ = ctx.bcsd_predict(st2, Ap, gid_p)
        assert np.array_equal(out2, out)
    # estimators, one cell
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = BcsdTemperature(return_anoms=return_anoms, qm_kwargs={"detrend": True}).fit(pd.DataFrame(X[:, :1], index=index),
                                                                                      pd.DataFrame(y[:, :1], index=index))
        res = m.predict(pd.DataFrame(Xp[:, :1], index=index_p)).values
        assert_close(res[:, 0], g[f"tas_out_{tag}"][:, 0], what="BcsdTemperature(qm_kwargs detrend)")
        line = m.quantile_mappers_[7].x_trend_fit_.lr_model_
        np.testing.assert_allclose([np.ravel(line.coef_)[0], np.ravel(line.intercept_)[0]], g["tas_line_month7"], rtol=1e-9)
        np.testing.assert_allclose(m.quantile_mappers_[7].x_cdf_fit_.cdf_.vals, g["tas_cdf_month7"], rtol=1e-9, atol=1e-9)
        m2 = pickle.loads(pickle.dumps(m))
        assert np.array_equal(m2.predict(pd.DataFrame(Xp[:, :1], index=index_p)).values, res)
        mp = BcsdPrecipitation(return_anoms=return_anoms, qm_kwargs={"detrend": True}).fit(pd.DataFrame(P[:, :1], index=index),
                                                                                         pd.DataFrame(yP[:, :1], index=index))
        assert_close(mp.predict(pd.DataFrame(Pp[:, :1], index=index_p)).values[:, 0], g[f"pr_out_{tag}"][:, 0], what="BcsdPrecipitation")
        # grid driver
        T, Tp, C = X.shape[0], Xp.shape[0], X.shape[1]
        mk = lambda a, idx: GridArray(a.reshape(len(idx), C, 1), ("time", "y", "x"), {"time": idx})  # noqa: E731
        pw = PointWiseDownscaler(BcsdTemperature(return_anoms=return_anoms, qm_kwargs={"detrend": True}))
        pw.fit(mk(X, index), mk(y, index))
        assert_close(np.asarray(pw.predict(mk(Xp, index_p)).values).reshape(Tp, C), g[f"tas_out_{tag}"], what="PointWiseDownscaler")


def test_bcsd_detrend_daily_nasanex_golden(ctx, case):
    from skdownscale_amd import BcsdTemperature
    from skdownscale_amd.groupers import padded_doy_table

    g, index, index_p, (X, y, Xp), _ = case
    order, offsets = padded_doy_table(index)
    gq, gt = np.asarray(index_p.day, dtype=np.int32) - 1, np.asarray(index_p.month, dtype=np.int32) - 1
    st = ctx.bcsd_fit_groups(0, X, y, order, offsets, return_anoms=False, detrend=True)
    out, status = ctx.bcsd_predict_trend(st, Xp, gq, gt, 12)
    assert (status == 0).all()
    assert_close(out, g["tas_nasanex_out"], what="daily_nasa-nex detrend (engine)")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = BcsdTemperature(time_grouper="daily_nasa-nex", return_anoms=False, qm_kwargs={"detrend": True}).fit(
            pd.DataFrame(X[:, :1], index=index), pd.DataFrame(y[:, :1], index=index))
        assert_close(m.predict(pd.DataFrame(Xp[:, :1], index=index_p)).values[:, 0], g["tas_nasanex_out"][:, 0], what="estimator")


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("T,Tp,C", [(1461, 1461, 37), (3000, 1200, 9), (700, 2500, 16), (40, 33, 5)])
def test_bcsd_detrend_vs_oracle(ctx, kind, T, Tp, C):
    """fresh random fields with a drift, ragged cell counts, fit and predict segments of different lengths"""
    rng = np.random.default_rng(T + Tp + C + kind)
    index, index_p = pd.date_range("1990-01-01", periods=T), pd.date_range("1992-06-01", periods=Tp)
    gid, gid_p = bo.month_group_id(index).astype(np.int32), bo.month_group_id(index_p).astype(np.int32)
    if not set(gid_p) <= set(gid):  # short fits do not see every month
        index_p = index[:Tp]
        gid_p = bo.month_group_id(index_p).astype(np.int32)
    slope = rng.normal(0, 2e-3, C)
    X = 12 + 7 * rng.standard_normal((T, C)) + np.arange(T)[:, None] * slope
    y = 10 + 5 * rng.standard_normal((T, C)) + np.arange(T)[:, None] * slope * 0.5
    Xp = 13 + 7 * rng.standard_normal((Tp, C)) + np.arange(Tp)[:, None] * slope * 2
    if kind == 1:
        X, y, Xp = np.abs(X) + 0.1, np.abs(y) + 0.1, np.abs(Xp) + 0.1
    G = int(gid.max()) + 1
    exp, _ = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=G, detrend=True)
    out, status = ctx.bcsd_predict(ctx.bcsd_fit(kind, X, y, gid, G, True, detrend=True), Xp, gid_p)
    assert (status == 0).all()
    assert_close(out, exp, what=f"detrend vs oracle kind={kind} T={T} Tp={Tp} C={C}")
    plain, _ = ctx.bcsd_predict(ctx.bcsd_fit(kind, X, y, gid, G, True), Xp, gid_p)
    assert not np.allclose(plain, out)  # the option does something


def test_bcsd_detrend_long_segments_match_the_oracle(ctx):
    """segments beyond the register-sort kernels (> 2 112 samples) take the workgroup-sort kernels, which detrend as well
    (csrc/sd_bcsd.hip: bcsd_long_*): both kinds, fit longer / shorter than predict, against the pinned oracle; beyond
    19 456 samples the engine still refuses"""
    rng = np.random.default_rng(1)
    for kind, T, Tp in ((0, 3000, 3000), (0, 5000, 2600), (1, 2500, 4100)):
        t, tp = np.arange(T)[:, None], np.arange(Tp)[:, None]
        X = rng.standard_normal((T, 3)) * 3 + 2e-3 * t
        y = rng.standard_normal((T, 3)) * 4 + 1e-3 * t
        Xp = rng.standard_normal((Tp, 3)) * 3 + 3e-3 * tp
        if kind == 1:
            X, y, Xp = np.abs(X) + 0.1, np.abs(y) + 0.1, np.abs(Xp) + 0.1
        gid, gid_p = np.zeros(T, dtype=np.int32), np.zeros(Tp, dtype=np.int32)
        exp, _ = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=1, detrend=True)
        out, status = ctx.bcsd_predict(ctx.bcsd_fit(kind, X, y, gid, 1, True, detrend=True), Xp, gid_p)
        assert (status == 0).all()
        assert_close(out, exp, what=f"long detrended segment kind={kind} T={T} Tp={Tp}")
    big = rng.standard_normal((20000, 2))
    with pytest.raises(NotImplementedError, match="detrended quantile mapping"):
        ctx.bcsd_fit(0, big, big + 1, np.zeros(20000, dtype=np.int32), 1, True, detrend=True)


def test_quantile_mapper_detrend_on_a_40_year_daily_series():
    """g17_detrend_long.npz: QuantileMapper(detrend=True) of the real reference fitted on whole 14 600-sample series (one
    segment per cell: quantile.py:81-147), transform of a series of the same length and of a 16 000-sample one."""
    from skdownscale_amd import QuantileMapper, synth

    g = load("g17_detrend_long")
    C = int(g["C"])
    cells = np.arange(C)
    index, index_p = synth.daily_calendar(14600), synth.daily_calendar(16000)
    X = synth.tas_field("X_hist", 7 + 17, index, cells, 1000) + 1e-4 * np.arange(14600)[:, None] * (1 + cells)
    Xs = synth.tas_field("X_fut", 7 + 17, index, cells, 1000) + 2e-4 * np.arange(14600)[:, None]
    Xl = synth.tas_field("X_fut", 7 + 18, index_p, cells, 1000) + 2e-4 * np.arange(16000)[:, None]
    for name, B in (("same", Xs), ("longer", Xl)):
        for c in range(C):
            m = QuantileMapper(detrend=True).fit(X[:, c:c + 1])
            out = m.transform(B[:, c:c + 1])
            assert_close(out[:, 0], g[f"out_{name}"][:, c], what=f"QuantileMapper(detrend=True) 14 600 samples, {name}, cell {c}")
            line = np.array([float(np.ravel(m.x_trend_fit_.lr_model_.coef_)[0]), float(np.ravel(m.x_trend_fit_.lr_model_.intercept_)[0])])
            assert np.allclose(line, g[f"line_{name}"][c], rtol=1e-9, atol=1e-12)
