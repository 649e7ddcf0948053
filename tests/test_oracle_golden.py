"""CPU: the NumPy oracle reproduces the golden vectors generated from the real reference."""
import numpy as np
import pytest

import analog_oracle as ao
import bcsd_oracle as bo
from _cases import analog_inputs, assert_close, load, month_gid, pr_inputs, tas_inputs


@pytest.mark.parametrize("name", ["g1_tas_same", "g2_tas_long", "g2_tas_short", "g1_tas_small"])
def test_bcsd_temperature(name):
    g = load(name)
    index, index_p, X, y, Xp = tas_inputs(g)
    for key, ra in (("out_anoms", True), ("out_abs", False)):
        out, st = bo.pointwise_fit_predict(bo.TAS, X, y, Xp, month_gid(index), month_gid(index_p), return_anoms=ra)
        assert_close(out, g[key], what=f"{name}/{key}")
        assert np.array_equal(st, g["status"])


def test_bcsd_temperature_ties():
    g = load("g_tas_ties")
    index, index_p, X, y, Xp = tas_inputs(g)
    X, y, Xp = np.round(X * 2) / 2, np.round(y * 2) / 2, np.round(Xp * 2) / 2
    out, _ = bo.pointwise_fit_predict(bo.TAS, X, y, Xp, month_gid(index), month_gid(index_p))
    assert_close(out, g["out_anoms"], what="ties")


@pytest.mark.parametrize("name", ["g3_pr_same", "g3_pr_long", "g3_pr_small"])
def test_bcsd_precipitation(name):
    g = load(name)
    index, index_p, X, y, Xp = pr_inputs(g)
    for key, ra in (("out_anoms", True), ("out_abs", False)):
        out, st = bo.pointwise_fit_predict(bo.PR, X, y, Xp, month_gid(index), month_gid(index_p), return_anoms=ra)
        assert_close(out, g[key], what=f"{name}/{key}")
        assert np.array_equal(st, g["status"])


def test_bcsd_precipitation_bad_climatology():
    g = load("g3_pr_badclimo")
    index, _, X, y, Xp = pr_inputs(g)
    y[np.asarray(index.month) == 7, 1] = 0.0
    out, st = bo.pointwise_fit_predict(bo.PR, X, y, Xp, month_gid(index), month_gid(index), return_anoms=True)
    assert np.array_equal(st, g["status"]) and st[1] == bo.STATUS_BAD_CLIMO
    assert "Invalid value in target climatology" in str(g["errors"][1])
    assert_close(out, g["out_anoms"], what="badclimo/anoms")
    out, st = bo.pointwise_fit_predict(bo.PR, X, y, Xp, month_gid(index), month_gid(index), return_anoms=False)
    assert np.array_equal(st, g["status_abs"])
    assert_close(out, g["out_abs"], what="badclimo/abs")


def test_ndarray_input_fabricated_index():
    g = load("g4_ndarray")
    gid, gid_p = bo.fabricated_group_id(100), bo.fabricated_group_id(150)
    out, _ = bo.pointwise_fit_predict(bo.TAS, g["X"], g["y"], g["Xp"], gid, gid_p)
    assert_close(out[:, 0], g["out_tas"], what="ndarray tas")
    out, _ = bo.pointwise_fit_predict(bo.TAS, g["X"], g["y"], g["X"], gid, gid)
    assert_close(out[:, 0], g["out_tas_same"], what="ndarray tas same")
    out, _ = bo.pointwise_fit_predict(bo.PR, g["P"], g["yP"], g["PP"], gid, gid_p)
    assert_close(out[:, 0], g["out_pr"], what="ndarray pr")
    assert any("making one up" in str(m) for m in g["warn_msgs"])


def test_masked_and_nan_cells():
    g = load("g7_masked")
    index, index_p, X, y, Xp = tas_inputs(g)
    X[0, 1] = np.nan
    X[0, 4] = np.nan
    y[0, 1] = np.nan
    out, st = bo.pointwise_fit_predict(bo.TAS, X, y, Xp, month_gid(index), month_gid(index_p))
    assert np.array_equal(st, g["status"])
    assert_close(out, g["out_anoms"], what="masked")
    X[100, 2] = np.nan
    _, st = bo.pointwise_fit_predict(bo.TAS, X, y, Xp, month_gid(index), month_gid(index_p))
    assert np.array_equal(st, g["status_nan_inside"]) and st[2] == bo.STATUS_NONFINITE
    assert "NaN" in str(g["errors_nan_inside"][2])


def test_reference_own_tests():
    """test_pointwise_models.py:81-90 (QuantileMapper identity) and the 365-day smoke inputs."""
    g = load("g8_reference_tests")
    exp = g["qm_expected"][:, 0]
    q = bo.qm_segment(exp + 2, np.sort(exp))
    np.testing.assert_almost_equal(q, exp)  # the reference's own assertion
    np.testing.assert_almost_equal(q, g["qm_actual"][:, 0])
    import pandas as pd

    index = pd.date_range("2019-01-01", periods=365)
    x = g["sine365"]
    out, _ = bo.pointwise_fit_predict(bo.TAS, x[:, None], x[:, None] + 2, x[:, None], month_gid(index), month_gid(index))
    assert_close(out[:, 0], g["bcsd_tas_sine365"], what="sine365")
    x = g["pr_random365"]
    out, _ = bo.pointwise_fit_predict(bo.PR, x[:, None], x[:, None] + 2, x[:, None], month_gid(index), month_gid(index))
    assert_close(out[:, 0], g["bcsd_pr_random365"], what="pr365")


@pytest.mark.parametrize("F", [1, 3])
def test_analog_knn_bit_exact(F):
    g = load(f"g5_analog_F{F}")
    X, y, Xq = analog_inputs(g)
    for k in (1, 30):
        for c in range(X.shape[2]):
            d, i = ao.knn(X[:, :, c], Xq[:, :, c], k)
            assert np.array_equal(i, g[f"inds_k{k}"][:, :, c])  # bit-exact index selection
            assert np.array_equal(d, g[f"dist_k{k}"][:, :, c])  # and bit-exact distances


@pytest.mark.parametrize("F", [1, 3])
@pytest.mark.parametrize("kind", ["best_analog", "sample_analogs", "weight_analogs", "mean_analogs"])
def test_pure_analog(F, kind):
    g = load(f"g5_analog_F{F}")
    X, y, Xq = analog_inputs(g)
    for k in (1, 30):
        for thresh, tt in ((None, "none"), (0.0, "t0")):
            tag = f"{kind}_k{k}_{tt}"
            samp = g["samp_" + tag] if kind == "sample_analogs" else None
            out = ao.pointwise_analog(X, y, Xq, k, ao.KIND_NAMES[kind], thresh, samp)
            assert_close(out, g["out_" + tag], what=tag)


@pytest.mark.parametrize("F", [1, 3])
def test_analog_regression(F):
    g = load(f"g5_analog_F{F}")
    X, y, Xq = analog_inputs(g)
    Tr = int(g["Tr"])
    out = ao.pointwise_analog(X, y, Xq[:Tr], 30, None, regression=True)
    assert_close(out, g["out_analogreg_k30"], what="analogreg")


def test_quantile_mapping_regressors_match_reference():
    """QuantileMappingReressor / EquidistantCdfMatcher restatement (oracle/qm_oracle.py) vs g9_qm.npz: every
    extrapolate mode, n_endpoints 10 / 2, equal / longer / shorter predict series.

    extrapolate in (None, '1to1') is pinned on every sample.  For 'min' / 'max' / 'both' the reference interpolates
    across synthetic end points at +-1e20 (quantile.py:17-18, 338-346): samples that leave the fitted range come out
    of a cancellation with ~1e5 of absolute rounding noise (they depend on the last bits of LAPACK's least squares),
    so only the samples inside the fitted range are compared there -- parity of those tails is unpinned."""
    import qm_oracle as qo

    g = load("g9_qm")
    for case in range(3):
        X, y, Xp = g[f"X{case}"], g[f"y{case}"], g[f"Xp{case}"]
        n, m = len(X), len(Xp)
        inside_x = (Xp >= X.min(axis=0)) & (Xp <= X.max(axis=0))
        pp_new = qo.plotting_positions(m)[np.argsort(np.argsort(Xp, axis=0, kind="stable"), axis=0, kind="stable")]
        pp_fit = qo.plotting_positions(n)
        inside_p = (pp_new >= pp_fit[0]) & (pp_new <= pp_fit[-1])
        for ex in qo.EXTRAPOLATE:
            pinned = ex in (None, "1to1")
            for ne in (10, 2):
                got = qo.pointwise_qm("qmr", X, y, Xp, ex, ne)
                exp = g[f"out{case}_qmr_{ex}_{ne}"]
                sel = np.ones_like(inside_x) if pinned else inside_x
                assert np.isfinite(got).all()
                assert_close(np.where(sel, got, 0.0), np.where(sel, exp, 0.0), scale=np.std(exp[sel]), what=f"qmr {case} {ex} {ne}")
                for kind in ("difference", "ratio"):
                    got = qo.pointwise_qm("ecm", X, y, Xp, ex, ne, kind)
                    exp = g[f"out{case}_ecm_{kind}_{ex}_{ne}"]
                    sel = np.ones_like(inside_p) if pinned else inside_p
                    assert_close(np.where(sel, got, 0.0), np.where(sel, exp, 0.0), scale=np.std(exp[sel]),
                                 what=f"ecm {kind} {case} {ex} {ne}")
    x = np.arange(1, 22.0)  # the reference's test_EquidistantCdfMatcher: exact
    assert np.array_equal(qo.ecm_predict(qo.qm_fit(x, x + 3), x + 2, "difference"), g["reftest_difference"])
    assert np.array_equal(qo.ecm_predict(qo.qm_fit(x, x + 3), x * 2, "ratio"), g["reftest_ratio"])


def test_cunnane_oracle_matches_golden():
    """CunnaneTransformer restatement (oracle/qm_oracle.py) vs g10_cunnane.npz (the real reference): transform inside
    the fitted range for every mode, beyond it for the clamping modes, fit_transform, and inverse_transform with the
    least-squares tails (n_endpoints 10 / 3, and a 6-sample fit shorter than n_endpoints)."""
    import qm_oracle as qo

    g = load("g10_cunnane")
    for case in range(3):
        cdf = qo.cunnane_fit(g[f"x{case}"])
        for ex in qo.EXTRAPOLATE:
            for ne in (10, 3):
                got = qo.cunnane_transform(cdf, g[f"inside{case}"], ex)
                assert np.array_equal(got, g[f"fwd{case}_{ex}_{ne}"]), (case, ex)
                if ex in (None, "1to1"):
                    assert np.array_equal(qo.cunnane_transform(cdf, g[f"outside{case}"], ex), g[f"fwd_out{case}_{ex}_{ne}"])
                assert_close(qo.cunnane_inverse(cdf, g[f"p{case}"], ex, ne), g[f"inv{case}_{ex}_{ne}"], rtol=1e-12,
                             what=f"cunnane inverse {case} {ex} {ne}")
        assert np.array_equal(qo.cunnane_transform(cdf, g[f"x{case}"], "both"), g[f"fit_transform{case}"])


def test_pure_regression_oracle_matches_golden():
    """PureRegression(thresh=None) restatement (oracle/analog_oracle.py) vs g11_pure_regression.npz: 1 / 3 features,
    two collinear features (minimum-norm coefficients), a noise-free target."""
    import analog_oracle as ao

    g = load("g11_pure_regression")
    for case in range(4):
        out, coef, icpt, err = ao.pure_regression(g[f"X{case}"], g[f"y{case}"], g[f"Xq{case}"])
        assert_close(out[:, 0], g[f"out{case}"][:, 0], rtol=1e-9, what=f"pure regression pred {case}")
        assert np.array_equal(out[:, 1], g[f"out{case}"][:, 1])
        assert_close(coef, g[f"coef{case}"], rtol=1e-9, what="coef")
        assert abs(icpt - float(g[f"intercept{case}"])) <= 1e-9 * (1 + abs(icpt))
        assert abs(err - float(g[f"fit_error{case}"])) <= 1e-9 * (1 + err) + 1e-12


def test_bcsd_daily_nasanex_and_separate_trend_grouper():
    """g13_nasanex.npz (from the reference): fit on the 366 padded day-of-year groups (bcsd.py:36-38,50-55, groupers.py:19-89),
    predict by day-of-month keys with the rolling mean over months (bcsd.py:247-267, return_anoms=False), and the monthly
    model with a day-of-month climate-trend grouper."""
    from _cases import nasanex_inputs

    g = load("g13_nasanex")
    for case in (0, 1):
        index, index_p, (X, y, Xp), (P, yP, Pp) = nasanex_inputs(g, case)
        table = bo.padded_doy_table(index)
        gq, gt = np.asarray(index_p.day) - 1, np.asarray(index_p.month) - 1
        for c in range(X.shape[1]):
            st, status = bo.bcsd_fit_cell(bo.TAS, X[:, c], y[:, c], None, table=table, return_anoms=False)
            assert status == 0
            np.testing.assert_allclose(st["y_climo"], g[f"y_climo{case}"][:, c], rtol=1e-12)
            np.testing.assert_allclose(st["x_climo"], g[f"x_climo{case}"][:, c], rtol=1e-12)
            for k in (1, 59, 60, 200, 366):
                assert np.array_equal(st["ys"][st["off"][k - 1]:st["off"][k]], g[f"cdf{case}_{k}"][:, c])
            out, _ = bo.bcsd_predict_trend_cell(st, Xp[:, c], gq, gt, return_anoms=False)
            assert_close(out, g[f"tas_out{case}"][:, c], what=f"daily tas case {case} cell {c}")
            stm, _ = bo.bcsd_fit_cell(bo.TAS, X[:, c], y[:, c], bo.month_group_id(index))
            out, _ = bo.bcsd_predict_trend_cell(stm, Xp[:, c], bo.month_group_id(index_p), gq)
            assert_close(out, g[f"tas_daytrend_out{case}"][:, c], what=f"day-of-month trend case {case} cell {c}")
            stp, status = bo.bcsd_fit_cell(bo.PR, P[:, c], yP[:, c], None, table=table, return_anoms=False)
            np.testing.assert_allclose(stp["y_climo"], g[f"pr_y_climo{case}"][:, c], rtol=1e-12)
            out, _ = bo.bcsd_predict_cell(stp, Pp[:, c], gq, return_anoms=False)
            assert_close(out, g[f"pr_out{case}"][:, c], what=f"daily pr case {case} cell {c}")


def test_detrended_quantile_mapping():
    """g15_detrend.npz (from the reference): QuantileMapper(detrend=True) on whole series (quantile.py:95-98,128-145), and
    BCSD with qm_kwargs={'detrend': True} per month / per padded day-of-year group (bcsd.py:65-67)."""
    from _cases import detrend_inputs

    g = load("g15_detrend")
    index, index_p, (X, y, Xp), (P, yP, Pp) = detrend_inputs(g)
    C = X.shape[1]
    for name, A, B in (("short", X, Xp), ("long", y, Xp)):
        A, B = A[:int(g[f"qmap_{name}_nfit"])], B[:int(g[f"qmap_{name}_n"])]
        for c in range(C):
            st, _ = bo.bcsd_fit_cell(bo.PR, None, A[:, c], np.zeros(len(A), dtype=int), G=1, return_anoms=False, detrend=True)
            np.testing.assert_allclose(st["ys"], g[f"qmap_{name}_cdf"][:, c], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(st["y_trend_icpt"][0], g[f"qmap_{name}_line"][c, 1], rtol=1e-10)
            out, _ = bo.bcsd_predict_cell(st, B[:, c], np.zeros(len(B), dtype=int), return_anoms=False)
            assert_close(out, g[f"qmap_{name}_out"][:, c], what=f"QuantileMapper(detrend) {name} cell {c}")
    gid, gid_p = bo.month_group_id(index), bo.month_group_id(index_p)
    for ra in (True, False):
        out, _ = bo.pointwise_fit_predict(bo.TAS, X, y, Xp, gid, gid_p, return_anoms=ra, detrend=True)
        assert_close(out, g[f"tas_out_anoms{int(ra)}"], what=f"tas detrend anoms={ra}")
        out, _ = bo.pointwise_fit_predict(bo.PR, P, yP, Pp, gid, gid_p, return_anoms=ra, detrend=True)
        assert_close(out, g[f"pr_out_anoms{int(ra)}"], what=f"pr detrend anoms={ra}")
    table = bo.padded_doy_table(index)
    for c in range(C):
        st, _ = bo.bcsd_fit_cell(bo.TAS, X[:, c], y[:, c], None, table=table, return_anoms=False, detrend=True)
        out, _ = bo.bcsd_predict_trend_cell(st, Xp[:, c], np.asarray(index_p.day) - 1, np.asarray(index_p.month) - 1, return_anoms=False)
        assert_close(out, g["tas_nasanex_out"][:, c], what=f"daily_nasa-nex detrend cell {c}")


PROB_TIGHT = 1e-6    # exceedance probability vs the reference's objective solved tightly (logistic_kwargs tol=1e-12)
PROB_DEFAULT = 1e-3  # ... vs the reference's default LogisticRegression: its L-BFGS stops at tol=1e-4, within ~2e-4 of the optimum


def test_thresholded_regressions():
    """g14_thresholded_regressions.npz (from the reference): AnalogRegression(thresh) (gard.py:201-219), PureRegression(thresh)
    (gard.py:416-470) incl. the cell that drops its threshold, and PureRegression on features of very different scales."""
    g = load("g14_thresholded_regressions")
    for case in (0, 1):
        X, y, Xq = g[f"ar_X{case}"], g[f"ar_y{case}"], g[f"ar_Xq{case}"]
        out = ao.pointwise_analog(X, y, Xq, int(g[f"ar_k{case}"]), None, thresh=float(g[f"ar_thresh{case}"]), regression=True)
        for name, tol in (("tight", PROB_TIGHT), ("default", PROB_DEFAULT)):
            exp = g[f"ar_out{case}_{name}"]
            assert_close(out[:, 0], exp[:, 0], what=f"analog regression pred case {case}")
            assert_close(out[:, 2], exp[:, 2], what=f"analog regression error case {case}")
            assert np.abs(out[:, 1] - exp[:, 1]).max() <= tol, (case, name)
    for case in (0, 1):
        X, y, Xq, thresh = g[f"pr_X{case}"], g[f"pr_y{case}"], g[f"pr_Xq{case}"], float(g[f"pr_thresh{case}"])
        for c in range(X.shape[2]):
            out = ao.pure_regression_thresh(X[:, :, c], y[:, c], Xq[:, :, c], thresh)[0]
            for name, tol in (("tight", PROB_TIGHT), ("default", PROB_DEFAULT)):
                exp = g[f"pr_out{case}_{name}"][:, :, c]
                assert_close(out[:, [0, 2]], exp[:, [0, 2]], what=f"pure regression case {case} cell {c}")
                assert np.abs(out[:, 1] - exp[:, 1]).max() <= tol, (case, c, name)
            assert bool(g[f"pr_dropped{case}"][c]) == bool((y[:, c] > thresh).all())
    out, coef, icpt, _ = ao.pure_regression(g["ms_X"], g["ms_y"], g["ms_Xq"])
    assert_close(out, g["ms_out"], what="mixed-scale features")
    np.testing.assert_allclose(coef, g["ms_coef"], rtol=1e-9)


def test_trend_aware_regressor_oracle_matches_golden():
    """g16_trend_aware.npz (the reference's TrendAwareQuantileMappingRegressor around QMR / QMR 'both' / ECM) against the
    NumPy restatement oracle/qm_oracle.py: trend_aware_predict."""
    import pandas as pd

    import qm_oracle as qo
    from skdownscale_amd import synth

    g = load("g16_trend_aware")
    T, Tp, C = int(g["T"]), int(g["Tp"]), int(g["C"])
    index, index_p = pd.date_range(str(g["start"]), periods=T), pd.date_range(str(g["pstart"]), periods=Tp)
    cells = np.arange(C)
    X = synth.tas_field("X_hist", 7 + 16, index, cells, 1000) + 3e-3 * np.arange(T)[:, None]
    y = synth.tas_field("y_obs", 7 + 16, index, cells, 1000) + 2e-3 * np.arange(T)[:, None]
    Xp = synth.tas_field("X_fut", 7 + 16, index_p, cells, 1000) + 4e-3 * np.arange(Tp)[:, None]
    for name, model, ex in (("qmr", "qmr", None), ("ecm", "ecm", None)):
        for c in range(C):
            out = qo.trend_aware_predict(model, X[:, c], y[:, c], Xp[:, c], extrapolate=ex)
            assert_close(out, g[f"out_{name}"][:, c], what=f"trend-aware oracle {name} cell {c}")
    for c in range(C):  # 'both': beyond the fitted range the +-1e20 nodes leave ~1e-4 of absolute noise in the reference's own output
        out = qo.trend_aware_predict("qmr", X[:, c], y[:, c], Xp[:, c], extrapolate="both")
        err = np.abs(out - g["out_qmr_both"][:, c])
        assert np.median(err) <= 1e-6 * np.std(out) and err.max() <= 1e-3 * np.std(out)


def test_detrended_mapping_of_a_40_year_series_oracle_matches_golden():
    """g17_detrend_long.npz (QuantileMapper(detrend=True) fitted on whole 14 600-sample series) against oracle/bcsd_oracle.py
    with the whole series as one group."""
    from skdownscale_amd import synth

    g = load("g17_detrend_long")
    C = int(g["C"])
    cells = np.arange(C)
    index, index_p = synth.daily_calendar(14600), synth.daily_calendar(16000)
    X = synth.tas_field("X_hist", 7 + 17, index, cells, 1000) + 1e-4 * np.arange(14600)[:, None] * (1 + cells)
    Xs = synth.tas_field("X_fut", 7 + 17, index, cells, 1000) + 2e-4 * np.arange(14600)[:, None]
    Xl = synth.tas_field("X_fut", 7 + 18, index_p, cells, 1000) + 2e-4 * np.arange(16000)[:, None]
    for name, B in (("same", Xs), ("longer", Xl)):
        for c in range(C):
            st, _ = bo.bcsd_fit_cell(bo.PR, None, X[:, c], np.zeros(14600, dtype=int), G=1, return_anoms=False, detrend=True)
            np.testing.assert_allclose(st["y_trend_icpt"][0], g[f"line_{name}"][c, 1], rtol=1e-9)
            out, _ = bo.bcsd_predict_cell(st, B[:, c], np.zeros(len(B), dtype=int), return_anoms=False)
            assert_close(out, g[f"out_{name}"][:, c], what=f"QuantileMapper(detrend) 14 600 samples {name} cell {c}")


def test_bcsd_qt_kwargs_oracle_matches_golden():
    """g19_qt_kwargs.npz (BcsdTemperature / BcsdPrecipitation with qm_kwargs={'qt_kwargs': ...} in the real reference; predict
    series longer than the fit series, so the tails of the fitted inverse CDFs are reached): the oracle with `extrapolate` /
    `n_endpoints` (quantile.py:523-545) reproduces every variant; `alpha` / `beta` have no parameter in the oracle because they
    have no effect in the reference (tas6 == the defaults, asserted when the golden was made)."""
    from skdownscale_amd import synth

    g = load("g19_qt_kwargs")
    T, Tp, C = int(g["T"]), int(g["Tp"]), int(g["C"])
    index, index_p = synth.daily_calendar(T), synth.daily_calendar(Tp)
    cells = np.arange(C)
    tas = [synth.tas_field(n, int(g["seed"]), i, cells, int(g["c_full"])) for n, i in (("X_hist", index), ("y_obs", index), ("X_fut", index_p))]
    pr = [synth.pr_field(n, int(g["seed"]), t, cells, int(g["c_full"])) for n, t in (("X_hist", T), ("y_obs", T), ("X_fut", Tp))]
    variants = [dict(n_endpoints=5), dict(n_endpoints=3, extrapolate="both"), dict(extrapolate="min"), dict(extrapolate="max"),
                dict(extrapolate=None), dict(extrapolate="1to1"), dict(), dict(n_endpoints=40)]
    assert len(variants) == int(g["n_variants"])
    for kind, (X, y, Xp), n, key in ((bo.TAS, tas, len(variants), "tas"), (bo.PR, pr, int(g["n_pr"]), "pr")):
        for i, kw in enumerate(variants[:n]):
            out, st = bo.pointwise_fit_predict(kind, X, y, Xp, month_gid(index), month_gid(index_p), **kw)
            assert (st == 0).all()
            assert_close(out, g[f"{key}{i}"], what=f"qt_kwargs {kw} {key}")
    assert not np.allclose(g["tas2"], g["tas3"])  # (the variants do differ)


def test_pointwise_transformer_loop_oracle_matches_golden():
    """g18_pointwise_transformers.npz (the reference's per-cell loop: CunnaneTransformer transform / inverse_transform,
    QuantileMapper.transform, BcsdTemperature.y_climo_) against the NumPy restatements."""
    import pandas as pd

    import qm_oracle as qo
    from skdownscale_amd import synth

    g = load("g18_pointwise_transformers")
    ny, nx, T, Tz = (int(g[k]) for k in ("ny", "nx", "T", "Tz"))
    C = ny * nx
    index = pd.date_range(str(g["start"]), periods=T)
    cells = np.arange(C)
    X = synth.tas_field("X_hist", 7 + 19, index, cells, 1000)
    y = synth.tas_field("y_obs", 7 + 19, index, cells, 1000)
    Z = synth.tas_field("X_fut", 7 + 19, index[:Tz], cells, 1000)
    gid = bo.month_group_id(index)
    for c in range(C):
        if c == 4:  # the masked cell of the golden
            assert np.isnan(g["ct_fwd"][:, c]).all() and np.isnan(g["y_climo"][:, c]).all()
            continue
        cdf = qo.cunnane_fit(X[:, c])
        zc = np.clip(Z[:, c], X[:, c].min(), X[:, c].max())
        assert_close(qo.cunnane_transform(cdf, zc), g["ct_fwd"][:, c], scale=1.0, what=f"cunnane transform cell {c}")
        assert_close(qo.cunnane_inverse(cdf, g["pp"]), g["ct_inv"][:, c], what=f"cunnane inverse cell {c}")
        st, _ = bo.bcsd_fit_cell(bo.PR, None, X[:, c], np.zeros(T, dtype=int), G=1, return_anoms=False)
        out, _ = bo.bcsd_predict_cell(st, Z[:, c], np.zeros(Tz, dtype=int), return_anoms=False)
        assert_close(out, g["qm_fwd"][:, c], what=f"QuantileMapper.transform cell {c}")
        stt, _ = bo.bcsd_fit_cell(bo.TAS, X[:, c], y[:, c], gid)
        np.testing.assert_allclose(stt["y_climo"], g["y_climo"][:, c], rtol=1e-12)


def test_quantile_mapper_qt_kwargs_oracle_matches_golden():
    """g20_qm_qt_kwargs.npz (stand-alone QuantileMapper(qt_kwargs=...) of the real reference, transform series longer than the
    fitted one, fits of 800 and 3 000 samples): the oracle's qm_segment with the same `extrapolate` / `n_endpoints`
    (quantile.py:92, 136, 418-431, 523-545)."""
    import ast

    import bcsd_oracle

    g = load("g20_qm_qt_kwargs")
    for tag in ("s", "l"):
        fit, new = g[f"{tag}_fit"][:, 0], g[f"{tag}_new"][:, 0]
        for i in range(int(g["n_variants"])):
            kw = dict(ast.literal_eval(str(g["variants"][i])))
            got = bcsd_oracle.qm_segment(new, np.sort(fit), None, kw.get("extrapolate", "both"), kw.get("n_endpoints", 10))
            exp = g[f"{tag}{i}"][:, 0]
            assert np.allclose(got, exp, rtol=1e-12, atol=1e-12), (tag, kw, np.abs(got - exp).max())
