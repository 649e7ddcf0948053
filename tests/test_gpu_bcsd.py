"""GPU parity: HIP BCSD path (through the C ABI) vs golden vectors from the reference and vs the oracle."""
import warnings

import numpy as np
import pandas as pd
import pytest

import bcsd_oracle as bo
from _cases import assert_close, load, month_gid, pr_inputs, tas_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from skdownscale_amd.engine import default_context

    return default_context()


def run_engine(ctx, kind, X, y, Xp, gid, gid_p, G=12, return_anoms=True, resident=False):
    if resident:
        dX, dy, dXp = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xp)
        st = ctx.bcsd_fit(kind, dX, dy, gid, G, return_anoms)
        out, status = ctx.bcsd_predict(st, dXp, gid_p)
        return out.to_host(), status
    st = ctx.bcsd_fit(kind, X, y, gid, G, return_anoms)
    return ctx.bcsd_predict(st, Xp, gid_p)


@pytest.mark.parametrize("name", ["g1_tas_same", "g2_tas_long", "g2_tas_short", "g1_tas_small"])
@pytest.mark.parametrize("resident", [False, True])
def test_bcsd_temperature_golden(ctx, name, resident):
    g = load(name)
    index, index_p, X, y, Xp = tas_inputs(g)
    for key, ra in (("out_anoms", True), ("out_abs", False)):
        out, st = run_engine(ctx, 0, X, y, Xp, month_gid(index), month_gid(index_p), return_anoms=ra, resident=resident)
        assert_close(out, g[key], what=f"{name}/{key}")  # 1e-6 relative, north_star tolerance
        assert np.array_equal(st, g["status"])


def test_bcsd_temperature_ties(ctx):
    g = load("g_tas_ties")
    index, index_p, X, y, Xp = tas_inputs(g)
    X, y, Xp = np.round(X * 2) / 2, np.round(y * 2) / 2, np.round(Xp * 2) / 2
    out, _ = run_engine(ctx, 0, X, y, Xp, month_gid(index), month_gid(index_p))
    assert_close(out, g["out_anoms"], what="ties")


@pytest.mark.parametrize("name", ["g3_pr_same", "g3_pr_long", "g3_pr_small"])
def test_bcsd_precipitation_golden(ctx, name):
    g = load(name)
    index, index_p, X, y, Xp = pr_inputs(g)
    for key, ra in (("out_anoms", True), ("out_abs", False)):
        out, st = run_engine(ctx, 1, X, y, Xp, month_gid(index), month_gid(index_p), return_anoms=ra)
        assert_close(out, g[key], what=f"{name}/{key}")
        assert np.array_equal(st, g["status"])


def test_bcsd_precipitation_bad_climatology(ctx):
    g = load("g3_pr_badclimo")
    index, _, X, y, Xp = pr_inputs(g)
    y[np.asarray(index.month) == 7, 1] = 0.0
    out, st = run_engine(ctx, 1, X, y, Xp, month_gid(index), month_gid(index), return_anoms=True)
    assert np.array_equal(st, g["status"])
    assert_close(out, g["out_anoms"], what="badclimo")
    out, st = run_engine(ctx, 1, X, y, Xp, month_gid(index), month_gid(index), return_anoms=False)
    assert np.array_equal(st, g["status_abs"])
    assert_close(out, g["out_abs"], what="badclimo/abs")


@pytest.mark.parametrize("p_dry", [0.0, 0.3, 0.6, 0.95, 1.0])
def test_bcsd_precipitation_dry_fraction_sweep(ctx, p_dry):
    """BcsdPrecipitation fit + predict on whole-lane months (40-year daily series: the kernel that sorts only the wet days,
    bcsd_fxc_kernel) for series that are all wet, mostly wet (more wet days than its narrow sort holds: the segments come back on
    the second list and take the K-wide kernel), zero-inflated, almost dry and entirely dry, per cell mixed: against the NumPy
    oracle (bcsd.py:115-185, quantile.py:488: every zero takes the largest rank among the zeros)."""
    import bcsd_oracle
    from skdownscale_amd import synth

    T, C = 14600, 19  # (a partly filled last tile)
    index = synth.daily_calendar(T)
    gid = month_gid(index)
    cells = np.arange(C)
    fields = []
    for name, stream in (("X_hist", 10), ("y_obs", 11), ("X_fut", 12)):
        f = synth.fill(synth.PRECIP, 77, stream, np.arange(T), cells, C, amp=40.0 + stream, p_dry=p_dry)
        if p_dry == 0.6:  # cells of other regimes beside it in the same tiles
            f[:, 3] = synth.fill(synth.PRECIP, 78, stream, np.arange(T), cells[:1], C, amp=30.0, p_dry=0.1)[:, 0]
            f[:, 11] = synth.fill(synth.PRECIP, 79, stream, np.arange(T), cells[:1], C, amp=30.0, p_dry=0.99)[:, 0]
        fields.append(f)
    X, y, Xp = fields
    if p_dry == 1.0:
        y[5::7, :] = 1.5  # (an all-zero climatology is the reference's ValueError: keep the observations positive somewhere)
    exp, est = bcsd_oracle.pointwise_fit_predict(1, X, y, Xp, gid, gid, return_anoms=True)
    out, st = ctx.bcsd_fit_predict(1, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid, True)
    assert np.array_equal(st, est)
    assert_close(out.to_host(), exp, what=f"p_dry={p_dry}")
    exp, est = bcsd_oracle.pointwise_fit_predict(1, X, y, Xp, gid, gid, return_anoms=False)
    out, st = ctx.bcsd_fit_predict(1, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid, False)
    assert np.array_equal(st, est)
    assert_close(out.to_host(), exp, what=f"p_dry={p_dry} abs")


QT_VARIANTS = [dict(n_endpoints=5), dict(n_endpoints=3, extrapolate="both"), dict(extrapolate="min"), dict(extrapolate="max"),
               dict(extrapolate=None), dict(extrapolate="1to1"), dict(alpha=0.3, beta=0.2), dict(n_endpoints=40)]


def test_qt_kwargs_golden(ctx):
    """qm_kwargs={'qt_kwargs': ...} (bcsd.py:59-67 -> quantile.py:418-431, 523-545) against g19_qt_kwargs.npz from the real
    reference: a predict series longer than the fit series reaches the tails of the fitted inverse CDFs, where `extrapolate`
    picks OLS line or end value per side and `n_endpoints` the points of the line; `alpha` / `beta` are without effect there
    and here.  Through the engine (state.set_tails) for all cells at once, and through the estimator surface for one cell."""
    from skdownscale_amd import BcsdPrecipitation, BcsdTemperature, synth

    g = load("g19_qt_kwargs")
    T, Tp, C = int(g["T"]), int(g["Tp"]), int(g["C"])
    index, index_p = synth.daily_calendar(T), synth.daily_calendar(Tp)
    cells = np.arange(C)
    gid, gid_p = month_gid(index), month_gid(index_p)
    tas = [synth.tas_field(n, int(g["seed"]), i, cells, int(g["c_full"])) for n, i in (("X_hist", index), ("y_obs", index), ("X_fut", index_p))]
    pr = [synth.pr_field(n, int(g["seed"]), t, cells, int(g["c_full"])) for n, t in (("X_hist", T), ("y_obs", T), ("X_fut", Tp))]
    assert len(QT_VARIANTS) == int(g["n_variants"])
    for kind, fields, n, cls in ((0, tas, len(QT_VARIANTS), BcsdTemperature), (1, pr, int(g["n_pr"]), BcsdPrecipitation)):
        X, y, Xp = fields
        for i, kw in enumerate(QT_VARIANTS[:n]):
            exp = g[("tas" if kind == 0 else "pr") + str(i)]
            st = ctx.bcsd_fit(kind, X, y, gid, 12, True)
            st.set_tails(kw.get("extrapolate", "both"), kw.get("n_endpoints", 10))
            out, status = ctx.bcsd_predict(st, Xp, gid_p)
            assert (status == 0).all()
            assert_close(out, exp, what=f"qt_kwargs {kw} kind={kind}")
            dout, _ = ctx.bcsd_predict(st, ctx.to_device(Xp), gid_p)
            assert np.array_equal(dout.to_host(), out)
            m = cls(qm_kwargs={"qt_kwargs": kw})
            m.fit(pd.DataFrame({"x": X[:, 1]}, index=index), pd.DataFrame({"y": y[:, 1]}, index=index))
            got = m.predict(pd.DataFrame({"x": Xp[:, 1]}, index=index_p)).values[:, 0]
            assert_close(got, exp[:, 1], what=f"estimator qt_kwargs {kw} kind={kind}")
    assert not np.allclose(g["tas0"], g["tas6"])  # (n_endpoints does reach the result; alpha / beta do not)


def test_masked_and_nan_cells(ctx):
    g = load("g7_masked")
    index, index_p, X, y, Xp = tas_inputs(g)
    X[0, 1] = np.nan
    X[0, 4] = np.nan
    y[0, 1] = np.nan
    out, st = run_engine(ctx, 0, X, y, Xp, month_gid(index), month_gid(index_p))
    assert np.array_equal(st, g["status"])
    assert_close(out, g["out_anoms"], what="masked")
    X[100, 2] = np.nan
    out, st = run_engine(ctx, 0, X, y, Xp, month_gid(index), month_gid(index_p))
    assert np.array_equal(st, g["status_nan_inside"])
    assert np.isnan(out[:, 2]).all()


@pytest.mark.parametrize("C,T,Tp", [(1, 365, 365), (7, 731, 1000), (9, 100, 90), (33, 1461, 1461), (5, 400, 370)])
@pytest.mark.parametrize("kind", [0, 1])
def test_vs_oracle_ragged_sizes(ctx, kind, C, T, Tp):
    """Edge sizes: single cell, cell counts not a multiple of the tile, tiny segments (n_g = 2..3)."""
    rng = np.random.default_rng(C * 1000 + T)
    index = pd.date_range("1999-01-01", periods=T, freq="D")  # predict months must exist in fit (bcsd.py:77)
    index_p = pd.date_range("2031-01-01", periods=Tp, freq="D")
    if kind == 0:
        X, y, Xp = (15 + 8 * rng.standard_normal((n, C)) for n in (T, T, Tp))
    else:
        X, y, Xp = (rng.gamma(0.7, 4.0, (n, C)) * (rng.random((n, C)) > 0.5) for n in (T, T, Tp))
        y = y + 0.01
    gid, gid_p = month_gid(index), month_gid(index_p)
    exp, est = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p)
    out, st = run_engine(ctx, kind, X, y, Xp, gid, gid_p)
    assert np.array_equal(st, est)
    assert_close(out, exp, what=f"ragged {kind} {C}x{T}->{Tp}")


@pytest.mark.parametrize("T,Tp,C", [(365, 365, 6), (3650, 3650, 9), (8000, 8000, 8), (14600, 14600, 8), (14600, 20000, 5),
                                    (14600, 3000, 3), (14600, 365, 4), (20000, 14600, 4), (24000, 24000, 3)])
@pytest.mark.parametrize("kind", [0, 1])
def test_every_register_sort_width_split_and_fused(ctx, kind, T, Tp, C):
    """Segment lengths from 31 to 2 046 samples exercise every K instantiation (5, 13, 21, 33) of the
    register/LDS merge-sort kernels, in the split (fit, predict) and the fused (fit_predict) entry points,
    with equal / longer / much shorter predict series (identity path, tail OLS path, table path)."""
    rng = np.random.default_rng(T + Tp + kind)
    index = pd.date_range("1980-01-01", periods=T, freq="D")
    index_p = pd.date_range("1980-01-01", periods=Tp, freq="D")
    if kind == 0:
        X, y, Xp = (15 + 8 * rng.standard_normal((n, C)) for n in (T, T, Tp))
    else:
        X, y, Xp = (rng.gamma(0.7, 4.0, (n, C)) * (rng.random((n, C)) > 0.5) for n in (T, T, Tp))
        y = y + 0.01
    gid, gid_p = month_gid(index), month_gid(index_p)
    exp, est = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p)
    out, st = run_engine(ctx, kind, X, y, Xp, gid, gid_p)
    assert np.array_equal(st, est)
    assert_close(out, exp, what=f"split {kind} {T}->{Tp}")
    fused, st = ctx.bcsd_fit_predict(kind, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid_p)
    assert np.array_equal(st, est)
    # same kernels; climatology sums are ordered by the register width K, which can differ between the fit
    # launch (sized for the fit segments) and the fused launch (sized for the longer of fit / predict)
    assert_close(fused.to_host(), out, rtol=1e-12, what=f"fused vs split {kind} {T}->{Tp}")
    assert_close(fused.to_host(), exp, what=f"fused {kind} {T}->{Tp}")


def test_fused_kernel_variants_agree_to_rounding(dev_ctx, monkeypatch):
    """BcsdTemperature takes the fused kernel (ranks read off position tags carried through the sort).  The
    development library can switch it off (SD_BCSD_FUSED=0: RANK + APPLY with the explicit rank search for every
    segment).  Same arithmetic up to the order of ONE sum: the fused kernels add the y_obs climatology in lane blocks of
    20 samples, RANK / APPLY / FIT in blocks of 21 (19, 13, 5), so the two paths -- and, inside one grid, the segments a fused
    kernel hands to the work list, and predict-from-a-state against fit_predict -- agree to the last bits of that mean
    (1e-13), not bit for bit.  Fused entry point and predict from a state, equal and unequal segment lengths (identity /
    table + tail paths), every kernel width."""
    ctx = dev_ctx
    rng = np.random.default_rng(11)
    for T, Tp, C in ((365, 365, 6), (3650, 3650, 9), (14600, 14600, 8), (14600, 20000, 5), (14600, 3000, 3), (9000, 8000, 11)):
        index = pd.date_range("1980-01-01", periods=T, freq="D")
        index_p = pd.date_range("1980-01-01", periods=Tp, freq="D")
        X, y, Xp = (15 + 8 * rng.standard_normal((n, C)) for n in (T, T, Tp))
        gid, gid_p = month_gid(index), month_gid(index_p)
        dX, dy, dXp = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xp)
        res = {}
        for name, env in (("fused", {}), ("search", {"SD_BCSD_FUSED": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            a, sa = ctx.bcsd_fit_predict(0, dX, dy, gid, 12, dXp, gid_p)
            st = ctx.bcsd_fit(0, dX, dy, gid, 12, True)
            b, sb = ctx.bcsd_predict(st, dXp, gid_p)
            res[name] = (a.to_host(), b.to_host())
            assert (sa == 0).all() and (sb == 0).all()
            for k in env:
                monkeypatch.delenv(k)
        # The fused kernel (sd_bcsd_fx.hip) sums the y_obs climatology over blocks of 20 samples per lane, RANK / APPLY
        # over blocks of 21 (19, 13, 5): the two paths agree to the last bits of that one mean, not bit for bit.
        for k in (0, 1):
            assert_close(res["fused"][k], res["search"][k], rtol=1e-13, what=f"fused vs search {T}->{Tp}")
        assert_close(res["fused"][0], res["fused"][1], rtol=1e-13, what=f"fit+predict vs predict from state {T}->{Tp}")
        exp, _ = bo.pointwise_fit_predict(0, X[:, :2], y[:, :2], Xp[:, :2], gid, gid_p)
        assert_close(res["fused"][0][:, :2], exp, what=f"fused {T}->{Tp}")


def test_fused_kernel_hands_tied_segments_back(ctx):
    """Exactly tied shifted samples (a constant stretch of x_fut: rolling mean == x, so u == x_climo for every
    sample of the stretch) cannot be ranked by the position tags: the workgroup appends its (tile, month) to the
    work list and RANK / APPLY (explicit search, max rank among ties: quantile.py:488) redo it.  Tied and clean cells,
    tied and clean months are mixed so that both paths write parts of the same output; plus -0.0 / +0.0."""
    rng = np.random.default_rng(23)
    T, C = 14600, 43
    index = pd.date_range("1980-01-01", periods=T, freq="D")
    gid = month_gid(index)
    X, y, Xp = (15 + 8 * rng.standard_normal((T, C)) for _ in range(3))
    for c in (0, 5, 6, 17, 40, 42):  # constant stretches inside one month / across a month boundary
        t0 = int(rng.integers(100, T - 200))
        Xp[t0:t0 + 24, c] = Xp[t0, c]
    Xp[3000:3040, 9] = np.round(Xp[3000:3040, 9])  # a few coarse values: some exact ties after the shift or none
    Xp[500, 11], Xp[501, 11] = 0.0, -0.0
    exp, est = bo.pointwise_fit_predict(0, X, y, Xp, gid, gid)
    out, st = ctx.bcsd_fit_predict(0, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid)
    assert np.array_equal(st, est)
    assert_close(out.to_host(), exp, what="mixed tied / clean segments")
    state = ctx.bcsd_fit(0, X, y, gid, 12, True)
    out2, _ = ctx.bcsd_predict(state, Xp, gid)
    assert_close(out2, exp, what="mixed tied / clean segments, from a state")


@pytest.mark.parametrize("Ct,c0", [(37, 3), (40, 4), (40, 7)])
def test_cell_shard_views_of_a_resident_grid(ctx, Ct, c0):
    """A cell range of a resident [T, C_total] field is passed as pointer + leading dimension (no copy): odd
    leading dimensions and odd cell offsets take the 8-byte load/store paths, even ones the 16-byte paths; the
    output is written into a view of a larger field without touching its other cells."""
    rng = np.random.default_rng(Ct + c0)
    T, Tp, C = 1461, 1461, 19
    index = pd.date_range("1990-01-01", periods=T, freq="D")
    gid = month_gid(index)
    full = {k: 15 + 8 * rng.standard_normal((T, Ct)) for k in ("X", "y", "Xp")}
    dev = {k: ctx.to_device(v) for k, v in full.items()}
    sl = slice(c0, c0 + C)
    for kind in (0, 1):
        host = {k: (np.abs(v[:, sl]) if kind else v[:, sl]).copy() for k, v in full.items()}
        if kind:
            dev = {k: ctx.to_device(np.abs(v)) for k, v in full.items()}
        ref, st_ref = ctx.bcsd_fit_predict(kind, ctx.to_device(host["X"]), ctx.to_device(host["y"]), gid, 12,
                                           ctx.to_device(host["Xp"]), gid)
        big = ctx.to_device(np.full((Tp, Ct), -777.0))
        out, st = ctx.bcsd_fit_predict(kind, dev["X"].cells(c0, c0 + C), dev["y"].cells(c0, c0 + C), gid, 12,
                                       dev["Xp"].cells(c0, c0 + C), gid, out=big.cells(c0, c0 + C))
        got = big.to_host()
        assert np.array_equal(st, st_ref)
        assert np.array_equal(got[:, sl], ref.to_host()), (kind, Ct, c0)
        assert (np.delete(got, np.s_[c0:c0 + C], axis=1) == -777.0).all()  # neighbours of the shard untouched
        state = ctx.bcsd_fit(kind, dev["X"].cells(c0, c0 + C), dev["y"].cells(c0, c0 + C), gid, 12, True)
        out2, _ = ctx.bcsd_predict(state, dev["Xp"].cells(c0, c0 + C), gid)
        exp, _ = bo.pointwise_fit_predict(kind, host["X"], host["y"], host["Xp"], gid, gid)
        assert_close(out2.to_host(), exp, what=f"view split {kind}")
        assert_close(got[:, sl], exp, what=f"view fused {kind}")


def test_generic_lds_kernels_still_agree(dev_ctx, monkeypatch):
    """SD_BCSD_PATH=v1 (development library) forces the generic LDS-bitonic kernels (fallback for segments > 2 112 samples)."""
    ctx = dev_ctx
    monkeypatch.setenv("SD_BCSD_PATH", "v1")
    g = load("g1_tas_small")
    index, index_p, X, y, Xp = tas_inputs(g)
    out, st = run_engine(ctx, 0, X, y, Xp, month_gid(index), month_gid(index_p))
    assert_close(out, g["out_anoms"], what="generic kernels")
    # a segment longer than the register-sort limit takes the generic path by itself
    rng = np.random.default_rng(3)
    T = 2300 * 2
    gid = (np.arange(T) % 2).astype(np.int32)
    X, y, Xp = (10 + rng.standard_normal((T, 3)) for _ in range(3))
    monkeypatch.delenv("SD_BCSD_PATH")
    exp, _ = bo.pointwise_fit_predict(0, X, y, Xp, gid, gid, G=2)
    st2 = ctx.bcsd_fit(0, X, y, gid, 2, True)
    out2, _ = ctx.bcsd_predict(st2, Xp, gid)
    assert_close(out2, exp, what="long segments")


def test_state_export_import_roundtrip(ctx):
    g = load("g1_tas_small")
    index, index_p, X, y, Xp = tas_inputs(g)
    st = ctx.bcsd_fit(0, X, y, month_gid(index), 12, True)
    e = st.export()
    # sorted state is what np.sort gives, per cell and group (bit-exact)
    order, off = bo.group_table(month_gid(index), 12)
    for c in range(X.shape[1]):
        for gg in range(12):
            seg = np.sort(y[order[off[gg]:off[gg + 1]], c])
            assert np.array_equal(e["y_sorted"][c, off[gg]:off[gg + 1]], seg)
    st2 = ctx.bcsd_import(e)
    out1, _ = ctx.bcsd_predict(st, Xp, month_gid(index_p))
    out2, _ = ctx.bcsd_predict(st2, Xp, month_gid(index_p))
    assert np.array_equal(out1, out2)


def test_fused_fit_predict_matches_two_step(ctx):
    g = load("g1_tas_small")
    index, index_p, X, y, Xp = tas_inputs(g)
    dX, dy, dXp = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xp)
    out, st = ctx.bcsd_fit_predict(0, dX, dy, month_gid(index), 12, dXp, month_gid(index_p))
    assert_close(out.to_host(), g["out_anoms"], what="fused")


def test_synth_device_matches_host(ctx):
    """The on-device generator is bit-identical to the NumPy mirror (so bench inputs are checkable)."""
    from skdownscale_amd import synth

    index = synth.daily_calendar(400)
    tab = synth.tas_tables(index)["y_obs"]
    d = ctx.empty((400, 37))
    ctx.synth_fill(d, synth.GAUSS, 5, tab["stream"], c_offset=11, c_full=1000, base=tab["base"], amp=tab["amp"], cell_scale=tab["cell_scale"])
    h = synth.fill(synth.GAUSS, 5, tab["stream"], np.arange(400), np.arange(11, 48), 1000, base=tab["base"], amp=tab["amp"], cell_scale=tab["cell_scale"])
    assert np.array_equal(d.to_host(), h)
    p = synth.PR_FIELDS["X_fut"]
    ctx.synth_fill(d, synth.PRECIP, 5, p["stream"], c_offset=0, c_full=64, amp=p["amp"], p_dry=p["p_dry"])
    assert np.array_equal(d.to_host(), synth.fill(synth.PRECIP, 5, p["stream"], np.arange(400), np.arange(37), 64, amp=p["amp"], p_dry=p["p_dry"]))
    ctx.synth_fill(d, synth.GAUSS, 5, 20, c_full=37, amp=2.0, stream2=21, amp2=1.0)
    assert np.array_equal(d.to_host(), synth.fill(synth.GAUSS, 5, 20, np.arange(400), np.arange(37), 37, amp=2.0, stream2=21, amp2=1.0))


# ---- estimator surface (reads like the reference's tests, test_pointwise_models.py:111-141, 221-233) ----

def test_estimators_reference_smoke_and_goldens():
    from skdownscale_amd import BcsdPrecipitation, BcsdTemperature

    g = load("g8_reference_tests")
    n = 365
    index = pd.date_range("2019-01-01", periods=n)
    X = pd.DataFrame({"foo": g["sine365"]}, index=index)
    y = X + 2
    model = BcsdTemperature()
    model.fit(X, y)
    y_hat = model.predict(X)
    assert len(y_hat) == len(X)
    assert isinstance(y_hat, pd.DataFrame) and y_hat.index.equals(index)
    assert_close(y_hat.values[:, 0], g["bcsd_tas_sine365"], what="sine365")
    assert list(model.y_climo_.index) == list(range(1, 13)) and model.n_features_in_ == 1
    assert set(model.quantile_mappers_) == set(range(1, 13))
    Xr = pd.DataFrame({"foo": g["pr_random365"]}, index=index)
    m = BcsdPrecipitation().fit(Xr, Xr + 2)
    assert_close(m.predict(Xr).values[:, 0], g["bcsd_pr_random365"], what="pr365")


def test_estimators_ndarray_input_and_errors():
    from sklearn.exceptions import NotFittedError

    from skdownscale_amd import BcsdPrecipitation, BcsdTemperature

    g = load("g4_ndarray")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = BcsdTemperature().fit(g["X"], g["y"])
        out = m.predict(g["Xp"])
    assert any("making one up" in str(x.message) for x in w)
    assert_close(out.values[:, 0], g["out_tas"], what="ndarray")
    assert out.index[0] == pd.Timestamp("1950-01-31")  # note N7
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert_close(BcsdPrecipitation().fit(g["P"], g["yP"]).predict(g["PP"]).values[:, 0], g["out_pr"], what="ndarray pr")
        with pytest.raises(NotFittedError):
            BcsdTemperature().predict(g["Xp"])
        with pytest.raises(ValueError, match="BCSD only supports up to 4 features, found 2"):
            BcsdTemperature().fit(np.zeros((50, 2)), np.zeros(50))
        with pytest.raises(ValueError, match="BCSD only supports 1 feature, found 2"):
            BcsdPrecipitation().fit(np.ones((50, 2)), np.ones(50))
        with pytest.raises(ValueError, match="Input X contains NaN"):
            X = g["X"].copy()
            X[3] = np.nan
            BcsdTemperature().fit(X, g["y"])
        with pytest.raises(ValueError, match="Invalid value in target climatology"):
            BcsdPrecipitation().fit(g["P"], np.zeros_like(g["yP"]))
    # 100 days leave most day-of-year groups empty: the reference's QuantileMapper.fit refuses an empty group (bcsd.py:66-67)
    with pytest.raises(ValueError, match="Found array with 0 sample"):
        BcsdTemperature(time_grouper="daily_nasa-nex").fit(pd.DataFrame(g["X"], index=pd.date_range("2000", periods=100)),
                                                         pd.DataFrame(g["y"], index=pd.date_range("2000", periods=100)))
    with pytest.raises(KeyError):  # a pandas frequency string ends in df.groupby('M') in the reference (bcsd.py:39-41, 49)
        BcsdTemperature(time_grouper="M").fit(pd.DataFrame(g["X"], index=pd.date_range("2000", periods=100)),
                                              pd.DataFrame(g["y"], index=pd.date_range("2000", periods=100)))


def test_BcsdTemperature_nasanex():
    """The reference's test (test_pointwise_models.py:315-320) restated: fit with time_grouper='daily_nasa-nex' swaps the
    grouper class in; plus what the reference's fitted object exposes."""
    from skdownscale_amd import BcsdTemperature, PaddedDOYGrouper

    rng = np.random.default_rng(0)
    index = pd.date_range(start="1980-01-01", end="1982-12-31")
    X = pd.DataFrame({"foo": rng.random(len(index))}, index=index)
    y = pd.DataFrame({"foo": rng.random(len(index))}, index=index)
    model_nasanex = BcsdTemperature(time_grouper="daily_nasa-nex", return_anoms=False).fit(X, y)
    assert issubclass(model_nasanex.time_grouper, PaddedDOYGrouper)
    assert model_nasanex.timestep == "daily"
    assert model_nasanex.y_climo_.shape == (366, 1) and list(model_nasanex.y_climo_.index[:3]) == [1, 2, 3]
    assert len(model_nasanex.quantile_mappers_) == 366
    assert model_nasanex.quantile_mappers_[366].x_cdf_fit_.cdf_.vals.shape == (89,)
    assert model_nasanex.predict(X).shape == (len(index), 1)


def test_bcsd_daily_nasanex_golden(ctx):
    """g13_nasanex.npz (from the reference): 366 padded day-of-year groups in fit (bcsd.py:36-38,50-55), predict by
    day-of-month keys with the 9-sample rolling mean over months (bcsd.py:247-267), the exceptions of return_anoms=True,
    the monthly model with a day-of-month climate-trend grouper -- estimators, engine and PointWiseDownscaler."""
    from _cases import nasanex_inputs
    from skdownscale_amd import BcsdPrecipitation, BcsdTemperature, PointWiseDownscaler
    from skdownscale_amd.core import GridArray
    from skdownscale_amd.groupers import DAY_GROUPER, padded_doy_table

    g = load("g13_nasanex")
    for case in (0, 1):
        index, index_p, (X, y, Xp), (P, yP, Pp) = nasanex_inputs(g, case)
        C = X.shape[1]
        order, offsets = padded_doy_table(index)
        gq, gt = np.asarray(index_p.day, dtype=np.int32) - 1, np.asarray(index_p.month, dtype=np.int32) - 1
        # engine, all cells at once
        st = ctx.bcsd_fit_groups(0, X, y, order, offsets, return_anoms=False)
        e = st.export()
        assert e["info"]["T"] == len(order) and np.array_equal(e["group_offsets"], offsets)
        np.testing.assert_allclose(e["y_climo"].T, g[f"y_climo{case}"], rtol=1e-12)
        np.testing.assert_allclose(e["x_climo"].T, g[f"x_climo{case}"], rtol=1e-12)
        for k in (1, 59, 60, 200, 366):
            assert np.array_equal(e["y_sorted"][:, offsets[k - 1]:offsets[k]].T, g[f"cdf{case}_{k}"])
        out, status = ctx.bcsd_predict_trend(st, Xp, gq, gt, 12)
        assert (status == 0).all()
        assert_close(out, g[f"tas_out{case}"], what=f"daily tas (engine) case {case}")
        dout, _ = ctx.bcsd_predict_trend(ctx.bcsd_fit_groups(0, ctx.to_device(X), ctx.to_device(y), order, offsets, False),
                                         ctx.to_device(Xp), gq, gt, 12)
        assert np.array_equal(dout.to_host(), out)
        stp = ctx.bcsd_fit_groups(1, P, yP, order, offsets, return_anoms=False)
        np.testing.assert_allclose(stp.export()["y_climo"].T, g[f"pr_y_climo{case}"], rtol=1e-12)
        pout, _ = ctx.bcsd_predict(stp, Pp, gq)
        assert_close(pout, g[f"pr_out{case}"], what=f"daily pr (engine) case {case}")
        # estimators, one cell
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = BcsdTemperature(time_grouper="daily_nasa-nex", return_anoms=False).fit(pd.DataFrame(X[:, :1], index=index),
                                                                                       pd.DataFrame(y[:, :1], index=index))
            assert_close(m.predict(pd.DataFrame(Xp[:, :1], index=index_p)).values[:, 0], g[f"tas_out{case}"][:, 0], what="estimator")
            np.testing.assert_allclose(m.y_climo_.values[:, 0], g[f"y_climo{case}"][:, 0], rtol=1e-12)
            assert np.array_equal(m.quantile_mappers_[60].x_cdf_fit_.cdf_.vals, g[f"cdf{case}_60"][:, 0])
            m = BcsdTemperature(time_grouper="daily_nasa-nex").fit(pd.DataFrame(X[:, :1], index=index), pd.DataFrame(y[:, :1], index=index))
            with pytest.raises(ValueError) as ei:
                m.predict(pd.DataFrame(Xp[:, :1], index=index_p))
            assert str(ei.value) == str(g[f"tas_anoms_error{case}"])
            mp = BcsdPrecipitation(time_grouper="daily_nasa-nex", return_anoms=False).fit(pd.DataFrame(P[:, :1], index=index),
                                                                                          pd.DataFrame(yP[:, :1], index=index))
            assert_close(mp.predict(pd.DataFrame(Pp[:, :1], index=index_p)).values[:, 0], g[f"pr_out{case}"][:, 0], what="pr estimator")
            mp = BcsdPrecipitation(time_grouper="daily_nasa-nex").fit(pd.DataFrame(P[:, :1], index=index), pd.DataFrame(yP[:, :1], index=index))
            with pytest.raises(ValueError) as ei:
                mp.predict(pd.DataFrame(Pp[:, :1], index=index_p))
            assert str(ei.value) == str(g[f"pr_anoms_error{case}"])
            mt = BcsdTemperature(climate_trend=DAY_GROUPER).fit(pd.DataFrame(X[:, :1], index=index), pd.DataFrame(y[:, :1], index=index))
            assert_close(mt.predict(pd.DataFrame(Xp[:, :1], index=index_p)).values[:, 0], g[f"tas_daytrend_out{case}"][:, 0],
                         what="day-of-month climate trend")
            # the grid driver batches the same configurations
            pw = PointWiseDownscaler(BcsdTemperature(time_grouper="daily_nasa-nex", return_anoms=False))
            pw.fit(GridArray(X, ("time", "point"), {"time": index}), GridArray(y, ("time", "point"), {"time": index}))
            got = pw.predict(GridArray(Xp, ("time", "point"), {"time": index_p}))
            assert_close(got.values, g[f"tas_out{case}"], what="PointWiseDownscaler daily")
            assert isinstance(pw._model.time_grouper, str)  # the prototype is not modified (each cell's copy is, in the reference)
            pw = PointWiseDownscaler(BcsdTemperature(climate_trend=DAY_GROUPER))
            pw.fit(GridArray(X, ("time", "point"), {"time": index}), GridArray(y, ("time", "point"), {"time": index}))
            got = pw.predict(GridArray(Xp, ("time", "point"), {"time": index_p}))
            assert_close(got.values, g[f"tas_daytrend_out{case}"], what="PointWiseDownscaler day-of-month trend")


def test_estimator_pickle_roundtrip():
    import pickle

    from skdownscale_amd import BcsdTemperature

    g = load("g8_reference_tests")
    index = pd.date_range("2019-01-01", periods=365)
    X = pd.DataFrame({"foo": g["sine365"]}, index=index)
    m = BcsdTemperature().fit(X, X + 2)
    m2 = pickle.loads(pickle.dumps(m))
    assert np.array_equal(m2.predict(X).values, m.predict(X).values)


def test_pointwise_downscaler_bcsd_grid():
    from skdownscale_amd import BcsdTemperature, GridArray, PointWiseDownscaler

    g = load("g7_masked")
    index, index_p, X, y, Xp = tas_inputs(g)
    X[0, 1] = np.nan
    X[0, 4] = np.nan
    y[0, 1] = np.nan
    T = X.shape[0]
    mk = lambda a, idx: GridArray(a.reshape(T, 2, 3), ("time", "y", "x"), {"time": idx, "y": np.arange(2), "x": np.arange(3)})  # noqa: E731
    pw = PointWiseDownscaler(BcsdTemperature())
    pw.fit(mk(X, index), mk(y, index))
    out = pw.predict(mk(Xp, index_p))
    assert out.dims == ("time", "y", "x") and out.sizes == {"time": T, "y": 2, "x": 3}
    assert_close(out.values.reshape(T, 6), g["out_anoms"], what="pointwise")
    yc = pw.get_attr("y_climo_")
    assert yc.dims == ("group", "y", "x") and np.isnan(yc.values[:, 0, 1]).all()
    X[100, 2] = np.nan
    with pytest.raises(ValueError, match="Input X contains NaN"):
        PointWiseDownscaler(BcsdTemperature()).fit(mk(X, index), mk(y, index))
    with pytest.raises(TypeError):
        PointWiseDownscaler(object())


def test_float32_inputs_follow_the_reference():
    """float32 fields: the reference's np.sort / np.interp promote to float64 and its estimators return
    float64 (checked against the reference: within 2e-7 of the float64 pipeline on the upcast inputs), while
    PointWiseDownscaler allocates its result in X's dtype (core.py:119).  The engine upcasts once on the way in."""
    from skdownscale_amd import BcsdPrecipitation, BcsdTemperature, GridArray, PointWiseDownscaler

    rng = np.random.default_rng(4)
    T, C = 1461, 6
    index = pd.date_range("1990-01-01", periods=T)
    X32 = (12 + 7 * rng.standard_normal((T, C))).astype(np.float32)
    y32 = (10 + 5 * rng.standard_normal((T, C))).astype(np.float32)
    gid = month_gid(index)
    exp, _ = bo.pointwise_fit_predict(bo.TAS, X32.astype(np.float64), y32.astype(np.float64), X32.astype(np.float64), gid, gid)
    m = BcsdTemperature().fit(pd.DataFrame(X32[:, :1], index=index), pd.DataFrame(y32[:, :1], index=index))
    out = m.predict(pd.DataFrame(X32[:, :1], index=index))
    assert out.values.dtype == np.float64
    assert_close(out.values[:, 0], exp[:, 0], what="float32 DataFrame through the estimator")
    mk = lambda a: GridArray(a.reshape(T, 2, 3), ("time", "y", "x"), {"time": index})  # noqa: E731
    pw = PointWiseDownscaler(BcsdTemperature())
    pw.fit(mk(X32), mk(y32))
    res = pw.predict(mk(X32))
    assert res.values.dtype == np.float32
    np.testing.assert_allclose(res.values.reshape(T, C), exp.astype(np.float32), rtol=2e-6, atol=2e-6)
    P32 = (np.abs(X32) + 0.1).astype(np.float32)
    pexp, _ = bo.pointwise_fit_predict(bo.PR, P32.astype(np.float64), P32.astype(np.float64) + 0.5, P32.astype(np.float64), gid, gid)
    mp = BcsdPrecipitation().fit(pd.DataFrame(P32[:, :1], index=index), pd.DataFrame(P32[:, :1] + np.float32(0.5), index=index))
    assert_close(mp.predict(pd.DataFrame(P32[:, :1], index=index)).values[:, 0], pexp[:, 0], what="float32 precipitation")


def test_pointwise_downscaler_xarray_inputs_on_the_engine():
    """xarray DataArrays in and out of the batched BCSD path (core.py:225-336), whole and spatially chunked; get_attr with a
    template.  Uses the real xarray where installed, else the stand-in of tests/xarray_stub (see its docstring)."""
    xr = pytest.importorskip("xarray")
    from skdownscale_amd import BcsdTemperature, GridArray, PointWiseDownscaler

    g = load("g7_masked")
    index, index_p, X, y, Xp = tas_inputs(g)
    X[0, 1] = X[0, 4] = y[0, 1] = np.nan  # the masked cells of the golden case (core.py:35-37)
    T = X.shape[0]
    mk = lambda a, idx: xr.DataArray(a.reshape(T, 2, 3), dims=("time", "y", "x"), coords={"time": idx, "y": np.arange(2), "x": np.arange(3)})  # noqa: E731
    mg = lambda a, idx: GridArray(a.reshape(T, 2, 3), ("time", "y", "x"), {"time": idx, "y": np.arange(2), "x": np.arange(3)})  # noqa: E731
    ref = PointWiseDownscaler(BcsdTemperature())
    ref.fit(mg(X, index), mg(y, index))
    expected = ref.predict(mg(Xp, index_p)).values
    pw = PointWiseDownscaler(BcsdTemperature())
    pw.fit(mk(X, index), mk(y, index))
    out = pw.predict(mk(Xp, index_p))
    assert isinstance(out, xr.DataArray) and tuple(out.dims) == ("time", "y", "x")
    assert np.array_equal(out.values, expected, equal_nan=True)
    assert_close(out.values.reshape(T, 6), g["out_anoms"], what="xarray in / out")
    chunked = PointWiseDownscaler(BcsdTemperature())
    chunked.fit(mk(X, index).chunk({"y": 1, "x": 2}), mk(y, index).chunk({"y": 1, "x": 2}))
    outc = chunked.predict(mk(Xp, index_p).chunk({"y": 1, "x": 2}))
    assert isinstance(outc, xr.DataArray) and np.array_equal(outc.values, expected, equal_nan=True)
    template = xr.DataArray(np.zeros((12, 2, 3)), dims=("month", "y", "x"), coords={"month": np.arange(1, 13)})
    yc = pw.get_attr("y_climo_", template_output=template)
    assert isinstance(yc, xr.DataArray) and dict(yc.sizes) == {"month": 12, "y": 2, "x": 3}
    np.testing.assert_allclose(yc.values, ref.get_attr("y_climo_").values, rtol=0, atol=0, equal_nan=True)


def test_per_group_kernel_width(dev_ctx, monkeypatch):
    """A 40-year daily series has 31-day months (1 240 samples: 21 per lane) and shorter ones (<= 1 216: 19 per lane):
    the shorter months get their own launch of the narrower kernels sharing the hand-off slabs.  The result must not
    depend on the split beyond the summation order of the monthly means (SD_RS_SPLIT=0 = one launch of the widest
    kernels; sorted states are bit-identical), for the fused entry point, fit -> predict from a state (shifted predict
    calendar: some months cross the width limit only there), and both kinds.  (The switch exists in the development library.)"""
    ctx = dev_ctx
    rng = np.random.default_rng(5)
    for kind, T, Tp, C in ((0, 14600, 14600, 19), (1, 14600, 14600, 9), (0, 14000, 14900, 11)):
        index = pd.date_range("1980-01-01", periods=T, freq="D")
        index_p = pd.date_range("1979-01-01", periods=Tp, freq="D")
        X, y, Xp = (15 + 8 * rng.standard_normal((n, C)) for n in (T, T, Tp))
        if kind == 1:
            X, y, Xp = np.abs(X) * (rng.random(X.shape) > 0.3), np.abs(y) + 0.1, np.abs(Xp) * (rng.random(Xp.shape) > 0.3)
        gid, gid_p = month_gid(index), month_gid(index_p)
        dX, dy, dXp = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xp)
        res = {}
        for split in ("1", "0"):
            monkeypatch.setenv("SD_RS_SPLIT", split)
            a, sa = ctx.bcsd_fit_predict(kind, dX, dy, gid, 12, dXp, gid_p)
            st = ctx.bcsd_fit(kind, dX, dy, gid, 12, True)
            b, sb = ctx.bcsd_predict(st, dXp, gid_p)
            res[split] = (a.to_host(), b.to_host(), st.export()["y_sorted"])
            assert (sa == 0).all() and (sb == 0).all()
        monkeypatch.delenv("SD_RS_SPLIT")
        assert np.array_equal(res["1"][2], res["0"][2]), (kind, T, Tp)
        for u, v in zip(res["1"][:2], res["0"][:2]):
            assert_close(u, v, rtol=1e-13, what=f"width split vs single launch kind={kind}")
        exp, _ = bo.pointwise_fit_predict(kind, X[:, :2], y[:, :2], Xp[:, :2], gid, gid_p, G=12, return_anoms=True)
        assert_close(res["1"][0][:, :2], exp, what=f"width split kind={kind}")


def test_full_size_grid_is_consistent(ctx):
    """BASELINE config 2 size (100 000 cells x 14 600 steps, 12 months) through a size-independent property: the grid
    is made of identical 8 192-cell blocks, so every block -- whatever tile, workgroup and XCD it lands on -- must
    reproduce the first one bit for bit; the first cells of block 0 are checked against the C oracle."""
    import c_oracle
    from skdownscale_amd import synth

    if not c_oracle.available():
        pytest.skip("C oracle not built")
    T, C, B = 14600, 100_000, 8192
    index = synth.daily_calendar(T)
    gid = month_gid(index)
    tabs = synth.tas_tables(index)
    fields = {}
    for name in ("X_hist", "y_obs", "X_fut"):
        d = ctx.empty((T, C))
        for c0 in range(0, C, B):
            c1 = min(C, c0 + B)
            ctx.synth_fill(d.cells(c0, c1), synth.GAUSS, 3, tabs[name]["stream"], c_offset=0, c_full=B, base=tabs[name]["base"],
                           amp=tabs[name]["amp"], cell_scale=tabs[name]["cell_scale"])
        fields[name] = d
    out, status = ctx.bcsd_fit_predict(0, fields["X_hist"], fields["y_obs"], gid, 12, fields["X_fut"], gid)
    assert (status == 0).all()
    rows = np.unique(np.linspace(0, T - 1, 48).astype(np.int64))
    got = np.empty((len(rows), C))
    for i, t in enumerate(rows):  # one 800 KB row at a time
        got[i] = ctx.wrap(out.ptr + int(t) * C * 8, (1, C)).to_host()[0]
    for c0 in range(B, C, B):
        c1 = min(C, c0 + B)
        assert np.array_equal(got[:, c0:c1], got[:, :c1 - c0]), f"block at cell {c0} differs from block 0"
    n = 16
    cells = np.arange(n)
    exp, _ = c_oracle.bcsd_fit_predict(0, synth.tas_field("X_hist", 3, index, cells, B), synth.tas_field("y_obs", 3, index, cells, B),
                                       synth.tas_field("X_fut", 3, index, cells, B), gid, gid)
    assert_close(got[:, :n], exp[rows], scale=float(np.std(exp)), what="full-size block 0 vs C oracle")
    for d in fields.values():
        d.free()
    out.free()


@pytest.mark.parametrize("kind", [0, 1])
def test_host_predict_in_cell_blocks_matches_resident(ctx, kind):
    """Host-buffer predict of a grid large enough (>= 256 MB per field) to go through sd_bcsd_predict's pipeline: four blocks
    of cells, each uploaded (2-D staged copy), predicted on its slice of the fitted state and drained by a second host thread
    while the next block comes in.  Must equal the resident path bit for bit, ragged last block, masked and non-finite
    cells (per-block status slices) included."""
    rng = np.random.default_rng(5 + kind)
    T, C = 14600, 2308  # 4 blocks of 584, 584, 584, 556 cells
    index = pd.date_range("1980-01-01", periods=T, freq="D")
    gid = month_gid(index)
    if kind == 0:
        X, y, Xp = (15 + 8 * rng.standard_normal((T, C)) for _ in range(3))
    else:
        X, y, Xp = (rng.gamma(0.7, 4.0, (T, C)) * (rng.random((T, C)) > 0.5) for _ in range(3))
        y = y + 0.01
    X[0, 700] = np.nan       # masked cell (block 1)
    Xp[1234, 2300] = np.inf  # non-finite predict sample (last block)
    st = ctx.bcsd_fit(kind, X, y, gid, 12, True)
    out_h, st_h = ctx.bcsd_predict(st, Xp, gid)                 # host buffers: the pipeline
    out_d, st_d = ctx.bcsd_predict(st, ctx.to_device(Xp), gid)  # resident
    assert np.array_equal(st_h, st_d) and st_h[700] != 0 and st_h[2300] != 0 and (np.delete(st_h, [700, 2300]) == 0).all()
    got, ref = out_h, out_d.to_host()
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(got[:, 700]).all() and np.isnan(got[:, 2300]).all()
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok], ref[ok])
    exp, _ = bo.pointwise_fit_predict(kind, X[:, :3], y[:, :3], Xp[:, :3], gid, gid)
    assert_close(got[:, :3], exp, what=f"host pipeline kind={kind}")


def test_full_size_precipitation_grid_is_consistent(ctx):
    """BASELINE config 3 size (BcsdPrecipitation, 250 000 cells x 14 600 steps, zero-inflated) through the same
    size-independent property: identical 8 192-cell blocks must reproduce block 0 bit for bit wherever they land; the
    first cells of block 0 are checked against the C oracle (bcsd.py:115-185: ratio anomalies, ties among the dry days)."""
    import c_oracle
    from skdownscale_amd import synth

    if not c_oracle.available():
        pytest.skip("C oracle not built")
    T, C, B = 14600, 250_000, 8192
    gid = month_gid(synth.daily_calendar(T))
    fields = {}
    for name in ("X_hist", "y_obs", "X_fut"):
        p = synth.PR_FIELDS[name]
        d = ctx.empty((T, C))
        for c0 in range(0, C, B):
            ctx.synth_fill(d.cells(c0, min(C, c0 + B)), synth.PRECIP, 7, p["stream"], c_offset=0, c_full=B, amp=p["amp"], p_dry=p["p_dry"])
        fields[name] = d
    out, status = ctx.bcsd_fit_predict(1, fields["X_hist"], fields["y_obs"], gid, 12, fields["X_fut"], gid)
    assert (status == 0).all()
    rows = np.unique(np.linspace(0, T - 1, 48).astype(np.int64))
    got = np.empty((len(rows), C))
    for i, t in enumerate(rows):
        got[i] = ctx.wrap(out.ptr + int(t) * C * 8, (1, C)).to_host()[0]
    for c0 in range(B, C, B):
        c1 = min(C, c0 + B)
        assert np.array_equal(got[:, c0:c1], got[:, :c1 - c0]), f"block at cell {c0} differs from block 0"
    n = 16
    cells = np.arange(n)
    Xh, yh, Xf = (synth.pr_field(name, 7, T, cells, B) for name in ("X_hist", "y_obs", "X_fut"))
    assert 0.3 < float((Xf == 0).mean()) < 0.7, "the generator is expected to leave about half the days dry"
    exp, _ = c_oracle.bcsd_fit_predict(1, Xh, yh, Xf, gid, gid)
    assert_close(got[:, :n], exp[rows], scale=float(np.std(exp)), what="full-size precipitation block 0 vs C oracle")
    for d in fields.values():
        d.free()
    out.free()


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("G,T,Tp,C", [(1, 14600, 14600, 3), (1, 2113, 2500, 2), (4, 14600, 9000, 3), (1, 19456, 19456, 1),
                                     (2, 6000, 30000, 2), (1, 14600, 1000, 2)])
def test_long_segments_workgroup_sort(ctx, kind, G, T, Tp, C):
    """Groups of 2 113 ... 19 456 samples (e.g. a whole 40-year daily series as one group) run on the workgroup merge
    sort: equal / longer / shorter predict series, tie-heavy dyadic data for the precipitation kind."""
    rng = np.random.default_rng(G * T + Tp + kind)
    X, y, Xp = (12 + 5 * rng.standard_normal((n, C)) for n in (T, T, Tp))
    if kind == 1:
        X, y, Xp = (np.round(np.abs(a) * 4) / 4 * (rng.random(a.shape) > 0.45) for a in (X, y, Xp))
        y = y + 0.25
    gid = (np.arange(T) * G // T).astype(np.int32)
    gid_p = (np.arange(Tp) % G).astype(np.int32)
    exp, est = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=G)
    st = ctx.bcsd_fit(kind, X, y, gid, G, True)
    out, status = ctx.bcsd_predict(st, Xp, gid_p)
    assert np.array_equal(status, est)
    assert_close(out, exp, what=f"long segments kind={kind} G={G} {T}->{Tp}")
    fused, _ = ctx.bcsd_fit_predict(kind, ctx.to_device(X), ctx.to_device(y), gid, G, ctx.to_device(Xp), gid_p)
    assert np.array_equal(fused.to_host(), out)


def test_segments_beyond_every_kernel_are_refused(ctx):
    rng = np.random.default_rng(0)
    T = 20000  # one group of 20 000 samples: beyond the workgroup sort and the generic predict kernel
    X, y = rng.standard_normal((T, 1)), rng.standard_normal((T, 1))
    gid = np.zeros(T, dtype=np.int32)
    st = ctx.bcsd_fit(0, X, y, gid, 1, True)  # (the fit alone still fits the generic kernel)
    with pytest.raises(NotImplementedError, match="does not fit"):
        ctx.bcsd_predict(st, X, gid)


def test_chunked_grids_and_attributes_on_the_engine():
    """Engine-batched estimators behind the grid driver: a chunked input is fitted / predicted block by block (every block
    one batched launch; core.py:256-262, 300-336) with the same result as the whole grid at once; get_attr rebuilds the
    per-cell fitted attributes from the exported state (core.py:405-425)."""
    from skdownscale_amd import AnalogRegression, BcsdTemperature, PointWiseDownscaler, PureRegression
    from skdownscale_amd.core import GridArray, _BlockedModels

    rng = np.random.default_rng(8)
    T, shape = 1461, (3, 5)
    index = pd.date_range("1980-01-01", periods=T)
    X = GridArray(15 + 8 * rng.standard_normal((T,) + shape), ("time", "y", "x"), {"time": index})
    y = GridArray(13 + 9 * rng.standard_normal((T,) + shape), ("time", "y", "x"), {"time": index})
    X.values[0, 1, 2] = np.nan  # a masked cell
    whole = PointWiseDownscaler(BcsdTemperature())
    whole.fit(X, y)
    expected = whole.predict(X)
    blocked = PointWiseDownscaler(BcsdTemperature())
    blocked.fit(X.chunk({"y": 2, "x": 2}), y.chunk({"y": 2, "x": 2}))
    assert isinstance(blocked._models, _BlockedModels) and len(blocked._models.blocks) == 6
    got = blocked.predict(X.chunk({"y": 2, "x": 2}))
    # the result of a chunked grid is lazy like the reference's map_blocks result: blocks on demand, one at a time ...
    from skdownscale_amd.core import LazyGridArray

    assert isinstance(got, LazyGridArray) and not got.computed and got.shape == expected.shape and got.dims == expected.dims
    assert got.chunksizes["y"] == (2, 1) and got.chunksizes["x"] == (2, 2, 1)
    seen = 0
    for sel, block in got.iter_blocks():
        ref_block = expected.values[tuple(sel.get(d, slice(None)) for d in expected.dims)]
        assert block.shape == ref_block.shape and np.array_equal(np.isnan(block.values), np.isnan(ref_block))
        np.testing.assert_allclose(block.values[~np.isnan(ref_block)], ref_block[~np.isnan(ref_block)], rtol=1e-12)
        seen += 1
    assert seen == 6 and not got.computed  # ... and nothing assembled until .values is asked for
    assert np.array_equal(np.isnan(got.values), np.isnan(expected.values)) and got.computed
    np.testing.assert_allclose(got.values[~np.isnan(got.values)], expected.values[~np.isnan(expected.values)], rtol=1e-12)
    # ... after which the blocks are views of the assembled field: iterating again computes nothing
    got._thunks = [(sel, lambda: (_ for _ in ()).throw(AssertionError("block computed twice"))) for sel, _ in got._thunks]
    for sel, block in got.iter_blocks():
        assert np.shares_memory(block.values, got.values)
    # attributes: scalars on the model grid, array-valued ones through a template (group axis first)
    n = whole.get_attr("n_features_in_", "int64")
    assert n.dims == ("y", "x") and n.values[0, 0] == 1
    template = GridArray(np.zeros((12, 1) + shape), ("group", "col", "y", "x"))
    yc = whole.get_attr("y_climo_", "float64", template_output=template)
    ycb = blocked.get_attr("y_climo_", "float64", template_output=template)
    assert yc.shape == (12, 1) + shape and np.isnan(yc.values[:, :, 1, 2]).all()
    np.testing.assert_allclose(yc.values[:, 0], whole.get_attr("y_climo_").values, rtol=0, atol=0)
    ok = ~np.isnan(yc.values)
    np.testing.assert_allclose(ycb.values[ok], yc.values[ok], rtol=1e-12)
    # regressions
    Xr = GridArray(rng.standard_normal((300, 2) + shape), ("time", "variable", "y", "x"))
    yr = GridArray(rng.standard_normal((300,) + shape), ("time", "y", "x"))
    pr = PointWiseDownscaler(PureRegression())
    pr.fit(Xr, yr)
    err = pr.get_attr("fit_error_", "float64")
    assert err.shape == shape and (err.values > 0).all()
    ar = PointWiseDownscaler(AnalogRegression(n_analogs=20))
    ar.fit(Xr, yr)
    assert (ar.get_attr("k_", "int64").values == 20).all()


def test_sharded_pointwise_downscaler_single_rank_matches_the_plain_one():
    """ShardedPointWiseDownscaler (skdownscale_amd/shard.py) with one rank on the GPU: same fields as PointWiseDownscaler
    for the batched BcsdTemperature (the N > 1 layout logic runs on CPU in tests/test_host.py)."""
    from skdownscale_amd import BcsdTemperature, PointWiseDownscaler
    from skdownscale_amd.core import GridArray
    from skdownscale_amd.shard import ShardedPointWiseDownscaler

    rng = np.random.default_rng(3)
    T, shape = 1461, (4, 6)
    index = pd.date_range("1984-01-01", periods=T)
    mk = lambda a: GridArray(a, ("time", "y", "x"), {"time": index})  # noqa: E731
    X, y = mk(15 + 8 * rng.standard_normal((T,) + shape)), mk(13 + 9 * rng.standard_normal((T,) + shape))
    X.values[0, 2, 3] = np.nan
    plain = PointWiseDownscaler(BcsdTemperature())
    plain.fit(X, y)
    exp = plain.predict(X)
    sh = ShardedPointWiseDownscaler(BcsdTemperature()).fit(X, y)
    got = sh.predict(X)
    assert got.dims == exp.dims and got.shape == exp.shape
    assert np.array_equal(np.isnan(got.values), np.isnan(exp.values))
    ok = ~np.isnan(exp.values)
    assert np.array_equal(got.values[ok], exp.values[ok])


def test_float32_fields_cross_pcie_as_float32():
    """float32 host grids take the float32 transport (widened on the device: exact; a float32 result is narrowed there: the
    rounding of ``.astype(np.float32)``): bit-identical to upcasting on the host, for the engine call and for
    PointWiseDownscaler, and the conversion entry points themselves round-trip."""
    from skdownscale_amd import BcsdTemperature, GridArray, PointWiseDownscaler
    from skdownscale_amd.engine import default_context

    ctx = default_context()
    rng = np.random.default_rng(11)
    T, C = 1461, 8
    index = pd.date_range("1990-01-01", periods=T)
    gid = month_gid(index)
    X32 = (12 + 7 * rng.standard_normal((T, C))).astype(np.float32)
    y32 = (10 + 5 * rng.standard_normal((T, C))).astype(np.float32)
    d = ctx.widen_to_device(X32)
    assert d.dtype == np.float64 and np.array_equal(d.to_host(), X32.astype(np.float64))
    assert np.array_equal(ctx.narrow_to_host(ctx.to_device(X32.astype(np.float64) * 1.000000123)), (X32.astype(np.float64) * 1.000000123).astype(np.float32))
    st64 = ctx.bcsd_fit(0, X32.astype(np.float64), y32.astype(np.float64), gid, 12, True)
    exp, _ = ctx.bcsd_predict(st64, X32.astype(np.float64), gid)
    st32 = ctx.bcsd_fit(0, X32, y32, gid, 12, True)
    out, status = ctx.bcsd_predict(st32, X32, gid)
    assert out.dtype == np.float64 and (status == 0).all() and np.array_equal(out, exp)
    out32, _ = ctx.bcsd_predict(st32, X32, gid, out_dtype=np.float32)
    assert out32.dtype == np.float32 and np.array_equal(out32, exp.astype(np.float32))
    mk = lambda a: GridArray(a.reshape(T, 2, 4), ("time", "y", "x"), {"time": index})  # noqa: E731
    pw = PointWiseDownscaler(BcsdTemperature())
    pw.fit(mk(X32), mk(y32))
    res = pw.predict(mk(X32))
    assert res.values.dtype == np.float32 and np.array_equal(res.values.reshape(T, C), exp.astype(np.float32))


@pytest.mark.parametrize("masked", [False, True])
def test_config1_grid_through_pointwise_downscaler(masked):
    """BASELINE configs[0]: BcsdTemperature on the 16 x 16 synthetic grid, 40-year daily series (14 600 steps), through the drop-in
    surface PointWiseDownscaler.fit / .predict (core.py:225-338) -- all 256 cells against the C oracle's per-cell loop
    (core.py:86-96, 137-141 around bcsd.py:197-269); the masked variant blanks cells at t = 0 (core.py:35-37) and expects NaN
    columns there (core.py:119)."""
    import c_oracle
    from skdownscale_amd import BcsdTemperature, GridArray, PointWiseDownscaler, synth

    if not c_oracle.available():
        pytest.skip("C oracle not built")
    T, ny, nx = 14600, 16, 16
    C = ny * nx
    index = synth.daily_calendar(T)
    gid = month_gid(index)
    cells = np.arange(C)
    X, y, Xp = (synth.tas_field(n, 11, index, cells, C) for n in ("X_hist", "y_obs", "X_fut"))
    dead = np.array([0, 17, 100, 255]) if masked else np.array([], dtype=np.int64)
    X[0, dead] = np.nan
    y[0, dead] = np.nan
    mk = lambda a: GridArray(a.reshape(T, ny, nx), ("time", "lat", "lon"), {"time": index, "lat": np.arange(ny), "lon": np.arange(nx)})  # noqa: E731
    pw = PointWiseDownscaler(BcsdTemperature(return_anoms=False))
    pw.fit(mk(X), mk(y))
    out = pw.predict(mk(Xp))
    assert out.dims == ("time", "lat", "lon") and out.values.shape == (T, ny, nx) and out.values.dtype == np.float64
    got = out.values.reshape(T, C)
    live = np.setdiff1d(cells, dead)
    exp, st = c_oracle.bcsd_fit_predict(0, X[:, live], y[:, live], Xp[:, live], gid, gid, return_anoms=False, nthreads=8)
    assert (st == 0).all()
    assert np.isnan(got[:, dead]).all() and not np.isnan(got[:, live]).any()
    assert_close(got[:, live], exp, what="config 1: 16 x 16 x 14 600 through PointWiseDownscaler vs C oracle")


def test_precipitation_fit_predict_with_empty_groups(ctx):
    """BcsdPrecipitation fit + predict with group ids that leave a middle group and the last group empty (G larger than the
    populated ids): the kernel that sorts only the wet days must skip them like the other fused kernels do (their order-table
    entries belong to the next group, or lie past the table)."""
    from skdownscale_amd import synth

    T, C = 14600, 11
    index = synth.daily_calendar(T)
    gid = month_gid(index).copy()
    gid[gid >= 5] += 1  # group 5 stays empty; 13 groups populated of G = 14: the last one is empty as well
    G = 14
    cells = np.arange(C)
    X, y, Xp = (synth.fill(synth.PRECIP, 5, s, np.arange(T), cells, C, amp=30.0, p_dry=0.5) for s in (20, 21, 22))
    y = y + 0.0
    y[3::5, :] += 0.25
    for ra in (False, True):
        exp, est = bo.pointwise_fit_predict(1, X, y, Xp, gid, gid, G=G, return_anoms=ra)
        out, st = ctx.bcsd_fit_predict(1, ctx.to_device(X), ctx.to_device(y), gid, G, ctx.to_device(Xp), gid, ra)
        assert np.array_equal(st, est) and (st == 0).all()
        assert_close(out.to_host(), exp, what=f"empty groups, return_anoms={ra}")


@pytest.mark.parametrize("C,c0,Ct", [(8, 0, 8), (10, 0, 10), (42, 0, 42), (62, 4, 70), (24, 6, 40)])
def test_dma_tile_kernel_against_the_register_tile_kernel(dev_ctx, monkeypatch, C, c0, Ct):
    """bcsd_fd_kernel (round 6: whole-lane months of a 40-year daily series, tiles landing by LDS-DMA in the lane-per-chunk
    layout, second-level keys, half of the y tile in flight during the sort of u) against bcsd_fx_kernel<20, true, true> (tiles
    through registers into per-cell rows): same arithmetic in the same order, so the fields must agree BIT FOR BIT -- for grids
    whose last tile is ragged (even cell counts: the tile is fetched shifted back over its predecessor), cell views of a wider
    resident field (leading dimension > cells, even offsets), a masked cell, a cell with a non-finite sample in each field,
    and cells with exactly tied shifted samples (work list).  Both variants of the kernel (SD_FD_LATE: the whole y tile behind
    the vote, second-level keys in the wave's own column).  Against the oracle for the first cells (bcsd.py:197-269)."""
    ctx = dev_ctx
    rng = np.random.default_rng(100 + C)
    T = 14600
    index = pd.date_range("1980-01-01", periods=T, freq="D")
    gid = month_gid(index)
    full = {k: 15 + 8 * rng.standard_normal((T, Ct)) for k in ("X", "y", "Xp")}
    sl = slice(c0, c0 + C)
    X, y, Xp = (full[k][:, sl] for k in ("X", "y", "Xp"))  # (views: the edits below land in `full`)
    if C > 8:
        X[0, 1] = np.nan                  # masked cell (core.py:35-37)
        X[777, 3] = np.inf                # non-finite x_hist
        y[5000, C - 3] = np.nan           # non-finite y_obs
        Xp[9000, C - 1] = -np.inf         # non-finite x_fut (last cell: the ragged tile)
        Xp[4000:4024, 5] = Xp[4000, 5]    # a constant stretch: exactly tied shifted samples -> work list
        Xp[120:150, C - 2] = Xp[120, C - 2]
    dev = {k: ctx.to_device(v) for k, v in full.items()}
    args = lambda: (dev["X"].cells(c0, c0 + C), dev["y"].cells(c0, c0 + C), gid, 12, dev["Xp"].cells(c0, c0 + C), gid)  # noqa: E731
    res = {}
    for name, env in (("dma", {}), ("dma_late", {"SD_FD_LATE": "1"}), ("regs", {"SD_FX_NODMA": "1"})):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        big = ctx.to_device(np.full((T, Ct), -777.0))
        ctx.prof_reset()
        ctx.prof_enable(True)
        _, st = ctx.bcsd_fit_predict(0, *args(), out=big.cells(c0, c0 + C))
        ctx.prof_enable(False)
        kernels = set(ctx.prof())
        assert ("bcsd_fd_kernel" in kernels) == (name != "regs"), (name, kernels)
        res[name] = (big.to_host(), st)
        for k_ in env:
            monkeypatch.delenv(k_)
    ref, st_ref = res["regs"]
    for name in ("dma", "dma_late"):
        got, st = res[name]
        assert np.array_equal(st, st_ref), name
        assert np.array_equal(got, ref, equal_nan=True), f"{name}: differs from the register-tile kernel"
    assert (np.delete(ref, np.s_[c0:c0 + C], axis=1) == -777.0).all()  # neighbours of the view untouched
    n = min(C, 8)
    exp, est = bo.pointwise_fit_predict(0, X[:, :n], y[:, :n], Xp[:, :n], gid, gid)
    assert np.array_equal(st_ref[:n], est)
    ok = est == 0
    assert_close(ref[:, sl][:, :n][:, ok], exp[:, ok], what="DMA tile kernel vs oracle")
