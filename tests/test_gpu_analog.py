"""GPU parity: HIP analog path vs golden KDTree queries / PureAnalog / AnalogRegression outputs."""
import numpy as np
import pandas as pd
import pytest

import analog_oracle as ao
from _cases import analog_inputs, assert_close, load

pytestmark = pytest.mark.gpu
KINDS = {"best_analog": 0, "sample_analogs": 1, "weight_analogs": 2, "mean_analogs": 3}


@pytest.fixture(scope="module")
def ctx():
    from skdownscale_amd.engine import default_context

    return default_context()


@pytest.mark.parametrize("F", [1, 3])
@pytest.mark.parametrize("resident", [False, True])
def test_neighbors_bit_exact(ctx, F, resident):
    g = load(f"g5_analog_F{F}")
    X, y, Xq = analog_inputs(g)
    st = ctx.analog_fit(ctx.to_device(X), ctx.to_device(y)) if resident else ctx.analog_fit(X, y)
    for k in (1, 30):
        if resident:
            _, _, inds, dist = ctx.analog_predict(st, ctx.to_device(Xq), k, 3, want_neighbors=True)
            inds, dist = inds.to_host(), dist.to_host()
        else:
            _, _, inds, dist = ctx.analog_predict(st, Xq, k, 3, want_neighbors=True)
        assert np.array_equal(inds, g[f"inds_k{k}"])  # bit-exact analog index selection
        assert np.array_equal(dist, g[f"dist_k{k}"])


@pytest.mark.parametrize("F", [1, 3])
@pytest.mark.parametrize("kind", list(KINDS))
def test_pure_analog_golden(ctx, F, kind):
    g = load(f"g5_analog_F{F}")
    X, y, Xq = analog_inputs(g)
    st = ctx.analog_fit(X, y)
    for k in (1, 30):
        kk, kc = (1, 0) if (kind == "best_analog" or k == 1) else (k, KINDS[kind])
        for thresh, tt in ((None, "none"), (0.0, "t0")):
            tag = f"{kind}_k{k}_{tt}"
            samp = g["samp_" + tag] if kind == "sample_analogs" else None
            out, status = ctx.analog_predict(st, Xq, kk, kc, thresh, samp)
            assert (status == 0).all()
            assert_close(out, g["out_" + tag], what=tag)


@pytest.mark.parametrize("F", [1, 3])
def test_analog_regression_golden(ctx, F):
    g = load(f"g5_analog_F{F}")
    X, y, Xq = analog_inputs(g)
    st = ctx.analog_fit(X, y)
    out, _ = ctx.analogreg_predict(st, Xq[: int(g["Tr"])], 30)
    assert_close(out, g["out_analogreg_k30"], what="analogreg")


@pytest.mark.parametrize("F,T,Tq,C,k", [(1, 50, 70, 5, 50), (2, 300, 33, 3, 7), (1, 1000, 1, 1, 200), (4, 129, 257, 2, 30)])
def test_vs_oracle_edge_sizes_and_ties(ctx, F, T, Tq, C, k):
    """k == T, single query, heavy exact ties (integer data): index order defined as (rdist, index)."""
    rng = np.random.default_rng(F * 100 + T)
    X = rng.integers(-5, 6, (T, F, C)).astype(np.float64)  # ties everywhere
    y = rng.standard_normal((T, C))
    Xq = rng.integers(-6, 7, (Tq, F, C)).astype(np.float64)
    st = ctx.analog_fit(X, y)
    out, status, inds, dist = ctx.analog_predict(st, Xq, k, 2, None, None, True)
    for c in range(C):
        d, i = ao.knn(X[:, :, c], Xq[:, :, c], k)
        assert np.array_equal(inds[:, :, c], i)
        assert np.array_equal(dist[:, :, c], d)
    exp = ao.pointwise_analog(X, y, Xq, k, ao.KIND_WEIGHT)
    assert_close(out, exp, what="ties weight")


@pytest.mark.parametrize("data", ["continuous", "zero_inflated", "quantized"])
def test_window_path_full_length_series(ctx, data):
    """F=1 without neighbour outputs takes the windowed kernel: T=14600 needs two LDS value ranges; zero-inflated
    and quantized predictors put exact ties on window boundaries and range pivots (those queries fall back to the
    (rdist, index)-ordered walk).  Every kind, with and without a threshold, and AnalogRegression vs the oracle."""
    rng = np.random.default_rng(5)
    T, Tq, C, k = 14600, 1500, 3, 30
    if data == "continuous":
        X, Xq = rng.standard_normal((T, 1, C)), 1.3 * rng.standard_normal((Tq, 1, C))
    elif data == "zero_inflated":
        X = rng.gamma(0.7, 4.0, (T, 1, C)) * (rng.random((T, 1, C)) > 0.55)
        Xq = rng.gamma(0.7, 4.0, (Tq, 1, C)) * (rng.random((Tq, 1, C)) > 0.55)
    else:
        X, Xq = np.round(rng.standard_normal((T, 1, C)), 1), np.round(rng.standard_normal((Tq, 1, C)), 1)
    y = 0.5 * X[:, 0, :] + rng.standard_normal((T, C))
    st = ctx.analog_fit(X, y)
    for kind in ("best_analog", "weight_analogs", "mean_analogs"):
        for thresh in (None, 0.1):
            kk = 1 if kind == "best_analog" else k  # gard.py:291-296: best_analog queries a single neighbour
            out, status = ctx.analog_predict(st, Xq, kk, KINDS[kind], thresh)
            exp = ao.pointwise_analog(X, y, Xq, kk, KINDS[kind], thresh)
            assert (status == 0).all()
            assert_close(out, exp, what=f"window {data} {kind} thresh={thresh}")
    out, _ = ctx.analogreg_predict(st, Xq, k)
    exp = ao.pointwise_analog(X, y, Xq[:250], k, ao.KIND_MEAN, regression=True)  # (the oracle solves one lstsq per query)
    assert_close(out[:250], exp, what=f"window {data} regression")


@pytest.mark.parametrize("k", [2, 3, 7, 8, 9, 17, 29, 31])
def test_window_reads_any_window_length(ctx, k):
    """weight_analogs and the thresholded kinds read the k analog values of a window two per load (pairs at any 8-byte
    boundary, a single load for the last value of an odd k): window lengths on either side of the batches of 8, windows that
    start at odd and even positions, at both ends of the sorted series"""
    rng = np.random.default_rng(400 + k)
    T, Tq, C = 1201, 600, 3
    X = rng.standard_normal((T, 1, C))
    Xq = np.concatenate([1.4 * rng.standard_normal((Tq - 4, 1, C)), np.array([-9.0, 9.0, -8.0, 8.0])[:, None, None] * np.ones((1, 1, C))])
    y = 0.5 * X[:, 0, :] + rng.standard_normal((T, C))
    st = ctx.analog_fit(X, y)
    for kind, thresh in (("weight_analogs", None), ("weight_analogs", 0.2), ("mean_analogs", 0.2)):
        out, status = ctx.analog_predict(st, Xq, k, KINDS[kind], thresh)
        assert (status == 0).all()
        assert_close(out, ao.pointwise_analog(X, y, Xq, k, KINDS[kind], thresh), what=f"k={k} {kind} thresh={thresh}")
        # the fused fit + predict entry point answers the same call
        both, _ = ctx.analog_fit_predict(X, y, Xq, k, KINDS[kind], thresh)
        assert np.array_equal(both, out, equal_nan=True), (k, kind, thresh)


@pytest.mark.parametrize("case", ["exact_line", "constant_y", "dry_spells", "k2", "short"])
def test_regression_prefix_path_edges(ctx, case):
    """One-feature AnalogRegression takes the prefix-sum kernel (k >= 3): noise-free lines and constant analog sets
    leave no residual for the prefix differences to resolve (direct summation branch); k = 2 stays on the
    windowed kernel."""
    rng = np.random.default_rng(17)
    T, Tq, C, k = 3000, 400, 4, 30
    X, Xq = rng.standard_normal((T, 1, C)), 1.2 * rng.standard_normal((Tq, 1, C))
    y = 0.5 * X[:, 0, :] + rng.standard_normal((T, C))
    if case == "exact_line":
        y = 2.0 * X[:, 0, :] + 1.0
    elif case == "constant_y":
        y = np.full((T, C), 3.25)
    elif case == "dry_spells":
        y = np.where(X[:, 0, :] > 0.3, y, 0.0)  # every window left of 0.3 holds identical analog values
    elif case == "k2":
        k = 2
    elif case == "short":
        T, k = 40, 5
        X, y = X[:T], y[:T]
    st = ctx.analog_fit(X, y)
    out, status = ctx.analogreg_predict(st, Xq, k)
    exp = ao.pointwise_analog(X, y, Xq, k, ao.KIND_MEAN, regression=True)
    assert (status == 0).all()
    assert_close(out, exp, what=f"regression {case}")


def test_single_analog_is_best_analog_whatever_the_kind(ctx):
    """gard.py:291-296: n_analogs == 1 turns every kind into 'best_analog' (the analog itself, also when a threshold masks
    it) -- at the C ABI as well, not only in the estimator class"""
    rng = np.random.default_rng(31)
    X, Xq = rng.standard_normal((500, 1, 3)), rng.standard_normal((120, 1, 3))
    y = rng.standard_normal((500, 3))
    st = ctx.analog_fit(X, y)
    for thresh in (None, 0.0):
        best, _ = ctx.analog_predict(st, Xq, 1, KINDS["best_analog"], thresh)
        assert_close(best, ao.pointwise_analog(X, y, Xq, 1, KINDS["best_analog"], thresh), what=f"best thresh={thresh}")
        for kind in ("mean_analogs", "weight_analogs"):
            out, _ = ctx.analog_predict(st, Xq, 1, KINDS[kind], thresh)
            assert np.array_equal(out, best, equal_nan=True), (kind, thresh)
            assert_close(out, ao.pointwise_analog(X, y, Xq, 1, KINDS[kind], thresh), what=f"{kind} k=1 thresh={thresh}")


@pytest.mark.parametrize("T,Tq,k", [(2000, 20000, 30), (16384, 16385, 1), (1024, 40000, 200), (700, 14600, 700)])
def test_mean_path_many_queries_and_size_limits(ctx, T, Tq, k):
    """the BASELINE kernel (mean_analogs without a threshold, or one analog) beyond one pass of queries (16 x 1024 per
    workgroup pass), at the longest series the index tags of the fit serve (16 384), with k up to the series length"""
    rng = np.random.default_rng(T + k)
    C = 2
    X, Xq = rng.standard_normal((T, 1, C)), 1.1 * rng.standard_normal((Tq, 1, C))
    y = 0.5 * X[:, 0, :] + rng.standard_normal((T, C))
    st = ctx.analog_fit(X, y)
    kind = "mean_analogs" if k > 1 else "best_analog"
    out, status = ctx.analog_predict(st, Xq, k, KINDS[kind])
    assert (status == 0).all()
    sel = np.unique(np.concatenate([np.arange(300), np.arange(Tq - 300, Tq), rng.integers(0, Tq, 400)]))  # (the oracle is a loop)
    exp = ao.pointwise_analog(X, y, Xq[sel], k, KINDS[kind], None)
    assert_close(out[sel], exp, what=f"mean path T={T} Tq={Tq} k={k}")
    assert np.isfinite(out).all()


@pytest.mark.parametrize("case", ["constant_y", "dry_spells", "wet_spells"])
def test_pure_analog_constant_windows_full_length(ctx, case):
    """PureAnalog mean / weight (gard.py:301-346) at the BASELINE series length with windows of identical analog values:
    the prefix-sum differences of the mean path are then sums of equal numbers (expected spread 0: what is left is rounding
    noise ~1e-8 of the data scale, inside the 1e-6 tolerance), with and without a threshold.  'dry_spells': y is 0 for every x below a cut (all windows left of it are
    constant, those across it mix zeros and values); 'wet_spells': the other side; 'constant_y': everywhere."""
    rng = np.random.default_rng(23)
    T, Tq, C, k = 14600, 1200, 3, 30
    X, Xq = rng.standard_normal((T, 1, C)), 1.2 * rng.standard_normal((Tq, 1, C))
    y = 0.5 * X[:, 0, :] + rng.gamma(0.8, 3.0, (T, C))
    if case == "constant_y":
        y = np.full((T, C), 3.25)
    elif case == "dry_spells":
        y = np.where(X[:, 0, :] > 0.3, y, 0.0)
    else:
        y = np.where(X[:, 0, :] < -0.2, y, 1.5)
    st = ctx.analog_fit(X, y)
    for kind in ("mean_analogs", "weight_analogs"):
        for thresh in (None, 0.0, 1.5):
            out, status = ctx.analog_predict(st, Xq, k, KINDS[kind], thresh)
            exp = ao.pointwise_analog(X, y, Xq, k, KINDS[kind], thresh)
            assert (status == 0).all()
            assert_close(out, exp, what=f"{case} {kind} thresh={thresh}")


@pytest.mark.parametrize("F,T,Tq,C,k,data", [
    (3, 5000, 700, 3, 30, "normal"),     # slab much narrower than the series
    (3, 6000, 5000, 2, 30, "normal"),    # enough queries for the class order by default
    (2, 3000, 333, 2, 7, "quantized"),   # ties on the sort axis and in the distances
    (4, 200, 65, 2, 200, "normal"),      # k == T: every point is a neighbour, both sides run to the ends
    (2, 64, 64, 2, 5, "normal"),         # one chunk
    (3, 200, 100, 4500, 4, "normal"),    # more cells than one staging chunk
    (3, 4000, 3000, 2, 30, "duplicates"),  # every training point four times: ties on the k-th distance (candidate lists hand back)
    (3, 3000, 1000, 2, 30, "huge"),      # squares beyond the float32 range: the pre-filter does not apply
    (3, 3000, 1000, 2, 29, "offset"),    # a large common offset: a coarse float32 image of the points, the selection stays exact
    (2, 3000, 700, 2, 31, "normal"),     # k above the candidate-list kernel's limit
    (5, 2500, 900, 2, 30, "normal"),
])
def test_slab_search_matches_full_scan(ctx, dev_ctx, monkeypatch, F, T, Tq, C, k, data):
    """F > 1: the feature-0 slab search (training points and queries sorted by feature 0, scan ends when the axis
    distance alone exceeds every k-th distance) selects bit-identical (rdist, index) lists to the full scan and
    to the oracle's brute force."""
    rng = np.random.default_rng(F * 1000 + T)
    X, Xq = rng.standard_normal((T, F, C)), 1.2 * rng.standard_normal((Tq, F, C))
    if data == "quantized":
        X, Xq = np.round(X * 4) / 4, np.round(Xq * 4) / 4  # dyadic grid: squared distances are exact
    if data == "duplicates":
        X[T // 4:] = np.concatenate([X[:T // 4]] * 3)
    if data == "huge":
        X, Xq = X * 1e17, Xq * 1e17
    if data == "offset":
        X, Xq = X + 3.0e5, Xq + 3.0e5
    y = rng.standard_normal((T, C))
    Xq[3, F - 1, 0] = np.nan  # one query without neighbours
    monkeypatch.delenv("SD_ANALOG_NOSLAB", raising=False)
    st = ctx.analog_fit(X, y)
    out, status, inds, dist = ctx.analog_predict(st, Xq, k, 3, want_neighbors=True)
    # the query order (classes by the other features, then feature 0) only groups the work: any class count agrees
    # (switches of the development library)
    monkeypatch.setenv("SD_ANALOG_SLAB_CLASSES", "8" if Tq < 4096 else "1")
    std = dev_ctx.analog_fit(X, y)
    out8, _, inds8, dist8 = dev_ctx.analog_predict(std, Xq, k, 3, want_neighbors=True)
    monkeypatch.delenv("SD_ANALOG_SLAB_CLASSES")
    ok8 = np.arange(Tq) != 3
    assert np.array_equal(inds8[ok8], inds[ok8]) and np.array_equal(dist8[ok8], dist[ok8]) and np.array_equal(out8[ok8], out[ok8])
    monkeypatch.setenv("SD_ANALOG_NOSLAB", "1")
    st0 = dev_ctx.analog_fit(X, y)
    out0, status0, inds0, dist0 = dev_ctx.analog_predict(st0, Xq, k, 3, want_neighbors=True)
    monkeypatch.delenv("SD_ANALOG_NOSLAB")
    ok = np.ones(Tq, bool)
    ok[3] = False
    assert status.tolist() == status0.tolist() and status[0] == 2
    assert np.array_equal(inds[ok], inds0[ok]) and np.array_equal(inds[3, :, 1:], inds0[3, :, 1:])
    assert np.array_equal(dist[ok], dist0[ok])
    assert np.array_equal(out[ok], out0[ok]) and np.isnan(out[3, :, 0]).all()
    # candidate lists pruned by the register network (k <= 30) against the LDS heap of the same slab scan
    monkeypatch.setenv("SD_ANALOG_HEAP", "1")
    outh, _, indsh, disth = dev_ctx.analog_predict(std, Xq, k, 3, want_neighbors=True)
    monkeypatch.delenv("SD_ANALOG_HEAP")
    assert np.array_equal(indsh[ok], inds[ok]) and np.array_equal(disth[ok], dist[ok]) and np.array_equal(outh[ok], out[ok])
    for c in range(min(C, 2)):
        d, i = ao.knn(X[:, :, c], Xq[ok][:, :, c], k)
        assert np.array_equal(inds[ok][:, :, c], i)
        assert np.array_equal(dist[ok][:, :, c], d)


def test_cell_shard_views(ctx):
    """Cell ranges of resident fields by pointer + leading dimension (odd offset, odd leading dimension)."""
    rng = np.random.default_rng(9)
    T, Tq, Ct, c0, C, k = 900, 300, 21, 5, 11, 30
    X, y, Xq = rng.standard_normal((T, 1, Ct)), rng.standard_normal((T, Ct)), rng.standard_normal((Tq, 1, Ct))
    dX, dy, dXq = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xq)
    st = ctx.analog_fit(dX.cells(c0, c0 + C), dy.cells(c0, c0 + C))
    big = ctx.to_device(np.full((Tq, 3, Ct), -777.0))
    for kind in (3, 2):
        out, status = ctx.analog_predict(st, dXq.cells(c0, c0 + C), k, kind, out=big.cells(c0, c0 + C))
        got = big.to_host()
        exp = ao.pointwise_analog(X[:, :, c0:c0 + C], y[:, c0:c0 + C], Xq[:, :, c0:c0 + C], k, kind)
        assert (status == 0).all()
        assert_close(got[:, :, c0:c0 + C], exp, what=f"analog view kind {kind}")
        assert (np.delete(got, np.s_[c0:c0 + C], axis=2) == -777.0).all()
    _, _, inds, _ = ctx.analog_predict(st, dXq.cells(c0, c0 + C), k, 3, want_neighbors=True)
    for c in range(C):
        _, i = ao.knn(X[:, :, c0 + c], Xq[:, :, c0 + c], k)
        assert np.array_equal(inds.to_host()[:, :, c], i)


def test_masked_cells_and_nan_query(ctx):
    g = load("g5_analog_F1")
    X, y, Xq = analog_inputs(g)
    X = np.concatenate([X, X[:, :, :1]], axis=2)
    y = np.concatenate([y, y[:, :1]], axis=1)
    Xq = np.concatenate([Xq, Xq[:, :, :1]], axis=2)
    X[0, 0, 1] = np.nan
    st = ctx.analog_fit(X, y)
    out, status = ctx.analog_predict(st, Xq, 30, 3)
    assert status.tolist() == [0, 1, 0] and np.isnan(out[:, :, 1]).all()
    assert_close(out[:, :, [0, 2]], g["out_mean_analogs_k30_none"][:, :, [0, 0]], what="masked neighbours")
    Xq[5, 0, 2] = np.nan
    _, status = ctx.analog_predict(st, Xq, 30, 3)
    assert status.tolist() == [0, 1, 2]


# ---- estimator surface (test_pointwise_models.py:144-200) ----

@pytest.fixture(scope="module")
def sample_X_y():
    n = 365
    index = pd.date_range("2019-01-01", periods=n)
    rng = np.random.default_rng(0)
    X = pd.DataFrame({"foo": np.sin(np.linspace(-10 * np.pi, 10 * np.pi, n)) * 10, "bar": rng.random(n)}, index=index)
    return X, X["foo"] + 2


@pytest.mark.parametrize("kind", list(KINDS))
def test_gard_analog_models(sample_X_y, kind):
    from skdownscale_amd import PureAnalog

    X, y = sample_X_y
    model = PureAnalog(kind=kind, n_analogs=3)
    model.fit(X, y)
    out = model.predict(X)
    assert len(out["pred"]) == len(out["prediction_error"]) == len(out["exceedance_prob"]) == len(X)
    assert (out["exceedance_prob"] == 1).all()
    model = PureAnalog(kind=kind, n_analogs=3, thresh=0)
    model.fit(X, y)
    prob = model.predict(X)["exceedance_prob"]
    assert (prob <= 1).all() and (prob >= 0).all()
    if kind != "sample_analogs":
        exp, _, _ = ao.pure_analog_predict(X.values, y.values, X.values, 3, ao.KIND_NAMES[kind], 0.0)
        assert_close(model.predict(X).values, exp, what=f"estimator {kind}")


def test_gard_models_default_and_regression(sample_X_y):
    import warnings

    from skdownscale_amd import AnalogRegression, PointWiseDownscaler, PureAnalog, GridArray

    X, y = sample_X_y
    assert len(PureAnalog().fit(X, y).predict(X)) == len(X)
    out = AnalogRegression().fit(X, y).predict(X)
    assert (out["exceedance_prob"] == 1).all() and list(out.columns) == ["pred", "exceedance_prob", "prediction_error"]
    exp, _ = ao.analog_regression_predict(X.values, y.values, X.values, 200)
    assert_close(out.values, exp, rtol=1e-6, what="analogreg estimator")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = PureAnalog(n_analogs=500, kind="mean_analogs").fit(X, y)
    assert m.k_ == 365 and any("n_analogs" in str(x.message) for x in w)
    with pytest.raises(NotImplementedError):
        AnalogRegression(lr_kwargs={"fit_intercept": False}).fit(X, y).predict(X)
    # grid driver: 3 output columns on a 'variable' axis (core.py:130-135)
    T = len(X)
    idx = X.index
    Xg = GridArray(np.repeat(X.values[:, :, None], 4, axis=2).reshape(T, 2, 2, 2), ("time", "variable", "y", "x"), {"time": idx})
    yg = GridArray(np.repeat(y.values[:, None], 4, axis=1).reshape(T, 2, 2), ("time", "y", "x"), {"time": idx})
    pw = PointWiseDownscaler(PureAnalog(n_analogs=10, kind="weight_analogs"))
    pw.fit(Xg, yg)
    res = pw.predict(Xg)
    assert res.dims == ("time", "variable", "y", "x") and res.shape == (T, 3, 2, 2)
    exp, _, _ = ao.pure_analog_predict(X.values, y.values, X.values, 10, ao.KIND_WEIGHT)
    assert_close(res.values[:, :, 1, 1], exp, what="pointwise analog")


def test_large_grid_is_consistent(ctx):
    """16 384 cells x 14 600 steps made of identical 2 048-cell blocks: every block (any chunk / workgroup of the
    windowed path) reproduces the first one bit for bit, and the first cells match the oracle."""
    from skdownscale_amd import synth

    T, C, B, k = 14600, 16384, 2048, 30
    fields = {}
    for name, stream, kw in (("X", 20, {}), ("y", 20, dict(amp=2.0, stream2=21, amp2=1.0)), ("Xq", 22, {})):
        d = ctx.empty((T, C))
        for c0 in range(0, C, B):
            ctx.synth_fill(d.cells(c0, c0 + B), synth.GAUSS, 5, stream, c_offset=0, c_full=B, **kw)
        fields[name] = d
    st = ctx.analog_fit(ctx.wrap(fields["X"].ptr, (T, 1, C)), fields["y"])
    out, status = ctx.analog_predict(st, ctx.wrap(fields["Xq"].ptr, (T, 1, C)), k, 3)
    assert (status == 0).all()
    rows = np.unique(np.linspace(0, T - 1, 40).astype(np.int64))
    got = np.stack([ctx.wrap(out.ptr + int(t) * 3 * C * 8, (3, C)).to_host() for t in rows])  # [rows, 3, C]
    for c0 in range(B, C, B):
        assert np.array_equal(got[:, :, c0:c0 + B], got[:, :, :B]), f"block at cell {c0} differs from block 0"
    n = 2
    Xh = fields["X"].cells(0, n).to_host()[:, None, :]
    yh = fields["y"].cells(0, n).to_host()
    Xqh = fields["Xq"].cells(0, n).to_host()[rows][:, None, :]
    assert_close(got[:, :, :n], ao.pointwise_analog(Xh, yh, Xqh, k, ao.KIND_MEAN), what="large grid vs oracle")
    st.close()
    for d in fields.values():
        d.free()
    out.free()


def test_full_size_grid_is_consistent(ctx):
    """BASELINE config 4 size (PureAnalog k = 30, F = 1, 100 000 cells x 14 600 steps, fit + predict) through the block
    property: identical 8 192-cell blocks reproduce block 0 bit for bit (any chunk of the fit's tile sort, any workgroup of
    the window search); the first cells match the oracle's brute-force neighbours; the fused fit + predict call reproduces it."""
    from skdownscale_amd import synth

    T, C, B, k = 14600, 100_000, 8192, 30
    fields = {}
    for name, stream, kw in (("X", 20, {}), ("y", 20, dict(amp=2.0, stream2=21, amp2=1.0)), ("Xq", 22, {})):
        d = ctx.empty((T, C))
        for c0 in range(0, C, B):
            ctx.synth_fill(d.cells(c0, min(C, c0 + B)), synth.GAUSS, 9, stream, c_offset=0, c_full=B, **kw)
        fields[name] = d
    st = ctx.analog_fit(ctx.wrap(fields["X"].ptr, (T, 1, C)), fields["y"])
    out, status = ctx.analog_predict(st, ctx.wrap(fields["Xq"].ptr, (T, 1, C)), k, 3)
    assert (status == 0).all()
    rows = np.unique(np.linspace(0, T - 1, 24).astype(np.int64))
    got = np.stack([ctx.wrap(out.ptr + int(t) * 3 * C * 8, (3, C)).to_host() for t in rows])  # [rows, 3, C]
    for c0 in range(B, C, B):
        c1 = min(C, c0 + B)
        assert np.array_equal(got[:, :, c0:c1], got[:, :, :c1 - c0]), f"block at cell {c0} differs from block 0"
    n = 2
    Xh = fields["X"].cells(0, n).to_host()[:, None, :]
    yh = fields["y"].cells(0, n).to_host()
    Xqh = fields["Xq"].cells(0, n).to_host()[rows][:, None, :]
    assert_close(got[:, :, :n], ao.pointwise_analog(Xh, yh, Xqh, k, ao.KIND_MEAN), what="full-size grid vs oracle")
    st.close()
    # the same grid through the fused call (what bench.py config 4 times): bit-identical on the sampled rows
    out2, status2 = ctx.analog_fit_predict(ctx.wrap(fields["X"].ptr, (T, 1, C)), fields["y"], ctx.wrap(fields["Xq"].ptr, (T, 1, C)), k, 3)
    assert (status2 == 0).all()
    for i, t in enumerate(rows):
        assert np.array_equal(ctx.wrap(out2.ptr + int(t) * 3 * C * 8, (3, C)).to_host(), got[i]), f"fused call differs in row {t}"
    out2.free()
    for d in fields.values():
        d.free()
    out.free()


def same_bits(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


@pytest.mark.parametrize("T,Tq,C", [(14600, 14600, 41), (13000, 9000, 19), (15360, 16384, 9), (14600, 20000, 5), (17000, 500, 3), (731, 400, 7)])
def test_fit_predict_in_one_call_is_bit_identical(ctx, T, Tq, C):
    """sd_analog_fit_predict*: fit + predict without a fitted state.  On long F = 1 series the per-cell workgroup that merged
    the sorted runs answers the queries itself (analog_f1_fused_kernel); the result equals analog_fit -> analog_predict bit for
    bit -- ragged tiles, a masked cell, a cell with a non-finite sample, a cell with a non-finite query, cells the fused kernel
    hands back (tied training values; a query with a tie on its window boundary) included; query series longer than one pass,
    series the tile-shaped fit does not serve and short series take the two calls internally.  Host buffers and resident
    fields, mean_analogs (k = 30), a single analog (k = 1, with and without a threshold) and the kinds the fused kernel does
    not serve."""
    rng = np.random.default_rng(T + Tq)
    X = rng.standard_normal((T, 1, C))
    y = 2.0 * X[:, 0, :] + rng.standard_normal((T, C))
    Xq = 1.1 * rng.standard_normal((Tq, 1, C))
    X[0, 0, 1] = np.nan                                  # masked cell (core.py:35-37)
    if C > 4:
        X[T // 2, 0, 2] = np.inf                         # a non-finite training sample
        Xq[Tq // 3, 0, 3] = np.nan                       # a non-finite query: its cell is reported, that row is NaN
        X[:, 0, 4] = np.round(X[:, 0, 4] * 8) / 8        # tied training values: handed back by the tag pass
    if C > 6:
        c = 6                                            # a tie on a window boundary: query midway between two training values, k = 1 ... 30
        order = np.argsort(X[:, 0, c])
        a, b = X[order[100], 0, c], X[order[101], 0, c]
        X[order[101], 0, c] = b = a + 0.25
        Xq[5, 0, c] = a + 0.125
    cases = [(30, 3, None), (1, 3, None), (1, 0, 0.3), (30, 2, None), (30, 3, 0.3), (2, 3, None)]
    for k, kind, thresh in cases:
        st = ctx.analog_fit(X, y)
        ref, sref = ctx.analog_predict(st, Xq, k, kind, thresh=thresh)
        st.close()
        got, sgot = ctx.analog_fit_predict(X, y, Xq, k, kind, thresh=thresh)
        assert sgot.tolist() == sref.tolist(), (k, kind, thresh)
        assert same_bits(got, ref), (k, kind, thresh)
    if C > 4:
        assert sref[1] == 1 and sref[2] == 2 and sref[3] == 2
    dX, dy, dq = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xq)
    st = ctx.analog_fit(dX, dy)
    ref, sref = ctx.analog_predict(st, dq, 30, 3)
    got, sgot = ctx.analog_fit_predict(dX, dy, dq, 30, 3)
    assert sgot.tolist() == sref.tolist() and same_bits(got.to_host(), ref.to_host())
    # queries that are a cell range of a wider resident field (their pitch differs from the training fields'): fused and split path
    wide = ctx.to_device(np.concatenate([Xq[:, 0, :], Xq[:, 0, :]], axis=1))  # [Tq, 2 C]
    from skdownscale_amd.engine import DeviceArray

    qv = DeviceArray(ctx, (Tq, 1, C), dptr=wide.cells(C, 2 * C).ptr, owner=False, ld=2 * C)
    for kind in (3, 2):
        g2, s2 = ctx.analog_fit_predict(dX, dy, qv, 30, kind)
        r2, rs2 = ctx.analog_predict(st, dq, 30, kind)
        assert s2.tolist() == rs2.tolist() and same_bits(g2.to_host(), r2.to_host()), kind
    live = [c for c in range(C) if c not in (1, 2)]
    rows = np.unique(np.linspace(0, Tq - 1, 12).astype(np.int64))
    assert_close(got.to_host()[rows][:, :, live[:2]], ao.pointwise_analog(X[:, :, live[:2]], y[:, live[:2]], Xq[rows][:, :, live[:2]], 30, ao.KIND_MEAN),
                 what="fit_predict vs oracle")
    st.close()


def test_fit_predict_random_sizes(ctx):
    """Random series lengths around the widths of the tile-shaped fit (K = 13 / 15, partly filled last runs), query counts up to
    one pass, k from 1 to 200, continuous and quantised data (quantised: every cell is handed back): fused call == fit -> predict."""
    rng = np.random.default_rng(404)
    for case in range(14):
        T = int(rng.integers(9300, 15361))
        Tq = int(rng.integers(1, 16385))
        C = int(rng.integers(1, 20))
        k = int(rng.choice([1, 2, 7, 30, 200]))
        X = rng.standard_normal((T, 1, C))
        if case % 5 == 4:
            X = np.round(X * 64) / 64
        y = rng.standard_normal((T, C)) + X[:, 0, :]
        Xq = 1.3 * rng.standard_normal((Tq, 1, C))
        st = ctx.analog_fit(X, y)
        ref, sref = ctx.analog_predict(st, Xq, k, 3)
        st.close()
        got, sgot = ctx.analog_fit_predict(X, y, Xq, k, 3)
        assert sgot.tolist() == sref.tolist() and same_bits(got, ref), (case, T, Tq, C, k)


def test_fit_predict_large_grid_is_bit_identical(ctx):
    """16 384 cells x 14 600 steps, k = 30: the fused call against fit -> predict on the same resident fields, every cell and
    sampled rows; a few cells with tied training values exercise the hand-back (packed columns through the split path)."""
    from skdownscale_amd import synth

    T, C, k = 14600, 16384, 30
    fields = {}
    for name, stream, kw in (("X", 20, {}), ("y", 20, dict(amp=2.0, stream2=21, amp2=1.0)), ("Xq", 22, {})):
        d = ctx.empty((T, C))
        ctx.synth_fill(d, synth.GAUSS, 5, stream, c_offset=0, c_full=C, **kw)
        fields[name] = d
    for c in (7, 4099, 16383):  # tied training values (the exact zeros of a dry-day series) in three cells
        ctx.synth_fill(fields["X"].cells(c, c + 1), synth.PRECIP, 5, 23, c_offset=c, c_full=C, p_dry=0.5)
    X3, Xq3 = ctx.wrap(fields["X"].ptr, (T, 1, C)), ctx.wrap(fields["Xq"].ptr, (T, 1, C))
    st = ctx.analog_fit(X3, fields["y"])
    ref, sref = ctx.analog_predict(st, Xq3, k, 3)
    st.close()
    got, sgot = ctx.analog_fit_predict(X3, fields["y"], Xq3, k, 3)
    assert sgot.tolist() == sref.tolist() and (sgot == 0).all()
    rows = np.unique(np.linspace(0, T - 1, 48).astype(np.int64))
    for t in rows:
        a = ctx.wrap(got.ptr + int(t) * 3 * C * 8, (3, C)).to_host()
        b = ctx.wrap(ref.ptr + int(t) * 3 * C * 8, (3, C)).to_host()
        assert same_bits(a, b), f"row {t}"
    for d in fields.values():
        d.free()
    got.free()
    ref.free()


PROB_TIGHT = 1e-6    # exceedance probability vs the reference's objective solved tightly (logistic_kwargs tol=1e-12)
PROB_DEFAULT = 1e-3  # ... vs the reference's default LogisticRegression: its L-BFGS stops at tol=1e-4, within ~2e-4 of the optimum


@pytest.mark.parametrize("case", [0, 1])
def test_analog_regression_thresh_golden(ctx, case):
    """g14_thresholded_regressions.npz: AnalogRegression(thresh) (gard.py:201-219) -- per query a logistic regression on its
    analogs (exceedance_prob = predict_proba(x)[0, 0], as the reference writes it) and the linear model on the exceeding
    analogs; one and three features (window walk and slab search)."""
    g = load("g14_thresholded_regressions")
    X, y, Xq, k, thresh = g[f"ar_X{case}"], g[f"ar_y{case}"], g[f"ar_Xq{case}"], int(g[f"ar_k{case}"]), float(g[f"ar_thresh{case}"])
    st = ctx.analog_fit(X, y)
    out, status = ctx.analogreg_predict(st, Xq, k, thresh)
    assert (status == 0).all()
    dout, _ = ctx.analogreg_predict(ctx.analog_fit(ctx.to_device(X), ctx.to_device(y)), ctx.to_device(Xq), k, thresh)
    assert np.array_equal(dout.to_host(), out)
    for name, tol in (("tight", PROB_TIGHT), ("default", PROB_DEFAULT)):
        exp = g[f"ar_out{case}_{name}"]
        assert_close(out[:, 0], exp[:, 0], what=f"pred case {case}")
        assert_close(out[:, 2], exp[:, 2], what=f"error case {case}")
        assert np.abs(out[:, 1] - exp[:, 1]).max() <= tol, (case, name, np.abs(out[:, 1] - exp[:, 1]).max())
    exp = ao.pointwise_analog(X, y, Xq, k, None, thresh=thresh, regression=True)
    assert np.abs(out[:, 1] - exp[:, 1]).max() <= 1e-8
    # without a threshold nothing changes
    out0, _ = ctx.analogreg_predict(st, Xq, k)
    assert_close(out0, ao.pointwise_analog(X, y, Xq, k, None, regression=True), what="thresh=None")


def test_analog_regression_thresh_surface():
    """AnalogRegression(thresh) as an estimator and through the grid driver -- the reference's driver test
    (test_pointwise_runner.py:13-63: PointWiseDownscaler(AnalogRegression(thresh=0)) on 3-point and 2x3 grids of uniform
    random data, 100 days) restated with GridArray; the one-class failure of the reference (gard.py:204-207)."""
    from skdownscale_amd import AnalogRegression, PointWiseDownscaler
    from skdownscale_amd.core import GridArray

    rng = np.random.default_rng(3)
    times = pd.date_range("2000-01-01", periods=100)
    for dims, shape in ((("time", "point"), (100, 3)), (("time", "y", "x"), (100, 2, 3))):
        X = GridArray(rng.random(shape), dims, {"time": times})
        y = GridArray(rng.random(shape), dims, {"time": times})
        model = PointWiseDownscaler(AnalogRegression(thresh=0))
        model.fit(X, y)
        y_pred = model.predict(X)
        assert isinstance(y_pred, GridArray)
        assert y_pred.sizes["variable"] == 3 and list(y_pred.coords["variable"]) == ["pred", "exceedance_prob", "prediction_error"]
        assert y_pred.sizes["time"] == 100 and tuple(y_pred.shape[2:]) == shape[1:]
        assert (y_pred.values[:, 1] == 1.0).all()  # every analog of uniform(0,1) data exceeds 0
    g = load("g14_thresholded_regressions")
    X, y, Xq, k, thresh = g["ar_X0"], g["ar_y0"], g["ar_Xq0"], int(g["ar_k0"]), float(g["ar_thresh0"])
    m = AnalogRegression(n_analogs=k, thresh=thresh).fit(pd.DataFrame(X[:, :, 0]), y[:, 0])
    out = m.predict(pd.DataFrame(Xq[:, :, 0]))
    assert list(out.columns) == ["pred", "exceedance_prob", "prediction_error"]
    assert np.abs(out.values[:, 1] - g["ar_out0_tight"][:, 1, 0]).max() <= PROB_TIGHT
    with pytest.raises(ValueError, match="only one class: np.int8\\(0\\)"):
        AnalogRegression(n_analogs=10, thresh=1e9).fit(X[:, :, 0], y[:, 0]).predict(Xq[:5, :, 0])
    with pytest.raises(NotImplementedError, match="metric"):
        AnalogRegression(kdtree_kwargs={"metric": "manhattan"}).fit(X[:, :, 0], y[:, 0])


@pytest.mark.parametrize("T", [14600, 16384, 13400])
def test_tile_shaped_fit_stage_is_bit_identical(ctx, dev_ctx, monkeypatch, T):
    """F = 1 fit of a long series: the tile-shaped first stage (analog_tile_sort_kernel: cell-major copies + sorted runs of 64 K
    tagged keys, then only the merge rounds 6 .. in analog_sort2_kernel) leaves the same fitted state as the two staging
    transposes + the full per-cell sort (SD_ANALOG_NOTILE of the development library): predictions, statistics, neighbour
    indices and distances of every kind agree bit for bit -- ragged last tile, a masked cell, a cell with a non-finite sample,
    cells with tied / signed-zero values (exact two-sort kernel) included -- and match the oracle's brute force."""
    rng = np.random.default_rng(T)
    C, Tq, k = 19, 700, 30
    X = rng.standard_normal((T, 1, C))
    y = 2.0 * X[:, 0, :] + rng.standard_normal((T, C))
    X[0, 0, 4] = np.nan                      # masked cell (core.py:35-37)
    X[T // 2, 0, 9] = np.inf                 # a non-finite sample: the cell is reported, its outputs are NaN
    X[:, 0, 12] = np.round(X[:, 0, 12] * 8) / 8   # ties (and both signed zeros): the exact two-sort kernel
    X[5, 0, 12], X[6, 0, 12] = 0.0, -0.0
    Xq = 1.1 * rng.standard_normal((Tq, 1, C))
    monkeypatch.delenv("SD_ANALOG_NOTILE", raising=False)
    st = ctx.analog_fit(X, y)
    monkeypatch.setenv("SD_ANALOG_NOTILE", "1")
    st0 = dev_ctx.analog_fit(X, y)
    monkeypatch.delenv("SD_ANALOG_NOTILE")
    for kind in (0, 2, 3):  # best, weight, mean
        a, sa = ctx.analog_predict(st, Xq, k, kind)
        b, sb = dev_ctx.analog_predict(st0, Xq, k, kind)
        assert sa.tolist() == sb.tolist() and sa[4] == 1 and sa[9] == 2
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), kind
    a, sa, ia, da = ctx.analog_predict(st, Xq, k, 3, want_neighbors=True)
    b, sb, ib, db = dev_ctx.analog_predict(st0, Xq, k, 3, want_neighbors=True)
    live = [c for c in range(C) if c not in (4, 9)]
    assert np.array_equal(ia[:, :, live], ib[:, :, live]) and np.array_equal(da[:, :, live], db[:, :, live])
    for c in (0, 12, 18):
        d, i = ao.knn(X[:, :, c], Xq[:, :, c], k)
        assert np.array_equal(ia[:, :, c], i) and np.array_equal(da[:, :, c], d)


@pytest.mark.parametrize("k", [5, 30, 64])
def test_regression_direct_window_sums_against_prefix_differences(ctx, dev_ctx, monkeypatch, k):
    """One-feature AnalogRegression with a window of k <= 64 analogs sums the window directly (reg_batch in analog_f1_mean_kernel); the
    development library's SD_ANALOG_REG_PREFIX keeps the form it replaces (prefix differences, what n_analogs > 64 still takes).
    Both are restatements of the same least squares: they agree to rounding on continuous data -- long query series (value-ordered
    runs) and short ones, a masked cell, a cell with a non-finite sample --, and both agree with the oracle's lstsq per query.
    (The straight-line batches of 'weight_analogs' need no such pair: same operations in the same order, covered bit for bit by every
    weight_analogs case of this file.)"""
    rng = np.random.default_rng(100 + k)
    T, C = 6000, 7
    X = rng.standard_normal((T, 1, C))
    y = 0.8 * X[:, 0, :] + 0.5 * rng.standard_normal((T, C))
    X[0, 0, 2] = np.nan
    y[17, 5] = np.inf
    st = ctx.analog_fit(X, y)
    monkeypatch.setenv("SD_ANALOG_REG_PREFIX", "1")
    st0 = dev_ctx.analog_fit(X, y)
    for Tq in (300, 5000):
        Xq = 1.1 * rng.standard_normal((Tq, 1, C))
        monkeypatch.delenv("SD_ANALOG_REG_PREFIX", raising=False)
        a, sa = ctx.analogreg_predict(st, Xq, k)
        monkeypatch.setenv("SD_ANALOG_REG_PREFIX", "1")
        b, sb = dev_ctx.analogreg_predict(st0, Xq, k)
        assert sa.tolist() == sb.tolist() and sa[2] == 1 and sa[5] == 2
        # (the prefix differences carry an absolute error of ~1e-16 of the cell totals: ~1e-8 of a five-analog window's residual sum)
        assert_close(a, b, rtol=1e-7, what=f"direct vs prefix k={k} Tq={Tq}")
        sel = np.arange(0, Tq, max(1, Tq // 150))
        exp = ao.pointwise_analog(X[:, :, [0, 6]], y[:, [0, 6]], Xq[sel][:, :, [0, 6]], k, ao.KIND_MEAN, regression=True)
        assert_close(a[sel][:, :, [0, 6]], exp, what=f"direct vs oracle k={k} Tq={Tq}")
    monkeypatch.delenv("SD_ANALOG_REG_PREFIX", raising=False)
