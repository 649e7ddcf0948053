"""CPU: host-side logic that needs no GPU -- grid driver plumbing, groupers, sharding (gloo, world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_groupers_and_group_keys():
    from skdownscale_amd.groupers import DAY_GROUPER, MONTH_GROUPER, group_keys

    idx = pd.date_range("1980-01-01", periods=14600, freq="D")
    k = group_keys(idx, MONTH_GROUPER)
    assert np.array_equal(k, [MONTH_GROUPER(x) for x in idx])
    sizes = np.bincount(k)[1:]
    assert sizes.tolist() == [1240, 1130, 1240, 1200, 1240, 1200, 1240, 1240, 1200, 1240, 1200, 1230]
    assert np.array_equal(group_keys(idx[:40], DAY_GROUPER), idx[:40].day)
    assert np.array_equal(group_keys(idx[:10], lambda x: x.year), [1980] * 10)


def test_estimator_params_and_clone():
    """The binding sklearn contract of the reference's test suite: check_estimator_cloneable."""
    from sklearn.base import clone

    from skdownscale_amd import AnalogRegression, BcsdPrecipitation, BcsdTemperature, PureAnalog

    for est in (BcsdTemperature(), BcsdPrecipitation(return_anoms=False), PureAnalog(n_analogs=7, kind="mean_analogs", thresh=0.1),
                AnalogRegression(n_analogs=12)):
        c = clone(est)
        assert type(c) is type(est) and c.get_params() == est.get_params()
        est.set_params(**est.get_params())
    assert PureAnalog.n_outputs == 3 and PureAnalog.output_names == ["pred", "exceedance_prob", "prediction_error"]
    assert set(BcsdTemperature().get_params()) == {"time_grouper", "climate_trend_grouper", "climate_trend", "return_anoms", "qm_kwargs"}


def test_pointwise_generic_loop_matches_reference_behaviour():
    """test_pointwise_runner.py:13-63 with a plain sklearn Pipeline: per-cell loop, sizes preserved, masked cells NaN."""
    from sklearn.linear_model import LinearRegression
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler

    from skdownscale_amd import GridArray, GridDataset, PointWiseDownscaler

    rng = np.random.default_rng(0)
    times = pd.date_range("1979-01-01", freq="1D", periods=100)
    a = rng.random((100, 2, 3))
    a[0, 1, 2] = np.nan  # masked cell
    X = GridDataset(a=GridArray(a, ("time", "y", "x"), {"time": times}))
    y = GridArray(X["a"].values * 3 + 1, ("time", "y", "x"), {"time": times})
    model = PointWiseDownscaler(Pipeline([("standardize", StandardScaler()), ("linear regression", LinearRegression())]))
    a[1:, 1, 2] = 0.5
    model.fit(X, y)
    out = model.predict(X)
    assert isinstance(out, GridArray) and out.sizes == {"time": 100, "y": 2, "x": 3}
    assert np.isnan(out.values[:, 1, 2]).all()
    ok = ~np.isnan(out.values)
    np.testing.assert_allclose(out.values[ok], y.values[ok], rtol=1e-9)
    with pytest.raises(TypeError):
        PointWiseDownscaler(object())
    with pytest.raises(ValueError, match="Expected at most 1 positional argument"):
        model.fit(X, y, y)


def test_unsupported_configurations_raise():
    from skdownscale_amd import BcsdTemperature
    from skdownscale_amd.bcsd import check_supported

    check_supported(BcsdTemperature())
    check_supported(BcsdTemperature(qm_kwargs={"qt_kwargs": {"n_endpoints": 10}}))
    check_supported(BcsdTemperature(qm_kwargs={"qt_kwargs": {"extrapolate": None, "n_endpoints": 3}}))
    nasanex = BcsdTemperature(time_grouper="daily_nasa-nex")
    nasanex._pre_fit()  # swaps PaddedDOYGrouper in (bcsd.py:36-38)
    check_supported(nasanex)
    from skdownscale_amd.groupers import DAY_GROUPER

    check_supported(BcsdTemperature(climate_trend=DAY_GROUPER))
    check_supported(BcsdTemperature(qm_kwargs={"detrend": True}))
    check_supported(BcsdTemperature(qm_kwargs={"detrend": True, "lt_kwargs": {"lr_kwargs": {"fit_intercept": True, "n_jobs": 2}}}))
    for bad in (BcsdTemperature(time_grouper="M"), BcsdTemperature(qm_kwargs={"detrend": True, "lt_kwargs": {"lr_kwargs": {"fit_intercept": False}}}),
                BcsdTemperature(qm_kwargs={"qt_kwargs": {"n_endpoints": 0}})):
        with pytest.raises(NotImplementedError):
            check_supported(bad)


def test_cell_partition():
    from skdownscale_amd.shard import cell_partition

    assert cell_partition(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert cell_partition(1_000_000, 8)[-1] == (875_000, 1_000_000)
    for C, W in [(1, 4), (7, 8), (100_000, 8)]:
        b = cell_partition(C, W)
        assert b[0][0] == 0 and b[-1][1] == C and all(b[i][1] == b[i + 1][0] for i in range(W - 1))


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [os.path.join(r"{root}", "scikit-downscale_amd"), os.path.join(r"{root}", "oracle"), os.path.join(r"{root}", "tests")]
import bcsd_oracle as bo
from skdownscale_amd import synth
from skdownscale_amd.shard import local_cells
from _torch_gather import gather_field
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
C, T = 7, 731
index = synth.daily_calendar(T)
gid = bo.month_group_id(index)
s, e = local_cells(C, world, rank)
cells = np.arange(s, e)
f = lambda n: synth.tas_field(n, 1, index, cells, C)
out, _ = bo.pointwise_fit_predict(bo.TAS, f("X_hist"), f("y_obs"), f("X_fut"), gid, gid)   # stand-in for the engine on CPU
full = gather_field(torch.from_numpy(out), C, dst=0)
if rank == 0:
    allc = np.arange(C)
    g = lambda n: synth.tas_field(n, 1, index, allc, C)
    exp, _ = bo.pointwise_fit_predict(bo.TAS, g("X_hist"), g("y_obs"), g("X_fut"), gid, gid)
    assert full.shape == (T, C) and np.array_equal(full.numpy(), exp), "sharded result differs from the single-rank result"
    print("GATHER_OK")
else:
    assert full is None
dist.destroy_process_group()
'''


def test_shard_cut_copies_only_its_slab(tmp_path):
    """ShardedPointWiseDownscaler cuts a rank's cell block out of the flattened spatial axes without flattening the grid first: a
    field whose spatial dims arrive in another order (a transposed view, here of a memory-mapped file) would otherwise be copied
    whole on every rank (core.py:86-93 iterates cells of the caller's array; nothing there copies the grid)."""
    import tracemalloc

    from skdownscale_amd.shard import _cell_block, cell_partition

    rng = np.random.default_rng(3)
    for shape, n_lead in (((5, 6, 7), 1), ((4, 2, 9, 5), 2), ((6, 11), 1), ((3, 4, 5, 6), 1), ((7,), 1)):
        a = rng.standard_normal(shape)
        perm = list(range(n_lead)) + list(range(len(shape) - 1, n_lead - 1, -1))  # spatial axes reversed: a strided view
        v = a.transpose(perm)
        C = int(np.prod(v.shape[n_lead:], dtype=np.int64)) if v.ndim > n_lead else 1
        flat = np.ascontiguousarray(v).reshape(v.shape[:n_lead] + (C,))
        for world in (1, 3, 8):
            for s, e in cell_partition(C, world):
                got = _cell_block(v, n_lead, s, e)
                assert got.flags.c_contiguous and np.array_equal(got, flat[..., s:e])
    # the full field is never materialised: 64 MB on disk, spatial axes transposed, one eighth of the cells
    T, ny, nx = 16, 512, 1024
    mm = np.lib.format.open_memmap(tmp_path / "field.npy", mode="w+", dtype=np.float64, shape=(T, nx, ny))
    mm[:] = np.arange(T * nx * ny, dtype=np.float64).reshape(T, nx, ny)
    v = mm.transpose(0, 2, 1)  # [T, ny, nx]
    s, e = cell_partition(ny * nx, 8)[5]
    tracemalloc.start()
    got = _cell_block(v, 1, s, e)
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert np.array_equal(got, np.ascontiguousarray(v).reshape(T, -1)[:, s:e])
    assert peak < 0.4 * mm.nbytes, (peak, mm.nbytes)  # slab + block (2/8 of the field), not the field


def test_sharded_gather_world_size_2_gloo(tmp_path):
    """N > 1 path on CPU: cells sharded over 2 ranks, ragged blocks, gather to root == unsharded result."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "GATHER_OK" in res.stdout


PRODUCT_WORKER = r'''
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(r"{root}", "scikit-downscale_amd"), os.path.join(r"{root}", "oracle"), os.path.join(r"{root}", "tests")]
import bcsd_oracle as bo
from skdownscale_amd import synth
from skdownscale_amd.shard import Rendezvous, cell_partition
from _host_comm import HostCommunicator
rdv = Rendezvous.from_env()                   # RANK / WORLD_SIZE / MASTER_* from the launcher, like bench.py
comm = HostCommunicator(rdv)                  # the product's gather layout over a test transport (tests/_host_comm.py)
rank, world = comm.rank, comm.world
C, T = 11, 400
index = synth.daily_calendar(T)
gid = bo.month_group_id(index)
bounds = cell_partition(C, world)             # ragged: 4 + 4 + 3 cells
s, e = bounds[rank]
f = lambda n, cells: synth.tas_field(n, 1, index, cells, C)
mine = np.arange(s, e)
out, _ = bo.pointwise_fit_predict(bo.TAS, f("X_hist", mine), f("y_obs", mine), f("X_fut", mine), gid, gid)  # stand-in for the engine
cells = np.array([b - a for a, b in bounds])
for buf in (None, np.full(T * C + 5, -1.0) if rank == 0 else None):   # engine-allocated and caller-supplied root buffer
    views = comm.gather_field(np.ascontiguousarray(out), cells, 0, buf)
    if rank == 0:
        allc = np.arange(C)
        exp, _ = bo.pointwise_fit_predict(bo.TAS, f("X_hist", allc), f("y_obs", allc), f("X_fut", allc), gid, gid)
        assert len(views) == world and [v.shape for v in views] == [(T, int(c)) for c in cells]
        base = views[0].base if views[0].base is not None else views[0]
        assert all(np.shares_memory(v, base) for v in views), "the per-rank views share the root buffer"
        # root layout [rank][T][C_r]: shard r starts T * sum(cells[:r]) doubles into the buffer
        flat = np.asarray(base).reshape(-1)
        for r, v in enumerate(views):
            off = T * int(cells[:r].sum())
            assert np.shares_memory(v, flat[off:off + v.size]) and np.array_equal(flat[off:off + v.size].reshape(v.shape), v)
        full = np.concatenate(views, axis=1)
        assert np.array_equal(full, exp), "sharded + gathered result differs from the unsharded result"
        if buf is not None:
            assert np.all(flat[T * C:] == -1.0), "nothing written past the gathered field"
    else:
        assert views is None
if rank == 0:
    for bad in ([4, 4], [[4, 4, 3]]):
        try:
            comm.gather_field(np.zeros((T, 4)), bad)
            raise SystemExit("cells of the wrong shape accepted")
        except ValueError:
            pass
    try:
        comm.gather_field(np.ascontiguousarray(out), cells, 0, np.empty(T * C - 1))
        raise SystemExit("short root buffer accepted")
    except ValueError:
        pass
    print("PRODUCT_GATHER_OK")
rdv.barrier()
rdv.close()
'''


def test_product_gather_layout_three_ranks_on_cpu(tmp_path):
    """The layout logic of the product gather (cell_partition -> [rank][T][C_r] root buffer -> per-rank views -> equality with
    the unsharded result) with three processes on CPU: tests/_host_comm.py puts a socket copy where ``Communicator`` calls RCCL,
    behind the same ``gather_field`` (same shared layout code, skdownscale_amd/shard.py:_GatherLayout)."""
    script = tmp_path / "worker.py"
    script.write_text(PRODUCT_WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                          "--master-port", "29641", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "PRODUCT_GATHER_OK" in res.stdout


SHARDED_PW_WORKER = r'''
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(r"{root}", "scikit-downscale_amd"), os.path.join(r"{root}", "tests")]
from sklearn.linear_model import LinearRegression
from skdownscale_amd import GridArray, PointWiseDownscaler
from skdownscale_amd.shard import Rendezvous, ShardedPointWiseDownscaler
from _host_comm import HostCommunicator
rdv = Rendezvous.from_env()
comm = HostCommunicator(rdv)
rng = np.random.default_rng(5)          # same inputs on every rank (SPMD)
X = GridArray(rng.standard_normal((40, 2, 3, 5)), ("time", "variable", "y", "x"), {{"y": np.arange(3.0), "x": np.arange(5.0)}})
y = GridArray(rng.standard_normal((40, 3, 5)), ("time", "y", "x"))
X.values[0, :, 1, 2] = np.nan            # a masked cell (core.py:35-37)
m = ShardedPointWiseDownscaler(LinearRegression(), comm=comm)   # 15 cells over 3 ranks; an sklearn estimator: the per-cell loop
m.fit(X, y.transpose("time", "x", "y"))  # y in another spatial dim order: aligned by name (core.py:86-93)
out = m.predict(X.transpose("time", "variable", "x", "y"))      # and so is X at predict (core.py:110-141)
try:
    m.predict(GridArray(X.values[:, :, :2], X.dims, {{"y": np.arange(2.0), "x": np.arange(5.0)}}))
    raise SystemExit("a grid of another shape was accepted")
except ValueError:
    pass
if comm.rank == 0:
    ref = PointWiseDownscaler(LinearRegression())
    ref.fit(X, y)
    exp = ref.predict(X)
    assert out.dims == exp.dims and out.shape == exp.shape, (out.dims, exp.dims)
    assert np.array_equal(np.isnan(out.values), np.isnan(exp.values)) and np.isnan(out.values[:, 1, 2]).all()
    ok = ~np.isnan(exp.values)
    assert np.array_equal(out.values[ok], exp.values[ok]), "sharded PointWiseDownscaler differs from the unsharded one"
    assert np.array_equal(out.coords["x"], np.arange(5.0))
else:
    assert out is None
# float32 grids come back as float32 (PointWiseDownscaler returns the input dtype); 2 cells over 3 ranks: one rank owns none
X32 = GridArray(X.values[:, :, :1, :2].astype(np.float32), X.dims, {{"y": np.arange(1.0), "x": np.arange(2.0)}})
y32 = GridArray(y.values[:, :1, :2].astype(np.float32), y.dims)
m2 = ShardedPointWiseDownscaler(LinearRegression(), comm=comm)
m2.fit(X32, y32)
out2 = m2.predict(X32)
if comm.rank == 0:
    ref2 = PointWiseDownscaler(LinearRegression())
    ref2.fit(X32, y32)
    exp2 = ref2.predict(X32)
    assert out2.dtype == exp2.dtype == np.float32 and np.array_equal(out2.values, exp2.values)
    print("SHARDED_PW_OK")
rdv.barrier()
rdv.close()
'''


def test_sharded_pointwise_downscaler_three_ranks_on_cpu(tmp_path):
    """The sharded drop-in surface (cells of the grid -> cell_partition -> per-rank PointWiseDownscaler -> gather to rank 0)
    with three processes on CPU: an sklearn estimator takes PointWiseDownscaler's per-cell loop, so no GPU is needed; on a
    GPU node the same class runs the batched estimators, one engine per rank."""
    script = tmp_path / "worker.py"
    script.write_text(SHARDED_PW_WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                          "--master-port", "29653", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "SHARDED_PW_OK" in res.stdout


def test_padded_doy_grouper_matches_the_reference():
    """groupers.py:19-89 through g12_padded_doy.npz (row positions of all 366 groups and their means, from the real
    reference), plus the assertion of the reference's own test_paddeddoygrouper (test_pointwise_models.py:302-312)."""
    import warnings

    from _cases import load
    from skdownscale_amd import PaddedDOYGrouper

    g = load("g12_padded_doy")
    for case in range(3):
        index = pd.date_range(start=str(g[f"start{case}"]), end=str(g[f"end{case}"]))
        X = pd.DataFrame({"foo": g[f"vals{case}"]}, index=index)
        pos_of = pd.Series(np.arange(len(index)), index=index)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            grouper = PaddedDOYGrouper(X, offset=int(g[f"offset{case}"]))
            groups = dict(list(grouper))
            means = grouper.mean()
        assert sorted(groups) == list(range(1, 367))
        rows, off = g[f"rows{case}"], g[f"off{case}"]
        for k in range(1, 367):
            assert np.array_equal(pos_of[groups[k].index].values, rows[off[k - 1]:off[k]]), (case, k)
        assert np.array_equal(means.values[:, 0], g[f"means{case}"]) and list(means.index) == list(range(1, 367))
    index = pd.date_range(start="1980-01-01", end="1982-12-31")
    groups = dict(list(PaddedDOYGrouper(pd.DataFrame({"foo": np.arange(len(index), dtype=float)}, index=index))))
    np.testing.assert_array_equal(np.unique(groups[123].index.dayofyear), np.arange(123 - 15, 123 + 16))


def test_bcsd_qt_kwargs_are_validated_like_the_reference():
    """qm_kwargs={'qt_kwargs': ...}: unknown keys fail like CunnaneTransformer.__init__ would, `alpha` / `beta` are accepted
    (and unused, quantile.py:462), `extrapolate` / `n_endpoints` are carried to the engine (no GPU needed for the checks)."""
    from skdownscale_amd import BcsdTemperature
    from skdownscale_amd.bcsd import qt_settings

    assert qt_settings(BcsdTemperature()) == ("both", 10)
    assert qt_settings(BcsdTemperature(qm_kwargs={"qt_kwargs": {"alpha": 0.3, "beta": 0.1}})) == ("both", 10)
    assert qt_settings(BcsdTemperature(qm_kwargs={"qt_kwargs": {"extrapolate": "min", "n_endpoints": 4}})) == ("min", 4)
    assert qt_settings(BcsdTemperature(qm_kwargs={"qt_kwargs": {"extrapolate": None}})) == (None, 10)
    assert qt_settings(BcsdTemperature(qm_kwargs={"qt_kwargs": {"extrapolate": "sideways"}})) == (None, 10)  # quantile.py:527-528
    with pytest.raises(TypeError, match="unexpected keyword argument 'gamma'"):
        qt_settings(BcsdTemperature(qm_kwargs={"qt_kwargs": {"gamma": 1}}))
    with pytest.raises(NotImplementedError):
        qt_settings(BcsdTemperature(qm_kwargs={"qt_kwargs": {"n_endpoints": 0}}))


def test_quantile_mapper_qt_kwargs_are_validated_like_the_reference():
    """the stand-alone QuantileMapper(qt_kwargs=...) (quantile.py:92, 136): the same checks and defaults as through
    BcsdTemperature(qm_kwargs=...), before anything touches the GPU; `lt_kwargs` of a detrended mapping is refused"""
    from skdownscale_amd import QuantileMapper

    assert QuantileMapper()._tails() == ("both", 10)
    assert QuantileMapper(qt_kwargs={"alpha": 0.3, "beta": 0.1})._tails() == ("both", 10)
    assert QuantileMapper(qt_kwargs={"extrapolate": "max", "n_endpoints": np.int64(7)})._tails() == ("max", 7)
    assert QuantileMapper(qt_kwargs={"extrapolate": "1to1"})._tails() == ("1to1", 10)
    assert QuantileMapper(qt_kwargs={"extrapolate": None})._tails() == (None, 10)
    assert QuantileMapper(qt_kwargs={"extrapolate": "sideways"})._tails() == (None, 10)  # quantile.py:527-528
    with pytest.raises(TypeError, match="unexpected keyword argument 'gamma'"):
        QuantileMapper(qt_kwargs={"gamma": 1}).fit(np.arange(10.0).reshape(-1, 1))
    for bad in (0, -3, 2.5, None):
        with pytest.raises(NotImplementedError, match="n_endpoints"):
            QuantileMapper(qt_kwargs={"n_endpoints": bad}).fit(np.arange(10.0).reshape(-1, 1))
    # lt_kwargs go to LinearTrendTransformer(**lt_kwargs) (quantile.py:96-97), which takes `lr_kwargs` only (trend.py:31): anything else is
    # the reference's TypeError; lr_kwargs that would change the fitted line are refused, result-neutral ones accepted
    with pytest.raises(TypeError, match="unexpected keyword argument 'fit_intercept'"):
        QuantileMapper(detrend=True, lt_kwargs={"fit_intercept": False}).fit(np.arange(10.0).reshape(-1, 1))
    with pytest.raises(NotImplementedError, match="lt_kwargs"):
        QuantileMapper(detrend=True, lt_kwargs={"lr_kwargs": {"fit_intercept": False}}).fit(np.arange(10.0).reshape(-1, 1))
    QuantileMapper(detrend=True, lt_kwargs={"lr_kwargs": {"fit_intercept": True, "copy_X": False}})._check()


def test_pure_regression_argument_checks():
    """PureRegression behaviour that needs no GPU (gard.py:402-412): parameters, refused configurations, fit state."""
    from sklearn.base import clone
    from sklearn.exceptions import NotFittedError

    from skdownscale_amd import PureRegression

    m = PureRegression()
    assert clone(m).get_params() == dict(thresh=None, logistic_kwargs=None, linear_kwargs=None)
    assert m.n_outputs == 3 and m.output_names == ["pred", "exceedance_prob", "prediction_error"]
    X, y = np.arange(10.0).reshape(-1, 1), np.arange(10.0)
    with pytest.raises(NotImplementedError, match="logistic_kwargs"):
        PureRegression(thresh=1.0, logistic_kwargs={"C": 10.0}).fit(X, y)
    with pytest.raises(NotImplementedError, match="linear_kwargs"):
        PureRegression(linear_kwargs={"fit_intercept": False}).fit(X, y)
    # keyword values that ask for the model the engine computes, or that do not change the fitted model, pass the check
    # (gard.py:389-402 forwards them to sklearn); anything that would change it is refused
    from skdownscale_amd.base import LINEAR_NEUTRAL, LOGISTIC_NEUTRAL, check_sklearn_kwargs

    for kw, table in ((None, LINEAR_NEUTRAL), ({"fit_intercept": True, "n_jobs": 4, "copy_X": False, "positive": False}, LINEAR_NEUTRAL),
                      ({"C": 1, "penalty": "l2", "tol": 1e-12, "max_iter": 1000, "solver": "newton-cg", "class_weight": None}, LOGISTIC_NEUTRAL)):
        check_sklearn_kwargs(kw, table, "kw", "model")
    PureRegression(thresh=1.0, logistic_kwargs={"C": 1.0, "max_iter": 500}, linear_kwargs={"n_jobs": 2})._check()
    for kw, table in (({"positive": True}, LINEAR_NEUTRAL), ({"fit_intercept": 1}, LINEAR_NEUTRAL), ({"unknown": 0}, LINEAR_NEUTRAL),
                      ({"penalty": "l1"}, LOGISTIC_NEUTRAL), ({"class_weight": "balanced"}, LOGISTIC_NEUTRAL), ({"C": True}, LOGISTIC_NEUTRAL)):
        with pytest.raises(NotImplementedError, match="not supported on the HIP engine"):
            check_sklearn_kwargs(kw, table, "kw", "model")
    with pytest.raises(NotFittedError):
        m.predict(X)
    with pytest.raises(ValueError, match="NaN"):
        m.fit(X * np.nan, y)
    with pytest.raises(ValueError, match="inconsistent numbers of samples"):
        m.fit(X, y[:5])


def test_cunnane_transformer_argument_checks():
    """CunnaneTransformer behaviour that needs no GPU (quantile.py:420-463): parameters, feature count, fit state."""
    from sklearn.base import clone
    from sklearn.exceptions import NotFittedError

    from skdownscale_amd import CunnaneGridModel, CunnaneTransformer

    t = CunnaneTransformer(extrapolate="max", n_endpoints=4)
    assert clone(t).get_params() == dict(alpha=0.4, beta=0.4, extrapolate="max", n_endpoints=4)
    with pytest.raises(ValueError, match="single feature"):
        t.fit(np.zeros((5, 2)))
    with pytest.raises(ValueError, match="unknown value for extrapolate"):
        CunnaneTransformer(extrapolate="sideways").fit(np.arange(5.0).reshape(-1, 1))
    with pytest.raises(ValueError, match="NaN"):
        t.fit(np.array([[1.0], [np.nan], [2.0]]))
    with pytest.raises(NotFittedError):
        t.inverse_transform(np.array([[0.5]]))
    with pytest.raises(ValueError, match="unknown value for extrapolate"):
        CunnaneGridModel("sideways", ctx=object())


def test_quantile_mapping_estimators_argument_checks():
    """Constructor / argument behaviour of the quantile-mapping regressors that needs no GPU (quantile.py:181-190, 576-593)."""
    from skdownscale_amd import EquidistantCdfMatcher, QuantileMappingReressor
    from skdownscale_amd.quantile import check_extrapolate, plotting_positions

    with pytest.raises(ValueError, match="n_endpoints"):
        QuantileMappingReressor(n_endpoints=1)
    with pytest.raises(ValueError, match="n_endpoints"):
        EquidistantCdfMatcher(n_endpoints=0)
    with pytest.raises(NotImplementedError, match="difference or ratio"):
        EquidistantCdfMatcher(kind="sum")
    for ex in (None, "1to1", "min", "max", "both"):
        check_extrapolate(ex)
    with pytest.raises(ValueError, match="unknown value for extrapolate"):
        check_extrapolate("sideways")
    with pytest.raises(ValueError, match="minimum of 21"):
        QuantileMappingReressor().fit(np.arange(10.0).reshape(-1, 1), np.arange(10.0))
    with pytest.raises(ValueError, match="maximum of 1"):
        QuantileMappingReressor().fit(np.zeros((30, 2)), np.arange(30.0))
    with pytest.raises(NotImplementedError, match="max_ratio"):
        EquidistantCdfMatcher(kind="ratio", max_ratio=5.0)._engine_code()
    from sklearn.base import clone

    m = clone(EquidistantCdfMatcher(kind="ratio", extrapolate="1to1", n_endpoints=4))
    assert (m.kind, m.extrapolate, m.n_endpoints, m.max_ratio) == ("ratio", "1to1", 4, None)
    assert np.allclose(plotting_positions(3), (np.arange(1, 4) - 0.4) / 3.2)


def test_pointwise_transform_and_inverse_transform_loop():
    """core.py:340-403 / 146-171: per-cell transformers through the generic loop (no engine involved)."""
    from sklearn.preprocessing import StandardScaler

    from skdownscale_amd import GridArray, PointWiseDownscaler

    rng = np.random.default_rng(0)
    X = 5 + 2 * rng.standard_normal((50, 2, 3))
    X[0, 1, 1] = np.nan  # masked cell (core.py:35-37)
    pw = PointWiseDownscaler(StandardScaler())
    pw.fit(GridArray(X, ("time", "y", "x")))
    out = pw.transform(GridArray(X, ("time", "y", "x")))
    assert out.dims == ("time", "variable", "y", "x") and out.shape == (50, 1, 2, 3)
    v = out.values[:, 0]
    assert np.isnan(v[:, 1, 1]).all()
    ok = np.ones((2, 3), bool)
    ok[1, 1] = False
    assert np.allclose(v[:, ok].mean(axis=0), 0.0, atol=1e-12) and np.allclose(v[:, ok].std(axis=0), 1.0)
    back = pw.inverse_transform(out)
    assert np.allclose(back.values[:, 0][:, ok], X[:, ok])
    with pytest.raises(ValueError, match="not fitted"):
        PointWiseDownscaler(StandardScaler()).transform(GridArray(X, ("time", "y", "x")))


def _runner_inputs(shape, n_vars, rng):
    """test/__init__.py:35-53 (random_point_data / random_grid_data) as GridArray / GridDataset"""
    from skdownscale_amd.core import GridArray, GridDataset

    dims = ("time", "point") if len(shape) == 1 else ("time", "y", "x")
    times = pd.date_range("2000-01-01", periods=100)
    ds = GridDataset()
    for i in range(n_vars):
        ds[chr(ord("a") + i)] = GridArray(rng.random((100,) + shape), dims, {"time": times})
    return ds


@pytest.mark.parametrize("shape,chunks", [((3,), None), ((2, 3), None), ((3,), {"point": 1}), ((2, 3), {"y": 1, "x": 1}), ((4, 5), {"y": 3, "x": 2})])
def test_pointwise_runner_restated(shape, chunks):
    """The reference's grid-driver tests (test_pointwise_runner.py:13-145) on GridArray / GridDataset with a scikit-learn
    pipeline (the per-cell loop of core.py:69-143), unchunked and chunked: chunked inputs are fitted and predicted block by
    block (core.py:256-262, 300-336) and give the same numbers and the same chunk structure."""
    from sklearn.linear_model import LinearRegression
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler

    from skdownscale_amd import PointWiseDownscaler
    from skdownscale_amd.core import GridArray, GridDataset, _BlockedModels

    rng = np.random.default_rng(len(shape))
    X, y = _runner_inputs(shape, 3, rng), _runner_inputs(shape, 1, rng)["a"]
    pipe = lambda: Pipeline([("scaler", StandardScaler()), ("lr", LinearRegression())])  # noqa: E731
    ref_model = PointWiseDownscaler(pipe())
    ref_model.fit(X, y)
    expected = ref_model.predict(X)
    if chunks:
        X, y = X.chunk(chunks), y.chunk(chunks)
    model = PointWiseDownscaler(pipe())
    model.fit(X, y)
    assert isinstance(model._models, _BlockedModels) == bool(chunks)
    y_pred = model.predict(X)
    assert isinstance(y_pred, GridArray) and y_pred.sizes == y.sizes  # test_pointwise_runner.py:48-52
    np.testing.assert_allclose(y_pred.values, expected.values, rtol=1e-12)
    if chunks:
        assert y_pred.chunks == y.chunks  # test_pointwise_runner.py:54-63
    # transform (test_pointwise_runner.py:66-90) and attributes (93-145)
    scaler = PointWiseDownscaler(StandardScaler())
    scaler.fit(X)
    xt = scaler.transform(X)
    assert xt.sizes["variable"] == 3 and {d: xt.sizes[d] for d in y.dims} == y.sizes
    np.testing.assert_allclose(scaler.inverse_transform(xt).values[:, 0], X["a"].values, rtol=1e-9)
    attrs = scaler.get_attr("n_features_in_", "int64")
    assert attrs.sizes == {d: y.sizes[d] for d in y.dims[1:]} and attrs.dtype == np.dtype("int64") and (attrs.values == 3).all()
    template = GridArray(np.zeros((3,) + shape), ("var",) + y.dims[1:], {"var": np.arange(3)})
    scale = scaler.get_attr("scale_", dtype="float64", template_output=template)
    assert scale.sizes == template.sizes and scale.dtype == np.dtype("float64")
    np.testing.assert_allclose(scale.values[0], X["a"].values.std(axis=0), rtol=1e-9)


def test_pointwise_runner_xarray():
    """The same through xarray objects (test_pointwise_runner.py:13-63): Dataset / DataArray in, DataArray out, chunked inputs
    block by block with the chunk structure preserved.  Where xarray is not installed (this container, the GPU box) the
    minimal stand-in of tests/xarray_stub runs instead -- it has xarray's dims / coords / isel / to_array / chunk metadata
    but nothing lazy."""
    xr = pytest.importorskip("xarray")
    from sklearn.linear_model import LinearRegression
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler

    from skdownscale_amd import PointWiseDownscaler

    rng = np.random.default_rng(0)
    times = pd.date_range("2000-01-01", periods=100)
    ds = xr.Dataset({k: (("time", "y", "x"), rng.random((100, 2, 3))) for k in "abc"}, coords={"time": times})
    y = xr.DataArray(rng.random((100, 2, 3)), dims=("time", "y", "x"), coords={"time": times})
    pipe = lambda: Pipeline([("scaler", StandardScaler()), ("lr", LinearRegression())])  # noqa: E731
    model = PointWiseDownscaler(pipe())
    model.fit(ds, y)
    pred = model.predict(ds)
    assert isinstance(pred, xr.DataArray) and dict(pred.sizes) == dict(y.sizes)
    expected = np.empty((100, 2, 3))
    X3 = np.stack([ds[k].values for k in "abc"], axis=1)  # [time, variable, y, x]
    for j in range(2):
        for i in range(3):
            expected[:, j, i] = pipe().fit(X3[:, :, j, i], y.values[:, j, i]).predict(X3[:, :, j, i])
    np.testing.assert_allclose(pred.values, expected, rtol=1e-12)
    # y with its spatial dims in another order is aligned by name (core.py:86-93 selects y[index] by dimension name)
    yt = xr.DataArray(y.values.transpose(0, 2, 1), dims=("time", "x", "y"), coords={"time": times})
    model_t = PointWiseDownscaler(pipe())
    model_t.fit(ds, yt)
    np.testing.assert_allclose(model_t.predict(ds).values, expected, rtol=1e-12)
    # transformers and attributes come back as xarray objects as well
    scaler = PointWiseDownscaler(StandardScaler())
    scaler.fit(ds)
    xt = scaler.transform(ds)
    assert isinstance(xt, xr.DataArray) and xt.sizes["variable"] == 3
    np.testing.assert_allclose(scaler.inverse_transform(xt).values[:, 0], ds["a"].values, rtol=1e-9)
    template = xr.DataArray(np.zeros((3, 2, 3)), dims=("var", "y", "x"), coords={"var": np.arange(3)})
    scale = scaler.get_attr("scale_", dtype="float64", template_output=template)
    assert isinstance(scale, xr.DataArray) and dict(scale.sizes) == {"var": 3, "y": 2, "x": 3}
    np.testing.assert_allclose(scale.values[0], ds["a"].values.std(axis=0), rtol=1e-9)
    if xr.__version__.endswith("stub"):
        chunked = True
    else:
        try:
            import dask  # noqa: F401
            chunked = True
        except ImportError:
            chunked = False
    if not chunked:
        return
    dsc, yc = ds.chunk({"y": 1, "x": 1}), y.chunk({"y": 1, "x": 1})
    model = PointWiseDownscaler(pipe())
    model.fit(dsc, yc)
    predc = model.predict(dsc)
    assert isinstance(predc, xr.DataArray) and dict(predc.sizes) == dict(y.sizes)
    assert predc.chunks == yc.chunks
    np.testing.assert_allclose(predc.values, pred.values, rtol=1e-12)


RDV_WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join(r"{root}", "scikit-downscale_amd"))
from skdownscale_amd.shard import Rendezvous
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rdv = Rendezvous(rank, world, addr="127.0.0.1", port=port, timeout=60.0)
payload = rdv.broadcast(bytes(range(128)) if rank == 0 else None)
assert payload == bytes(range(128))
for step in range(3):
    m = rdv.allreduce_max(10.0 * step + rank)
    assert m == 10.0 * step + world - 1, (rank, step, m)
rdv.barrier()
rdv.close()
print("ok", rank)
"""


def test_tcp_rendezvous_three_ranks(tmp_path):
    """the control plane of bench.py at N > 1 (unique-id broadcast, barrier, max-over-ranks clock): three processes on CPU"""
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rdv_worker.py"
    script.write_text(RDV_WORKER.format(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (2, 1, 0)]  # the clients start first and retry until rank 0 listens
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0 and out.startswith("ok"), (out, err)


def test_bench_cpu_legs_run_for_every_config():
    """bench.py's CPU side (per-config port, NumPy on one core and on one process per core, oracle result for the parity check)
    on a short series: every leg reports a positive rate, the checker receives the oracle's field for the first cells."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for kind, ndim in (("bcsd_tas", 2), ("bcsd_pr", 2), ("analog", 3)):
        seen = {}
        port, numpy_1, numpy_n, parity, c_port = bench.cpu_baseline(kind, 731, 0, 4096, 0.4, check=lambda exp: seen.setdefault("shape", exp.shape) and "ok")
        assert parity == "ok" and len(seen["shape"]) == ndim and seen["shape"][0] == 731
        for leg in (port, numpy_1, numpy_n):
            assert leg is not None and leg["value"] > 0 and leg["cores"] >= 1 and leg["unit"] == "cells/s"
        assert numpy_n["cores"] >= numpy_1["cores"]
        if kind != "analog":
            assert c_port is not None and "sd_oracle.c" in c_port["sample"]


def test_bench_scaling_curve_uses_one_per_gpu_workload_and_checks_its_host_legs():
    """bench.py: the N > 1 default runs the N = 1 line's cells per GPU (value(N) / value(1) is then a weak-scaling curve over identical
    per-GPU work; BASELINE configs[4]'s own 125 000 per GPU is a second leg), and the parity helper of the host-path legs
    (end_to_end / pointwise_end_to_end) accepts the oracle's own field and rejects a perturbed one."""
    import importlib.util

    import bcsd_oracle as bo
    from skdownscale_amd import synth

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.WORKLOADS[5]["cells"] == bench.WORKLOADS[2]["cells"] and bench.WORKLOADS[5]["cells_config5"] == 125_000
    T, n = 731, 5
    index = synth.daily_calendar(T)
    gid = (np.asarray(index.month) - 1).astype(np.int32)
    X, y, Xp = (synth.tas_field(name, 0, index, np.arange(n), 64) for name in ("X_hist", "y_obs", "X_fut"))
    exp, _ = bo.pointwise_fit_predict(0, X, y, Xp, gid, gid)
    assert bench.oracle_parity(X, y, Xp, gid, [exp, exp.copy()], [np.zeros(n, np.int32)] * 2).startswith("ok")
    bad = exp.copy()
    bad[100, 2] += 1e-3
    assert bench.oracle_parity(X, y, Xp, gid, [exp, bad]).startswith("FAILED")
    assert bench.oracle_parity(X, y, Xp, gid, [exp], [np.array([0, 0, 4, 0, 0])]).startswith("FAILED")
    assert bench.oracle_parity(X, y, Xp, gid, [exp[:, :3]]).startswith("FAILED shape")
