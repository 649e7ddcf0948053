"""GPU: error behaviour of the C ABI (argument checks, limits) as seen through the ctypes layer."""
import ctypes as C
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from skdownscale_amd.engine import default_context

    return default_context()


def test_bcsd_argument_errors(ctx):
    rng = np.random.default_rng(0)
    X, y = rng.standard_normal((100, 3)), rng.standard_normal((100, 3))
    gid = (np.arange(100) % 12).astype(np.int32)
    with pytest.raises(ValueError, match="sd_downscale"):
        ctx.bcsd_fit(7, X, y, gid, 12, True)  # unknown kind
    bad = gid.copy()
    bad[5] = 12  # group id out of range
    with pytest.raises(ValueError, match="group ids must lie"):  # the wrapper checks what the C ABI cannot (buffer sizes) and what it can
        ctx.bcsd_fit(0, X, y, bad, 12, True)
    import ctypes

    h = ctypes.c_void_p()
    rc = ctx.lib.sd_bcsd_fit(ctx.handle, 0, X.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                             bad.ctypes.data_as(ctypes.c_void_p), 12, 100, 3, 1, ctypes.byref(h))
    assert rc == 1 and b"outside" in ctx.lib.sd_last_error()  # ... and so does the C ABI itself
    with pytest.raises(ValueError, match="expected a \\[100, 3\\] field"):
        ctx.bcsd_fit(0, X[:50], y, gid, 12, True)
    with pytest.raises(ValueError, match="sd_downscale"):
        ctx.bcsd_fit(0, None, y, gid, 12, True)  # BcsdTemperature needs X
    st = ctx.bcsd_fit(0, X, y, gid, 12, True)
    bad_p = np.full(50, -1, dtype=np.int32)
    with pytest.raises(ValueError, match="group ids must lie"):
        ctx.bcsd_predict(st, rng.standard_normal((50, 3)), bad_p)
    with pytest.raises(ValueError, match="expected a"):
        ctx.bcsd_predict(st, rng.standard_normal((50, 4)), gid[:50])
    # a NULL state / context is an argument error, not a crash
    rc = ctx.lib.sd_bcsd_state_info(None, None, None, None, None, None)
    assert rc == 1 and b"NULL" in ctx.lib.sd_last_error() or rc == 1
    rc = ctx.lib.sd_ctx_synchronize(None)
    assert rc == 1
    # an empty month in the predict series is fine (that month simply has no samples)
    gid_p = np.where(gid == 3, 4, gid).astype(np.int32)
    out, status = ctx.bcsd_predict(st, rng.standard_normal((100, 3)), gid_p)
    assert np.isfinite(out).all() and (status == 0).all()


def test_analog_argument_errors(ctx):
    rng = np.random.default_rng(1)
    X, y, Xq = rng.standard_normal((40, 1, 2)), rng.standard_normal((40, 2)), rng.standard_normal((10, 1, 2))
    st = ctx.analog_fit(X, y)
    with pytest.raises(ValueError, match="k=41"):
        ctx.analog_predict(st, Xq, 41, 3)  # more analogs than training samples
    with pytest.raises(ValueError, match="bad sizes|k=0"):
        ctx.analog_predict(st, Xq, 0, 3)
    with pytest.raises(ValueError, match="unknown kind"):
        ctx.analog_predict(st, Xq, 5, 9)
    with pytest.raises(ValueError, match="sample_inds"):
        ctx.analog_predict(st, Xq, 5, 1)  # sample_analogs without the sampled indices
    with pytest.raises(ValueError, match="F=9"):
        ctx.analog_fit(rng.standard_normal((40, 9, 2)), y)  # more features than the engine supports
    out, status = ctx.analog_predict(st, Xq, 40, 3)  # k == T is allowed
    assert np.isfinite(out).all()
    # shapes the C ABI cannot check (raw pointers): refused by the wrappers
    with pytest.raises(ValueError, match="expected X"):
        ctx.analog_fit(X, y[:30])
    with pytest.raises(ValueError, match="Xq: expected"):
        ctx.analog_predict(st, Xq[:, :, :1], 5, 3)
    with pytest.raises(ValueError, match="Xq: expected"):
        ctx.analogreg_predict(st, rng.standard_normal((10, 2, 2)), 5)
    with pytest.raises(ValueError, match="sample_inds: expected shape"):
        ctx.analog_predict(st, Xq, 5, 1, sample_inds=np.zeros((3, 2), np.int32))
    with pytest.raises(ValueError, match=r"sample_inds must lie in \[0, 5\)"):
        ctx.analog_predict(st, Xq, 5, 1, sample_inds=np.full((10, 2), 5, np.int32))


def test_qm_limits(ctx):
    rng = np.random.default_rng(2)
    T = 20000  # beyond the workgroup sort (19 456 samples)
    with pytest.raises(NotImplementedError, match="19456"):
        ctx.qm_fit(rng.standard_normal((T, 1)), rng.standard_normal((T, 1)))
    st = ctx.qm_fit(rng.standard_normal((50, 2)), rng.standard_normal((50, 2)))
    with pytest.raises(ValueError, match="unknown model"):
        ctx.qm_predict(st, 5, rng.standard_normal((10, 2)))
    with pytest.raises(NotImplementedError, match="19456"):
        ctx.qm_predict(st, 1, rng.standard_normal((T, 2)))  # EquidistantCdfMatcher ranks the new series
    out, _ = ctx.qm_predict(st, 0, rng.standard_normal((T, 2)))  # the regressor itself has no such limit
    assert np.isfinite(out).all()
    with pytest.raises(ValueError, match="y: expected"):
        ctx.qm_fit(rng.standard_normal((50, 2)), rng.standard_normal((40, 2)))
    with pytest.raises(ValueError, match="X: expected"):
        ctx.qm_predict(st, 0, rng.standard_normal((10, 3)))
    cst = ctx.qm_fit(rng.standard_normal((50, 2)))
    with pytest.raises(ValueError, match="X: expected"):
        ctx.qm_cunnane(cst, 0, rng.standard_normal((10, 1)))
    with pytest.raises(ValueError, match="unknown value for extrapolate"):
        ctx.qm_cunnane(cst, 0, rng.standard_normal((10, 2)), extrapolate="sideways")


def test_state_use_after_destroy_and_release_cached(ctx):
    rng = np.random.default_rng(3)
    X, y = rng.standard_normal((60, 2)), rng.standard_normal((60, 2))
    st = ctx.bcsd_fit(0, X, y, (np.arange(60) % 12).astype(np.int32), 12, True)
    st.close()
    with pytest.raises(ValueError, match="destroyed"):
        ctx.bcsd_predict(st, X, (np.arange(60) % 12).astype(np.int32))
    ctx.release_cached()  # cached blocks back to the driver; the context keeps working
    st = ctx.bcsd_fit(0, X, y, (np.arange(60) % 12).astype(np.int32), 12, True)
    out, status = ctx.bcsd_predict(st, X, (np.arange(60) % 12).astype(np.int32))
    assert np.isfinite(out).all()


def test_single_rank_communicator_self_gather(ctx):
    """sd_comm_* through the C ABI with one rank: RCCL is loaded with dlopen, the communicator is created from a unique
    id, barrier / max-reduction work, and the gather of a [T, C] field to the root (= this rank) is an in-place copy into
    the [rank][T][C_r] layout.  (world_size 2 is covered on CPU for the partition logic, tests/test_host.py.)"""
    from skdownscale_amd.shard import Communicator

    comm = Communicator.from_env(ctx)  # RANK / WORLD_SIZE unset: one rank
    assert (comm.rank, comm.world) == (0, 1)
    comm.barrier()
    assert comm.allreduce_max(3.5) == 3.5
    rng = np.random.default_rng(4)
    a = rng.standard_normal((50, 7))
    views = comm.gather_field(ctx.to_device(a), [7])
    assert len(views) == 1 and np.array_equal(views[0].to_host(), a)
    views = comm.gather_field(ctx.to_device(a), [7], wait=False)
    comm.wait()
    assert np.array_equal(views[0].to_host(), a)
    with pytest.raises(ValueError, match="contiguous"):
        comm.gather_field(ctx.to_device(a).cells(1, 4), [3])
    comm.close()


def test_sharded_downscaler_on_a_one_rank_rccl_communicator(ctx):
    """The sharded drop-in surface on the RCCL transport with one rank: the BCSD prediction stays on the GPU from upload to
    gather (shard.py: _resident_bcsd + Communicator.gather_field), rank 0 downloads the gathered buffer once; y in another
    spatial dim order than X (aligned by name), a masked cell.  Must equal PointWiseDownscaler."""
    from skdownscale_amd import BcsdTemperature, GridArray, PointWiseDownscaler
    from skdownscale_amd.shard import Communicator, ShardedPointWiseDownscaler

    comm = Communicator.from_env(ctx)
    rng = np.random.default_rng(14)
    index = pd.date_range("1980-01-01", periods=1461)
    Xg = GridArray(15 + 8 * rng.standard_normal((1461, 3, 4)), ("time", "y", "x"), {"time": index})
    yg = GridArray(13 + 9 * rng.standard_normal((1461, 4, 3)), ("time", "x", "y"), {"time": index})
    Xg.values[0, 1, 2] = np.nan
    sharded = ShardedPointWiseDownscaler(BcsdTemperature(), comm=comm)
    sharded.fit(Xg, yg)
    calls = []
    orig = sharded._resident_bcsd
    sharded._resident_bcsd = lambda *a, **k: calls.append(1) or orig(*a, **k)
    from skdownscale_amd.engine import DeviceArray

    downloads = []
    to_host = DeviceArray.to_host
    DeviceArray.to_host = lambda self: downloads.append(self.base is None) or to_host(self)
    try:
        got = sharded.predict(Xg)
    finally:
        DeviceArray.to_host = to_host
    assert downloads == [True], downloads  # one download, of the root buffer itself (not one per rank view)
    plain = PointWiseDownscaler(BcsdTemperature())
    plain.fit(Xg, yg)
    want = plain.predict(Xg)
    assert calls and got.dims == want.dims and got.shape == want.shape
    assert np.array_equal(np.isnan(got.values), np.isnan(want.values)) and np.isnan(got.values[:, 1, 2]).all()
    ok = ~np.isnan(want.values)
    assert np.array_equal(got.values[ok], want.values[ok])
    comm.close()


def test_large_host_copies_take_the_staged_path(ctx):
    """host <-> device copies above 8 MB go through the pinned staging ring (several chunks, ragged tail)"""
    rng = np.random.default_rng(5)
    a = rng.standard_normal((3, 9_000_001))  # 216 MB: four 64 MB chunks, the last one partial
    d = ctx.to_device(a)
    assert np.array_equal(d.to_host(), a)
    d.free()


def test_large_host_predict_matches_resident(ctx):
    """sd_bcsd_fit / sd_bcsd_predict on host buffers of several hundred MB (staged copies, many chunks): same bits as the
    resident path; per-cell status and masked cells land in the right columns."""
    from skdownscale_amd import synth

    T, C = 14_600, 4_700  # 549 MB per field
    index = pd.date_range("1980-01-01", periods=T)
    cells = np.arange(C)
    X, y, Xp = (synth.tas_field(name, 3, index, cells, 10_000) for name in ("X_hist", "y_obs", "X_fut"))
    X[0, 17] = np.nan        # masked cell (core.py:35-37)
    Xp[100, 4_650] = np.inf  # non-finite predict sample
    gid = (np.asarray(index.month) - 1).astype(np.int32)
    st = ctx.bcsd_fit(0, X, y, gid, 12, True)
    out, status = ctx.bcsd_predict(st, Xp, gid)
    dout, dstatus = ctx.bcsd_predict(st, ctx.to_device(Xp), gid)
    assert np.array_equal(status, dstatus) and status[17] == 1 and status[4_650] == 2 and (np.delete(status, [17, 4_650]) == 0).all()
    assert np.array_equal(out, dout.to_host(), equal_nan=True)
    assert np.isnan(out[:, 17]).all() and np.isnan(out[:, 4_650]).all() and np.isfinite(out[:, 4_649]).all()


def _bench_two_ranks(extra_env, extra_args, port):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--cells", "4096", "--times", "1461", "--no-cpu-baseline", "--gather-steps", "1"] + extra_args,
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_line_and_stalled_leg_watchdog():
    """bench.py at N = 2 with both ranks on this box's GPU(s): the throughput line is complete (rccl field: the ranks RCCL itself
    counts, or its refusal of a shared device), and a leg behind the timed loop that never returns -- an RCCL start-up or collective
    that hangs on a node nobody has run it on -- costs the legs, not the line: the watchdog prints it and every rank exits 0."""
    d = _bench_two_ranks({}, ["--leg-timeout", "120"], 29561)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["cells_per_gpu"] == 4096
    assert d["rccl"]["world_size_env"] == 2 and ("ranks" in d["rccl"])
    # (on a box with one GPU RCCL refuses the shared device and the legs finish in seconds; on a box with several, this is a first
    # real run of the collectives -- whatever they do, the line above is what the test is about)
    if "legs_timed_out" in d:
        print("bench.py --gpus 2: legs behind the timed loop stalled:", d["legs_timed_out"])
    s = _bench_two_ranks({"SD_BENCH_FAKE_STALL": "1"}, ["--leg-timeout", "5"], 29562)
    assert s["n_gpus"] == 2 and s["value"] > 0 and s["roofline"]["frac"] > 0
    assert s["legs_timed_out"]["leg"].startswith("fake stall") and s["rccl"]["ranks"] is None
