#!/usr/bin/env python
"""Randomised sweep over the code paths round 3 added (development aid, run on the GPU box):

  tile      F = 1 analog fit of a 13 313 .. 16 384 sample series: tile-shaped first stage vs the two-transpose + full-sort
            path of the development library (SD_ANALOG_NOTILE), bit for bit, random widths / masked / non-finite / tied cells,
            and against the oracle's brute force for a few cells
  long      BCSD with qm_kwargs={'detrend': True} on segments of 2 113 .. 19 456 samples vs the oracle
  trend     TrendAwareQuantileMappingRegressor vs the oracle's restatement
  f32       float32 host grids: widened / narrowed on the device == the float64 call on the widened inputs

usage: fuzz_r3.py [seconds] [first seed] [case]   (loops over seeds until the time is used; exit code 1 on the first mismatch)
2 197 cases (seeds 1000..3196) passed on an MI355X at the end of round 3; tests/test_gpu_fuzz.py runs 80 more.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scikit-downscale_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import analog_oracle as ao  # noqa: E402
import bcsd_oracle as bo  # noqa: E402
import qm_oracle as qo  # noqa: E402
from _cases import assert_close  # noqa: E402
from skdownscale_amd import _lib  # noqa: E402
from skdownscale_amd.engine import Context  # noqa: E402


def same(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def case_tile(rng, ctx, dev):
    T = int(rng.integers(13313, 16385))
    C = int(rng.integers(1, 41))
    Tq = int(rng.integers(1, 900))
    k = int(rng.integers(1, 64))
    X = rng.standard_normal((T, 1, C)) * float(rng.choice([1.0, 1e-3, 1e4]))
    y = 2.0 * X[:, 0, :] + rng.standard_normal((T, C))
    Xq = 1.1 * rng.standard_normal((Tq, 1, C)) * (np.nanstd(X) if np.isfinite(np.nanstd(X)) else 1.0)
    notes = []
    for c in range(C):
        r = rng.random()
        if r < 0.08:
            X[0, 0, c] = np.nan
            notes.append((c, "masked"))
        elif r < 0.14:
            X[int(rng.integers(1, T)), 0, c] = rng.choice([np.inf, -np.inf, np.nan])
            notes.append((c, "nonfinite"))
        elif r < 0.30:
            g = float(rng.choice([2, 8, 64])) / (np.std(X[:, 0, c]) + 1e-300)
            X[:, 0, c] = np.round(X[:, 0, c] * g) / g
            notes.append((c, "ties"))
        elif r < 0.34:
            X[:, 0, c] = float(rng.standard_normal())
            notes.append((c, "constant"))
        elif r < 0.38:
            X[:, 0, c] *= 1e150
            notes.append((c, "huge"))
        elif r < 0.42:  # magnitudes in the pad range of the tagged keys: the exact kernel
            X[rng.integers(1, T, 3), 0, c] = np.array([1.7e308, -1.7e308, 1.79e308])
            notes.append((c, "padrange"))
    os.environ.pop("SD_ANALOG_NOTILE", None)
    st = ctx.analog_fit(X, y)
    os.environ["SD_ANALOG_NOTILE"] = "1"
    st0 = dev.analog_fit(X, y)
    del os.environ["SD_ANALOG_NOTILE"]
    what = f"tile T={T} C={C} Tq={Tq} k={k} {notes}"
    for kind in (0, 2, 3):
        a, sa = ctx.analog_predict(st, Xq, 1 if kind == 0 else k, kind)
        b, sb = dev.analog_predict(st0, Xq, 1 if kind == 0 else k, kind)
        assert sa.tolist() == sb.tolist(), (what, sa, sb)
        assert same(a, b), (what, kind)
    a, sa, ia, da = ctx.analog_predict(st, Xq, k, 3, want_neighbors=True)
    b, sb, ib, db = dev.analog_predict(st0, Xq, k, 3, want_neighbors=True)
    live = np.flatnonzero(np.asarray(sa) == 0)
    assert np.array_equal(ia[:, :, live], ib[:, :, live]) and np.array_equal(da[:, :, live], db[:, :, live]), what
    tied = {c for c, n in notes if n in ("ties", "constant", "padrange")}
    for c in [c for c in live if c not in tied][:2]:  # (tie order of the KDTree is unpinned: SURVEY 8c)
        sel = np.unique(rng.integers(0, Tq, min(Tq, 150)))
        d, i = ao.knn(X[:, :, c], Xq[sel][:, :, c], k)
        assert np.array_equal(ia[sel][:, :, c], i) and np.array_equal(da[sel][:, :, c], d), (what, c)
    r, sr = ctx.analogreg_predict(st, Xq[:100], max(k, 3))
    r0, sr0 = dev.analogreg_predict(st0, Xq[:100], max(k, 3))
    assert same(r, r0), (what, "analogreg")
    return what


def case_long(rng, ctx, dev):
    kind = int(rng.integers(0, 2))
    G = int(rng.choice([1, 1, 2, 4]))
    T = int(rng.integers(2113 * G, min(19456 * G, 40000) + 1))
    Tp = int(rng.choice([T, rng.integers(max(2113 * G, T // 2), min(19456 * G, 40000) + 1)]))
    C = int(rng.integers(1, 7))
    gid = (np.arange(T) * G // T).astype(np.int32)
    gid_p = (np.arange(Tp) * G // Tp).astype(np.int32)
    if max(np.bincount(gid).max(), np.bincount(gid_p).max()) > 19456:
        return None
    f = lambda n: 10 + 3 * rng.standard_normal((n, C))  # noqa: E731
    X, y, Xp = f(T), f(T) + 20, f(Tp)
    detrend = bool(rng.random() < 0.8)
    if detrend:
        X, y, Xp = (a + 1e-3 * rng.standard_normal() * np.arange(len(a))[:, None] for a in (X, y, Xp))
    if kind == 1:
        X, y, Xp = np.abs(X), np.abs(y) + 0.1, np.abs(Xp)
    ra = bool(rng.integers(0, 2))
    what = f"long kind={kind} G={G} T={T} Tp={Tp} C={C} detrend={detrend} ra={ra}"
    exp, est = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=G, return_anoms=ra, detrend=detrend)
    st = ctx.bcsd_fit(kind, X, y, gid, G, ra, detrend=detrend)
    out, status = ctx.bcsd_predict(st, Xp, gid_p)
    assert np.array_equal(status, est), (what, status, est)
    assert_close(out, exp, what=what)
    fused, _ = ctx.bcsd_fit_predict(kind, ctx.to_device(X), ctx.to_device(y), gid, G, ctx.to_device(Xp), gid_p, ra, detrend=detrend)
    assert_close(fused.to_host(), exp, what=what + " fused")
    return what


def case_trend(rng, ctx, dev):
    from skdownscale_amd import EquidistantCdfMatcher, QuantileMappingReressor, TrendAwareQuantileMappingRegressor

    T = int(rng.integers(60, 6000))
    Tp = int(rng.integers(30, 6000))
    name = str(rng.choice(["qmr", "ecm"]))
    t, tp = np.arange(T, dtype=np.float64), np.arange(Tp, dtype=np.float64)
    X = 10 + 2 * rng.standard_normal(T) + float(rng.normal(0, 2e-3)) * t
    y = 12 + 3 * rng.standard_normal(T) + float(rng.normal(0, 2e-3)) * t
    Xp = 11 + 2.2 * rng.standard_normal(Tp) + float(rng.normal(0, 2e-3)) * tp
    make = (lambda: QuantileMappingReressor()) if name == "qmr" else (lambda: EquidistantCdfMatcher())
    what = f"trend {name} T={T} Tp={Tp}"
    m = TrendAwareQuantileMappingRegressor(make()).fit(X[:, None], y[:, None])
    out = np.asarray(m.predict(Xp[:, None]))[:, 0]
    exp = np.asarray(qo.trend_aware_predict(name, X, y, Xp)).ravel()
    assert_close(out, exp, scale=float(np.std(exp)), what=what)
    return what


def case_f32(rng, ctx, dev):
    kind = int(rng.integers(0, 2))
    T = int(rng.integers(400, 5000))
    Tp = int(rng.integers(200, 5000))
    C = int(rng.integers(1, 50))
    gid = (np.arange(T) % 12).astype(np.int32)
    gid_p = (np.arange(Tp) % 12).astype(np.int32)
    f = lambda n: (10 + 3 * rng.standard_normal((n, C))).astype(np.float32)  # noqa: E731
    X, y, Xp = f(T), f(T) + np.float32(20), f(Tp)
    if kind == 1:
        X, y, Xp = np.abs(X) * (rng.random(X.shape) > 0.4), np.abs(y) + np.float32(0.1), np.abs(Xp) * (rng.random(Xp.shape) > 0.4)
        X, y, Xp = X.astype(np.float32), y.astype(np.float32), Xp.astype(np.float32)
    what = f"f32 kind={kind} T={T} Tp={Tp} C={C}"
    st = ctx.bcsd_fit(kind, X, y, gid, 12, True)
    out, s = ctx.bcsd_predict(st, Xp, gid_p, out_dtype=np.float32)
    st64 = ctx.bcsd_fit(kind, X.astype(np.float64), y.astype(np.float64), gid, 12, True)
    out64, s64 = ctx.bcsd_predict(st64, Xp.astype(np.float64), gid_p)
    assert out.dtype == np.float32 and np.array_equal(s, s64), (what, out.dtype, s, s64)
    assert same(out, out64.astype(np.float32)), (what, float(np.nanmax(np.abs(out - out64))))
    r64, _ = ctx.bcsd_predict(st64, Xp.astype(np.float64), gid_p, out_dtype=np.float32)  # float64 in, float32 out
    assert r64.dtype == np.float32 and same(r64, out), what + " (float64 in)"
    return what


CASES = {"tile": case_tile, "long": case_long, "trend": case_trend, "f32": case_f32}


def main(seconds=120.0, seed0=1000, only=None, max_cases=None):
    ctx = Context(0)
    dev = Context(0, lib_path=_lib.DEV_LIB_PATH)  # the SD_* switches exist only in the development library
    t0 = time.time()
    n, seed = {k: 0 for k in CASES}, seed0
    while time.time() - t0 < seconds and (max_cases is None or sum(n.values()) < max_cases):
        rng = np.random.default_rng(seed)
        name = only or str(rng.choice(["tile", "tile", "long", "trend", "f32"]))
        try:
            what = CASES[name](rng, ctx, dev)
        except AssertionError as e:
            print(f"FAILED seed={seed} {name}: {str(e)[:2000]}", flush=True)
            return 1
        if what is not None:
            n[name] += 1
            print(f"ok seed={seed} {what[:160]}", flush=True)
        seed += 1
    print(f"fuzz_r3: {n} cases in {time.time() - t0:.0f} s, seeds {seed0}..{seed - 1}: all ok", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
                  sys.argv[3] if len(sys.argv) > 3 else None))
