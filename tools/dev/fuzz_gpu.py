#!/usr/bin/env python
"""Randomised parity sweep on the GPU (development aid): random sizes / groups / kinds vs the oracles."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scikit-downscale_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import analog_oracle as ao  # noqa: E402
import bcsd_oracle as bo  # noqa: E402
import qm_oracle as qo  # noqa: E402
from _cases import assert_close  # noqa: E402
from skdownscale_amd.engine import default_context  # noqa: E402


def main(n_cases=40, seed=0):
    ctx = default_context()
    rng = np.random.default_rng(seed)
    for it in range(n_cases):
        what = rng.choice(["bcsd", "analog", "qm", "knn", "cunnane", "linreg", "linreg_thresh", "analogreg_thresh", "qm_modes", "nasanex"])
        if what == "bcsd":
            kind = int(rng.integers(0, 2))
            G = int(rng.choice([1, 3, 12, 12, 12]))
            T = int(rng.integers(G * 12, 6000))
            Tp = int(rng.choice([T, rng.integers(G * 3, 7000)]))
            C = int(rng.integers(1, 40))
            gid = rng.integers(0, G, T).astype(np.int32) if rng.random() < 0.3 else (np.arange(T) * G // T).astype(np.int32)
            gid_p = (np.arange(Tp) % G).astype(np.int32)
            gid[:G] = np.arange(G)  # every group is present in fit
            # tie-heavy data on a dyadic grid: sums are exact, so the tie structure does not depend on the order of
            # the climatology / rolling sums (with inexact decimals it does -- in the reference as well)
            q = float(rng.choice([1, 4, 16]))
            dyadic = rng.random() < 0.4
            f = (lambda n: np.round((10 + 3 * rng.standard_normal((n, C))) * q) / q) if dyadic else (lambda n: 10 + 3 * rng.standard_normal((n, C)))
            X, y, Xp = f(T), f(T) + 20, f(Tp)
            # qm_kwargs={'detrend': True}: continuous data only (a removed line turns exact ties into rounding-level near-ties,
            # whose order is unpinned in the reference as well), segments the register-sort kernels serve
            longest = max(np.bincount(gid, minlength=G).max(), np.bincount(gid_p, minlength=G).max())
            detrend = bool(not dyadic and longest <= 2112 and rng.random() < 0.5)
            if detrend:
                X, y, Xp = (a + 1e-3 * rng.standard_normal() * np.arange(len(a))[:, None] for a in (X, y, Xp))
            if kind == 1:
                X, y, Xp = np.abs(X) * (rng.random(X.shape) > 0.4), np.abs(y) + 0.1, np.abs(Xp) * (rng.random(Xp.shape) > 0.4)
            ra = bool(rng.integers(0, 2))
            exp, est = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=G, return_anoms=ra, detrend=detrend)
            st = ctx.bcsd_fit(kind, X, y, gid, G, ra, detrend=detrend)
            out, status = ctx.bcsd_predict(st, Xp, gid_p)
            assert np.array_equal(status, est), (it, status, est)
            assert_close(out, exp, what=f"case {it} bcsd kind={kind} G={G} T={T} Tp={Tp} C={C} detrend={detrend}")
            fused, _ = ctx.bcsd_fit_predict(kind, ctx.to_device(X), ctx.to_device(y), gid, G, ctx.to_device(Xp), gid_p, ra, detrend=detrend)
            assert_close(fused.to_host(), exp, what=f"case {it} fused")
        elif what == "analog":
            F = int(rng.choice([1, 1, 1, 2, 4]))
            T = int(rng.integers(40, 3000)) if rng.random() < 0.8 else int(rng.integers(5000, 16385))
            Tq = int(rng.integers(1, 600)) if rng.random() < 0.8 else int(rng.integers(15000, 20000))
            C = int(rng.integers(1, 6))
            k = int(rng.integers(1, min(T, 64)))
            if F > 1:
                T, Tq = min(T, 3000), min(Tq, 600)  # (the oracle is a Python loop over queries)
            quant = rng.random() < 0.4
            X = rng.standard_normal((T, F, C))
            Xq = rng.standard_normal((Tq, F, C))
            if quant:
                X, Xq = np.round(X, 1), np.round(Xq, 1)
            y = rng.standard_normal((T, C))
            st = ctx.analog_fit(X, y)
            kind = int(rng.choice([0, 2, 3]))
            kk = 1 if kind == 0 else k
            thresh = None if rng.random() < 0.5 else 0.0
            out, _ = ctx.analog_predict(st, Xq, kk, kind, thresh)
            sel = np.arange(Tq) if Tq <= 800 else np.unique(rng.integers(0, Tq, 800))
            exp = ao.pointwise_analog(X, y, Xq[sel], kk, kind, thresh)
            assert_close(out[sel], exp, what=f"case {it} analog F={F} T={T} Tq={Tq} k={kk} kind={kind} thresh={thresh} quant={quant}")
            Xq = Xq[sel[:200]]
            out, _ = ctx.analogreg_predict(st, Xq, k)
            # k <= F + 1 is under-determined: the reference's lstsq cut-off (eps * max(k, F)) sits at the rounding level of
            # the centred analogs, its answer flips between the minimum-norm solution and noise -- unpinned
            if k >= F + 2 and not quant:
                exp = ao.pointwise_analog(X, y, Xq, k, 3, regression=True)
                assert_close(out, exp, what=f"case {it} analogreg F={F} T={T} k={k}")
        elif what == "knn":
            # F > 1 neighbour lists: the feature-0 slab search against brute force, dyadic grids put exact ties everywhere
            F = int(rng.integers(2, 6))
            T = int(rng.integers(64, 4000))
            Tq = int(rng.integers(1, 1500))
            C = int(rng.integers(1, 4))
            k = int(rng.integers(1, min(T, 200)))
            X, Xq = rng.standard_normal((T, F, C)), 1.3 * rng.standard_normal((Tq, F, C))
            if rng.random() < 0.5:
                g = float(rng.choice([2, 4, 16]))
                X, Xq = np.round(X * g) / g, np.round(Xq * g) / g
            y = rng.standard_normal((T, C))
            os.environ["SD_ANALOG_SLAB_CLASSES"] = str(int(rng.choice([1, 2, 8])))
            st = ctx.analog_fit(X, y)
            _, _, inds, dist = ctx.analog_predict(st, Xq, k, 3, want_neighbors=True)
            del os.environ["SD_ANALOG_SLAB_CLASSES"]
            for c in range(C):
                d, i = ao.knn(X[:, :, c], Xq[:, :, c], k)
                assert np.array_equal(inds[:, :, c], i), f"case {it} knn indices F={F} T={T} Tq={Tq} k={k}"
                assert np.array_equal(dist[:, :, c], d), f"case {it} knn distances F={F} T={T} Tq={Tq} k={k}"
        elif what == "cunnane":
            n = int(rng.integers(2, 9000))
            Tp = int(rng.integers(1, 3000))
            C = int(rng.integers(1, 6))
            X = np.round(5 + 2 * rng.standard_normal((n, C)), int(rng.choice([1, 3, 12])))
            Xn = 5 + 2.4 * rng.standard_normal((Tp, C))
            P = rng.uniform(-0.1, 1.1, (Tp, C))
            ex = [None, "min", "max", "both", "1to1"][int(rng.integers(0, 5))]
            ne = int(rng.integers(1, 15))
            st = ctx.qm_fit(X)
            fwd, _ = ctx.qm_cunnane(st, 0, Xn, ex, ne)
            inv, _ = ctx.qm_cunnane(st, 1, P, ex, ne)
            for c in range(C):
                cdf = qo.cunnane_fit(X[:, c])
                exp = qo.cunnane_transform(cdf, Xn[:, c], ex)
                fin = np.isfinite(exp)
                assert np.array_equal(fwd[~fin, c], exp[~fin])
                assert_close(fwd[fin, c], exp[fin], rtol=1e-12, what=f"case {it} cunnane forward n={n} {ex}")
                assert_close(inv[:, c], qo.cunnane_inverse(cdf, P[:, c], ex, ne), rtol=1e-8, what=f"case {it} cunnane inverse n={n} {ex} {ne}")
        elif what == "linreg":
            F = int(rng.integers(1, 9))
            T = int(rng.integers(F + 2, 5000))
            Tq = int(rng.integers(1, 1000))
            C = int(rng.integers(1, 200))
            X = 280 + 10 * rng.standard_normal((T, F, C))
            y = np.einsum("tfc,fc->tc", X, rng.standard_normal((F, C))) + rng.standard_normal((T, C))
            Xq = 280 + 10 * rng.standard_normal((Tq, F, C))
            st = ctx.linreg_fit(X, y)
            out, status = ctx.linreg_predict(st, Xq)
            assert (status == 0).all()
            exp = ao.pointwise_pure_regression(X, y, Xq)
            assert_close(out[:, 0], exp[:, 0], rtol=1e-8, scale=float(np.std(y)), what=f"case {it} linreg F={F} T={T} C={C}")
            assert_close(out[:, 2], exp[:, 2], rtol=1e-8, scale=float(np.std(y)), what=f"case {it} linreg fit error")
        elif what == "nasanex":
            # time_grouper='daily_nasa-nex': 366 overlapping day-of-year groups in fit, day-of-month keys + rolling mean over
            # months in predict (bcsd.py:36-55, 247-267), random calendars, with and without detrended mapping
            import pandas as pd

            from skdownscale_amd.groupers import padded_doy_table

            start = pd.Timestamp("1970-01-01") + pd.Timedelta(days=int(rng.integers(0, 3000)))
            index = pd.date_range(start, periods=int(rng.integers(3 * 366, 9 * 366)))
            pstart = start + pd.Timedelta(days=int(rng.integers(0, 2000)))
            index_p = pd.date_range(pstart, periods=int(rng.integers(40, 6 * 366)))
            C = int(rng.integers(1, 6))
            kind = int(rng.integers(0, 2))
            detrend = bool(rng.random() < 0.4)
            T, Tp = len(index), len(index_p)
            # the residuals of three equally spaced samples about their least-squares line are always (r, -2r, r): a predict group of
            # exactly three samples has a tie between its first and last sample that rounding decides (unpinned in the reference too)
            # (groups of one or two samples are all-zero residuals likewise: DESIGN.md 2 (iii); >= 4 samples per detrended group)
            counts = np.bincount(np.asarray(index_p.day))
            if detrend and (counts[counts > 0] < 4).any():
                detrend = False
            X, y, Xp = (12 + 6 * rng.standard_normal((n, C)) for n in (T, T, Tp))
            if kind == 1:
                X, y, Xp = np.abs(X) * (rng.random(X.shape) > 0.3), np.abs(y) + 0.2, np.abs(Xp) * (rng.random(Xp.shape) > 0.3)
            order, offsets = padded_doy_table(index)
            gq, gt = np.asarray(index_p.day, dtype=np.int32) - 1, np.asarray(index_p.month, dtype=np.int32) - 1
            st = ctx.bcsd_fit_groups(kind, X, y, order, offsets, return_anoms=False, detrend=detrend)
            out, status = ctx.bcsd_predict_trend(st, Xp, gq, gt, 12)
            assert (status == 0).all()
            table = bo.padded_doy_table(index)
            for c in range(C):
                s1, _ = bo.bcsd_fit_cell(kind, X[:, c], y[:, c], None, table=table, return_anoms=False, detrend=detrend)
                if kind == 0:
                    exp, _ = bo.bcsd_predict_trend_cell(s1, Xp[:, c], gq, gt, return_anoms=False)
                else:
                    exp, _ = bo.bcsd_predict_cell(s1, Xp[:, c], gq, return_anoms=False)
                try:
                    assert_close(out[:, c], exp, what=f"case {it} nasanex kind={kind} T={T} Tp={Tp} detrend={detrend} cell {c}")
                except AssertionError:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{it}.npz"), X=X[:, c], y=y[:, c], Xp=Xp[:, c], out=out[:, c], exp=exp,
                             kind=kind, detrend=detrend, start=str(start), pstart=str(pstart), T=T, Tp=Tp)
                    raise
        elif what == "linreg_thresh":
            # PureRegression(thresh): logistic exceedance model + linear model on the exceeding samples (gard.py:416-470)
            F = int(rng.integers(1, 6))
            T = int(rng.integers(60, 3000))
            Tq = int(rng.integers(1, 300))
            C = int(rng.integers(1, 40))
            X = rng.standard_normal((T, F, C)) * rng.choice([1.0, 10.0, 0.01], (1, F, 1))
            y = np.einsum("tfc,fc->tc", X / np.abs(X).max(axis=0, keepdims=True), rng.standard_normal((F, C))) + rng.standard_normal((T, C))
            Xq = X[rng.integers(0, T, Tq)] * 1.1
            thresh = float(np.quantile(y, rng.uniform(0.2, 0.8)))
            st = ctx.linreg_fit(X, y, thresh)
            out, status = ctx.linreg_predict(st, Xq)
            for c in range(C):
                exc = y[:, c] > thresh
                if exc.all() or exc.sum() <= F + 1:
                    continue  # (the threshold is dropped / the subset fit is under-determined: covered by the tests)
                exp = ao.pure_regression_thresh(X[:, :, c], y[:, c], Xq[:, :, c], thresh)[0]
                assert_close(out[:, [0, 2], c], exp[:, [0, 2]], scale=float(np.std(y[:, c])), what=f"case {it} linreg thresh F={F} T={T} cell {c}")
                assert np.abs(out[:, 1, c] - exp[:, 1]).max() < 1e-6, f"case {it} linreg thresh probability F={F} T={T} cell {c}"
        elif what == "analogreg_thresh":
            # AnalogRegression(thresh): per query, logistic model over the analogs + linear model on the exceeding ones
            F = int(rng.integers(1, 4))
            T = int(rng.integers(150, 2500))
            Tq = int(rng.integers(1, 120))
            C = int(rng.integers(1, 4))
            k = int(rng.integers(24, 64))
            X, Xq = rng.standard_normal((T, F, C)), rng.standard_normal((Tq, F, C))
            y = 0.3 * X.sum(axis=1) + rng.standard_normal((T, C))
            thresh = float(np.median(y))
            st = ctx.analog_fit(X, y)
            out, status = ctx.analogreg_predict(st, Xq, k, thresh)
            try:
                exp = ao.pointwise_analog(X, y, Xq, k, None, thresh=thresh, regression=True)
            except ValueError:
                print(f"case {it}: analogreg_thresh one-class query (skipped)", flush=True)
                continue
            # a query with <= F + 1 exceeding analogs is under-determined: the reference's lstsq cut-off (eps * max(n, F) * s_max)
            # sits at the rounding level of the centred rows, its answer flips between the minimum-norm solution (what the
            # engine returns) and noise (seed 131: 2 exceeding analogs, second singular value 7.1e-17 vs cut-off 3.1e-17) -- unpinned
            pinned = np.ones((Tq, C), dtype=bool)
            for c in range(C):
                _, ii = ao.knn(X[:, :, c], Xq[:, :, c], k)
                pinned[:, c] = (y[ii, c] > thresh).sum(axis=1) >= F + 2
            got, want = out[:, [0, 2]], exp[:, [0, 2]]
            sel = np.broadcast_to(pinned[:, None, :], got.shape)
            assert np.array_equal(np.isnan(got), np.isnan(want)), f"case {it} analogreg thresh NaN pattern"
            assert_close(got[sel], want[sel], scale=float(np.nanstd(want)), what=f"case {it} analogreg thresh F={F} T={T} k={k}")
            assert np.abs(out[:, 1] - exp[:, 1]).max() < 1e-6, f"case {it} analogreg thresh probability F={F} T={T} k={k}"
        elif what == "qm_modes":
            # QuantileMappingReressor / EquidistantCdfMatcher with synthetic end points, samples inside the fitted range
            T = int(rng.integers(25, 6000))
            Tp = int(rng.integers(1, 4000))
            C = int(rng.integers(1, 10))
            ne = int(rng.integers(2, 12))
            X, y = 5 + 2 * rng.standard_normal((T, C)), 15 + 3 * rng.standard_normal((T, C))
            lo, hi = X.min(axis=0), X.max(axis=0)
            Xp = lo + (hi - lo) * rng.random((Tp, C))
            st = ctx.qm_fit(X, y)
            for ex in ("min", "max", "both"):
                out, _ = ctx.qm_predict(st, 0, Xp, ex, ne)
                assert_close(out, qo.pointwise_qm("qmr", X, y, Xp, ex, ne), what=f"case {it} qmr T={T} Tp={Tp} {ex} n_endpoints={ne}")
                for kind, code in (("difference", 1), ("ratio", 2)):
                    out, _ = ctx.qm_predict(st, code, Xp, ex, ne)
                    exp = qo.pointwise_qm("ecm", X, y, Xp, ex, ne, kind=kind)
                    fin = np.isfinite(exp) & np.isfinite(out)
                    assert_close(out[fin], exp[fin], what=f"case {it} ecm {kind} T={T} Tp={Tp} {ex} n_endpoints={ne}")
        else:
            T = int(rng.integers(21, 9000))
            Tp = int(rng.integers(1, 9000))
            C = int(rng.integers(1, 12))
            q = rng.choice([1, 2, 8])
            X, y, Xp = (np.round(5 + 2 * rng.standard_normal((n, C)), q) for n in (T, T, Tp))
            y = y + 10
            st = ctx.qm_fit(X, y)
            for ex in (None, "1to1"):
                out, _ = ctx.qm_predict(st, 0, Xp, ex == "1to1")
                assert_close(out, qo.pointwise_qm("qmr", X, y, Xp, ex), what=f"case {it} qmr T={T} Tp={Tp} {ex}")
                out, _ = ctx.qm_predict(st, 1, Xp, ex == "1to1")
                assert_close(out, qo.pointwise_qm("ecm", X, y, Xp, ex, kind="difference"), what=f"case {it} ecm T={T} Tp={Tp} {ex}")
        print(f"case {it}: {what} ok", flush=True)
    print("fuzz ok")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
