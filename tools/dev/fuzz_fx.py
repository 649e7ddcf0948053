#!/usr/bin/env python
"""Randomised parity sweep of the fused BCSD kernels (csrc/sd_bcsd_fx.hip) against the NumPy oracle: every kernel width
(segments of 1 .. 1 536 samples), equal / longer / shorter predict series, partly filled lanes, exact ties and near-ties,
zero-inflated series, masked and non-finite cells, fit + predict in one call and predict from a fitted state."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scikit-downscale_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import bcsd_oracle as bo  # noqa: E402
from _cases import assert_close  # noqa: E402
from skdownscale_amd.engine import default_context  # noqa: E402


def main(n_cases, seed):
    ctx = default_context()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    stats = {"cases": 0, "tas": 0, "pr": 0, "ties": 0, "state": 0}
    for it in range(n_cases):
        kind = int(rng.integers(0, 2))
        G = int(rng.choice([1, 2, 5, 12, 12]))
        seg = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1536), 1240, 1280, 1200]))
        T = min(G * seg, 18000)
        mode = rng.choice(["same", "longer", "shorter"])
        Tp = T if mode == "same" else int(T * rng.uniform(1.05, 1.3)) if mode == "longer" else max(G, int(T * rng.uniform(0.3, 0.95)))
        if Tp > G * 1536:
            Tp = G * 1536
        C = int(rng.integers(1, 20))
        gid = (np.arange(T) % G).astype(np.int32) if rng.random() < 0.5 else (np.arange(T) * G // T).astype(np.int32)
        gid_p = (np.arange(Tp) % G).astype(np.int32) if rng.random() < 0.5 else (np.arange(Tp) * G // Tp).astype(np.int32)
        style = rng.choice(["cont", "cont", "dyadic", "near", "const_stretch"])
        if kind == 0:
            f = lambda n: 12 + 6 * rng.standard_normal((n, C))  # noqa: E731
            X, y, Xp = f(T), f(T) + 3, f(Tp)
            if style == "dyadic":
                X, y, Xp = (np.round(a * 4) / 4 for a in (X, y, Xp))
            elif style == "near":  # observations a few ulps apart: equal 21-bit keys, distinct doubles (long equal-q runs in the
                # fix-up).  Not on the x side: predict samples that tie in exact arithmetic end up an ulp apart after the rolling
                # mean, in an order that depends on how x_climo was summed -- unpinned in the reference as well (pandas' online
                # rolling mean rounds differently again)
                y = np.round(y, 2) + rng.integers(0, 4, y.shape) * 2.0 ** -44
            elif style == "const_stretch" and Tp > 40:
                # a constant stretch of x_fut = exact ties after the shift -- on a dyadic grid, where every sum is exact: with
                # inexact decimals the samples at a segment's end (clipped windows) come out an ulp away from the others or not,
                # depending on the rounding of x_climo (unpinned in the reference as well)
                X, Xp = np.round(X * 4) / 4, np.round(Xp * 4) / 4
                t0c = int(rng.integers(0, Tp - 30))
                Xp[t0c:t0c + 25, 0] = Xp[t0c, 0]
        else:
            f = lambda n: rng.gamma(0.7, 4.0, (n, C)) * (rng.random((n, C)) > rng.uniform(0.2, 0.8))  # noqa: E731
            X, y, Xp = f(T), f(T) + (0.01 if rng.random() < 0.5 else 0.0), f(Tp)
            if style == "dyadic":
                X, y, Xp = (np.round(a * 8) / 8 for a in (X, y, Xp))
            if (y.sum(axis=0) == 0).any():
                y[0] += 0.5
        if rng.random() < 0.2 and C > 2:
            X[0, 1] = np.nan  # a masked cell
        if rng.random() < 0.15 and C > 3:
            Xp[int(rng.integers(0, Tp)), 2] = np.inf
        ra = bool(rng.random() < 0.7)
        exp, est = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=G, return_anoms=ra)
        dX, dy, dXp = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xp)
        out, st = ctx.bcsd_fit_predict(kind, dX, dy, gid, G, dXp, gid_p, ra)
        what = f"case {it}: kind={kind} G={G} T={T} Tp={Tp} C={C} style={style} anoms={ra}"
        assert np.array_equal(st, est), (what, st, est)
        try:
            assert_close(out.to_host(), exp, what="fused " + what)
        except AssertionError as e:
            got = out.to_host()
            bad = np.argwhere(~np.isclose(got, exp, rtol=1e-6, atol=1e-6 * np.nanstd(exp), equal_nan=True))
            print(e)
            for t, c in bad[:5]:
                g = gid_p[t]
                seg = np.flatnonzero(gid_p == g)
                pos = int(np.searchsorted(seg, t))
                print(f"  t={t} cell={c} group={g} position {pos} of {len(seg)} got={got[t, c]!r} exp={exp[t, c]!r}")
                ys = np.sort(y[gid == g, c])
                j = np.searchsorted(ys, got[t, c] - (0 if not ra else 0))
                print("   fit segment length", (gid == g).sum(), "neighbouring sorted y:", ys[max(0, j - 2):j + 3])
            np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_fx_fail.npz"), X=X, y=y, Xp=Xp, gid=gid, gid_p=gid_p, kind=kind, G=G, ra=ra, got=got, exp=exp)
            raise
        if rng.random() < 0.5:
            state = ctx.bcsd_fit(kind, dX, dy, gid, G, ra)
            out2, st2 = ctx.bcsd_predict(state, dXp, gid_p)
            assert np.array_equal(st2, est), (what, st2, est)
            assert_close(out2.to_host(), exp, what="state " + what)
            if rng.random() < 0.5:
                # the same state with other tails (qm_kwargs={'qt_kwargs': ...}: sd_bcsd_state_set_tails)
                ex = [None, "min", "max", "both", "1to1"][int(rng.integers(5))]
                ne = int(rng.choice([1, 2, 3, 5, 10, 25, 60]))
                state.set_tails(ex, ne)
                exp3, est3 = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=G, return_anoms=ra, extrapolate=ex, n_endpoints=ne)
                out3, st3 = ctx.bcsd_predict(state, dXp, gid_p)
                assert np.array_equal(st3, est3), (what, ex, ne)
                assert_close(out3.to_host(), exp3, what=f"state, extrapolate={ex} n_endpoints={ne} " + what)
                stats["tails"] = stats.get("tails", 0) + 1
            state.close()
            stats["state"] += 1
        stats["cases"] += 1
        stats["tas" if kind == 0 else "pr"] += 1
        stats["ties"] += style in ("dyadic", "near", "const_stretch")
    print(f"fuzz_fx: {stats} all ok, seed {seed}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
