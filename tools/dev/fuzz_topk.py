#!/usr/bin/env python
"""Randomised sweep of analog_slab_topk_kernel (csrc/sd_analog_topk.h, round 6) -- development library: F = 2 .. 6 features, k = 1 .. 30,
random series lengths and cell counts, PureAnalog kinds and AnalogRegression, with and without a threshold, neighbour outputs; data that
stresses the pieces: quantised coordinates (ties on the sort axis and in the distances), duplicated training points (ties on the k-th
distance: batches are handed back to the heap kernel), a common offset (the centred float32 images), a few far outliers (a coarse
pre-filter: many false positives, same selection), clustered queries (ascending runs of a wave that straddle classes), masked cells and
non-finite queries.  Must equal the heap kernel (SD_ANALOG_HEAP) and the v_readlane form of the pre-filter (SD_TOPK_READLANE) bit for bit,
and the oracle's brute force on the first cells.   usage: fuzz_topk.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scikit-downscale_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import analog_oracle as ao  # noqa: E402
from skdownscale_amd import _lib  # noqa: E402
from skdownscale_amd.engine import Context  # noqa: E402


def main(n_cases, seed):
    ctx = Context(0, lib_path=_lib.DEV_LIB_PATH)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    stats = {"cases": 0, "regression": 0, "thresh": 0, "quantised": 0, "duplicates": 0, "offset": 0, "outliers": 0, "oracle_cells": 0}
    kinds = {"best": _lib.ANALOG_BEST, "weight": _lib.ANALOG_WEIGHT, "mean": _lib.ANALOG_MEAN}
    for it in range(n_cases):
        F = int(rng.integers(2, 7))
        T = int(rng.choice([rng.integers(40, 400), rng.integers(400, 3000), rng.integers(3000, 9000)]))
        Tq = int(rng.choice([rng.integers(1, 200), rng.integers(200, 3000), rng.integers(3000, 7000)]))
        C = int(rng.integers(1, 6))
        k = min(int(rng.choice([1, 2, 5, 9, 16, 29, 30, 30])), T)
        X = rng.standard_normal((T, F, C))
        Xq = rng.standard_normal((Tq, F, C)) * rng.choice([0.5, 1.0, 1.5])
        y = X[:, 0, :] + rng.standard_normal((T, C))
        flavour = str(rng.choice(["plain", "plain", "quantised", "duplicates", "offset", "outliers"]))
        if flavour == "quantised":
            X, Xq = np.round(X * 4) / 4, np.round(Xq * 4) / 4
        elif flavour == "duplicates":
            X[T // 3:] = X[rng.integers(0, max(T // 3, 1), T - T // 3)]
        elif flavour == "offset":
            off = float(rng.choice([1e3, 3e5, -2e7]))
            X, Xq = X + off, Xq + off
        elif flavour == "outliers":
            X[rng.integers(0, T, 3), rng.integers(0, F, 3), 0] = rng.choice([1e4, -1e6, 1e9], 3)
        if rng.random() < 0.2:
            X[0, 0, int(rng.integers(0, C))] = np.nan  # a masked cell
        if rng.random() < 0.3:
            Xq[int(rng.integers(0, Tq)), int(rng.integers(0, F)), int(rng.integers(0, C))] = np.inf
        regression = rng.random() < 0.25 and k >= F + 2
        kind_name = str(rng.choice(["weight", "mean", "mean", "best"]))
        has_thresh = (not regression) and rng.random() < 0.4
        thresh = float(rng.normal()) if has_thresh else None
        neighbours = (not regression) and rng.random() < 0.5
        res = {}
        st = ctx.analog_fit(X, y)
        for name, env in (("topk", {}), ("readlane", {"SD_TOPK_READLANE": "1"}), ("heap", {"SD_ANALOG_HEAP": "1"})):
            os.environ.update(env)
            if regression:
                r = ctx.analogreg_predict(st, Xq, k)
            else:
                r = ctx.analog_predict(st, Xq, k, kinds[kind_name], thresh=thresh, want_neighbors=neighbours)
            res[name] = [np.array(a) for a in r]
            for key in env:
                del os.environ[key]
        st.close()
        # [Tq, C]: the neighbour lists of a non-finite query and of a masked / non-finite cell are not written
        fin = np.isfinite(Xq).all(axis=1) & np.isfinite(X).all(axis=(0, 1))[None, :]
        for name in ("topk", "readlane"):
            for pos, (a, b) in enumerate(zip(res[name], res["heap"])):
                if pos >= 2:
                    a, b = np.where(fin[:, None, :], a, 0), np.where(fin[:, None, :], b, 0)
                assert np.array_equal(a, b, equal_nan=True), (it, name, pos, F, T, Tq, C, k, flavour, kind_name, thresh, regression)
        if neighbours and T * Tq <= 4_000_000:
            status = res["topk"][1]
            for c in range(C):
                if status[c] != 0:
                    continue
                d, i = ao.knn(X[:, :, c], Xq[:, :, c], k)
                assert np.array_equal(res["topk"][2][:, :, c], i), (it, "oracle inds", F, T, Tq, c, k, flavour)
                assert np.array_equal(res["topk"][3][:, :, c], d), (it, "oracle dist", F, T, Tq, c, k, flavour)
                stats["oracle_cells"] += 1
        stats["cases"] += 1
        stats["regression"] += int(regression)
        stats["thresh"] += int(has_thresh)
        stats["quantised"] += int(flavour == "quantised")
        stats["duplicates"] += int(flavour == "duplicates")
        stats["offset"] += int(flavour == "offset")
        stats["outliers"] += int(flavour == "outliers")
    print(f"fuzz_topk: {stats} in {time.time() - t0:.0f} s, seed {seed}: candidate-list kernel (matrix-core and v_readlane pre-filters) "
          f"bit-identical to the heap kernel")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
