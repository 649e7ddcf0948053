#!/usr/bin/env python
"""Randomised sweep of the value-ordered query runs (csrc/sd_analog_runs.h, round 6) -- development library: PureAnalog kinds that read
their analog windows from memory (weight_analogs, mean / weight with a threshold, best with a threshold) and AnalogRegression, random
series lengths (queries >= 2 048 so that the runs apply; ragged last run), cell counts with ragged tiles, k, quantised data (ties:
the exact walk), masked cells and non-finite queries: must equal the time-ordered staging (SD_ANALOG_NORUNS) bit for bit -- and, with
SD_ANALOG_RUNS_ALWAYS, also the three-generation and the fused kernels on sorted queries.   usage: fuzz_runs.py [cases] [seed]"""
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
    stats = {"cases": 0, "regression": 0, "thresh": 0, "quantised": 0, "fused_entry": 0}
    kinds = {"best": _lib.ANALOG_BEST, "weight": _lib.ANALOG_WEIGHT, "mean": _lib.ANALOG_MEAN}
    for it in range(n_cases):
        T = int(rng.choice([rng.integers(600, 3000), rng.integers(3000, 9000), rng.integers(13400, 15000)]))
        Tq = int(rng.choice([rng.integers(2048, 3100), rng.integers(3100, 9000), rng.integers(13400, 15000)]))
        C = int(rng.integers(1, 22))
        k = int(rng.choice([1, 2, 5, 9, 30, 30, 64]))
        k = min(k, T)
        X = rng.standard_normal((T, 1, C))
        y = 2 * X[:, 0, :] + rng.standard_normal((T, C))
        Xq = rng.standard_normal((Tq, 1, C)) * rng.choice([1.0, 1.5])
        quant = rng.random() < 0.3
        if quant:
            X, Xq = np.round(X * 64) / 64, np.round(Xq * 64) / 64
        if rng.random() < 0.3:
            X[0, 0, int(rng.integers(0, C))] = np.nan
        if rng.random() < 0.3:
            Xq[int(rng.integers(0, Tq)), 0, int(rng.integers(0, C))] = np.inf
        regression = rng.random() < 0.3
        kind_name = str(rng.choice(["weight", "weight", "mean", "best"]))
        has_thresh = (not regression) and (kind_name != "weight" or rng.random() < 0.4) and rng.random() < 0.7
        thresh = float(rng.normal()) if has_thresh else None
        fused_entry = (not regression) and rng.random() < 0.3
        res = {}
        for name, env in (("runs", {"SD_ANALOG_RUNS_ALWAYS": "1"}), ("shipped", {}), ("time", {"SD_ANALOG_NORUNS": "1"})):
            os.environ.update(env)
            if regression:
                st = ctx.analog_fit(X, y)
                o, s = ctx.analogreg_predict(st, Xq, max(k, 3))
                st.close()
            elif fused_entry:
                o, s = ctx.analog_fit_predict(X, y, Xq, k, kinds[kind_name], thresh=thresh)
            else:
                st = ctx.analog_fit(X, y)
                o, s = ctx.analog_predict(st, Xq, k, kinds[kind_name], thresh=thresh)
                st.close()
            res[name] = (np.array(o), np.array(s))
            for key in env:
                del os.environ[key]
        for name in ("runs", "shipped"):
            assert np.array_equal(res[name][1], res["time"][1]), (it, name, "status")
            assert np.array_equal(res[name][0], res["time"][0], equal_nan=True), (it, name, T, Tq, C, k, kind_name, thresh, regression, fused_entry)
        stats["cases"] += 1
        stats["regression"] += int(regression)
        stats["thresh"] += int(has_thresh)
        stats["quantised"] += int(quant)
        stats["fused_entry"] += int(fused_entry)
    print(f"fuzz_runs: {stats} in {time.time() - t0:.0f} s, seed {seed}: value-ordered runs bit-identical to the time-ordered staging")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
