#!/usr/bin/env python
"""Secondary measurements (BASELINE configs 3 and 4): BcsdPrecipitation and PureAnalog on one MI355X.

Not the headline bench (bench.py); prints one JSON line per workload with the same roofline convention:
algorithmic bytes per cell / kernel time from HIP events on the engine's stream.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-downscale_amd"))
from skdownscale_amd import _lib, synth  # noqa: E402
from skdownscale_amd.engine import Context  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["bcsd_pr", "analog", "analogreg", "qmr", "ecm", "pure_regression"], default="analog")
    ap.add_argument("--cells", type=int, default=8192)
    ap.add_argument("--times", type=int, default=14600)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--kind", default="mean_analogs")
    ap.add_argument("--features", type=int, default=1)
    ap.add_argument("--out", default=None, help="append the JSON line to this file")
    args = ap.parse_args()
    ctx = Context(0)
    T, C = args.times, args.cells
    index = synth.daily_calendar(T)
    gid = (np.asarray(index.month) - 1).astype(np.int32)

    def field(kind, stream, **kw):
        d = ctx.empty((T, C))
        ctx.synth_fill(d, kind, 0, stream, c_full=C, **kw)
        return d

    if args.workload == "bcsd_pr":
        f = {n: field(synth.PRECIP, synth.PR_FIELDS[n]["stream"], amp=synth.PR_FIELDS[n]["amp"], p_dry=synth.PR_FIELDS[n]["p_dry"])
             for n in ("X_hist", "y_obs", "X_fut")}
        out = ctx.empty((T, C))
        step = lambda: ctx.bcsd_fit_predict(_lib.BCSD_PR, f["X_hist"], f["y_obs"], gid, 12, f["X_fut"], gid, True, out=out)  # noqa: E731
        bytes_per_cell = 8 * (T + 2 * T)  # y_obs, X_fut, out (X_hist is only validated: + 8*T actually read)
        name = f"BcsdPrecipitation zero-inflated, {C} cells x {T} steps"
    elif args.workload in ("qmr", "ecm"):
        f = {n: field(synth.GAUSS, s0, amp=a) for n, s0, a in (("X", 30, 3.0), ("y", 31, 4.0), ("Xp", 32, 3.5))}
        out = ctx.empty((T, C))
        code = 0 if args.workload == "qmr" else 1

        def step():
            st = ctx.qm_fit(f["X"], f["y"])
            r = ctx.qm_predict(st, code, f["Xp"], out=out)
            st.close()
            return r
        bytes_per_cell = 8 * (2 * T + 2 * T)  # X, y, Xp read, out written
        name = f"{'QuantileMappingReressor' if code == 0 else 'EquidistantCdfMatcher difference'} (whole series), {C} cells x {T} steps"
    else:
        F = args.features
        y = field(synth.GAUSS, 20, amp=2.0, stream2=21, amp2=1.0)
        X3, Xq3 = ctx.empty((T, F, C)), ctx.empty((T, F, C))
        for name, arr, s0 in (("X", X3, 20), ("Xq", Xq3, 22)):  # [T, F, C]: feature f of time t is row t*F + f
            rows = ctx.wrap(arr.ptr, (T * F, C))
            ctx.synth_fill(rows, synth.GAUSS, 0, s0, c_full=C)
        out = ctx.empty((T, 3, C))
        kinds = {"best_analog": 0, "sample_analogs": 1, "weight_analogs": 2, "mean_analogs": 3}

        def step():
            if args.workload == "pure_regression":
                st = ctx.linreg_fit(X3, y)
                r = ctx.linreg_predict(st, Xq3, out=out)
                st.close()
                return r
            st = ctx.analog_fit(X3, y)
            if args.workload == "analog":
                k_eff = 1 if args.kind == "best_analog" else args.k  # gard.py:291-296: best_analog queries one neighbour
                r = ctx.analog_predict(st, Xq3, k_eff, kinds[args.kind], out=out)
            else:
                r = ctx.analogreg_predict(st, Xq3, args.k, out=out)
            st.close()
            return r
        bytes_per_cell = 8 * (F * T + T + F * T + 3 * T)
        label = {"analog": "PureAnalog " + args.kind + f" k={args.k}", "analogreg": f"AnalogRegression k={args.k}",
                 "pure_regression": "PureRegression"}[args.workload]
        name = f"{label} F={args.features}, {C} cells x {T} steps"
    step()
    ctx.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ctx.prof_enable(False)
    prof = {k: v["ms"] / args.steps for k, v in ctx.prof().items()}
    kms = sum(prof.values())
    achieved = C * bytes_per_cell / (kms * 1e-3) / 1e9
    line = json.dumps({"workload": name, "cells_per_s": C / dt, "ms_per_step": dt * 1e3, "kernel_ms_per_step": prof,
                      "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                                   "algorithmic_bytes_per_cell": bytes_per_cell}})
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
