"""Where the time of the host-buffer API goes (PCIe inclusive): copies alone, fit, predict.  GPU box only."""
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scikit-downscale_amd"))
from skdownscale_amd import synth  # noqa: E402
from skdownscale_amd.engine import default_context  # noqa: E402

ctx = default_context()
T, C = 14_600, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
index = pd.date_range("1980-01-01", periods=T)
cells = np.arange(C)
X, y, Xp = (synth.tas_field(name, 3, index, cells, 100_000) for name in ("X_hist", "y_obs", "X_fut"))
gid = (np.asarray(index.month) - 1).astype(np.int32)
gb = X.nbytes / 1e9


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
        del r
    return min(ts)


d = ctx.to_device(X)
t = best(lambda: d.copy_from_host(X))
print(f"H2D of one field ({gb:.2f} GB): {t * 1e3:.1f} ms = {gb / t:.1f} GB/s")
t = best(lambda: d.to_host())
print(f"D2H into a fresh array: {t * 1e3:.1f} ms = {gb / t:.1f} GB/s")
st = ctx.bcsd_fit(0, X, y, gid, 12, True)
t_fit = best(lambda: ctx.bcsd_fit(0, X, y, gid, 12, True))
print(f"bcsd_fit (2 fields in): {t_fit * 1e3:.1f} ms = {2 * gb / t_fit:.1f} GB/s")
t_pred = best(lambda: ctx.bcsd_predict(st, Xp, gid))
print(f"bcsd_predict (1 in, 1 out): {t_pred * 1e3:.1f} ms = {2 * gb / t_pred:.1f} GB/s")
dX, dy, dXp = ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xp)
t = best(lambda: ctx.bcsd_fit(0, dX, dy, gid, 12, True))
print(f"bcsd_fit resident: {t * 1e3:.2f} ms")
st2 = ctx.bcsd_fit(0, dX, dy, gid, 12, True)
t = best(lambda: ctx.bcsd_predict(st2, dXp, gid))
print(f"bcsd_predict resident: {t * 1e3:.2f} ms")
print(f"fit + predict on host arrays: {(t_fit + t_pred) * 1e3:.1f} ms = {C / (t_fit + t_pred):.0f} cells/s")
