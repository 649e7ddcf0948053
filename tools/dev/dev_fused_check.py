"""Development helper: fused BCSD at moderate scale vs the C oracle, error pattern by cell/group."""
import sys
import numpy as np
sys.path[:0] = ["scikit-downscale_amd", "oracle", "tests"]
import c_oracle
from skdownscale_amd import synth
from skdownscale_amd.engine import default_context

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = default_context()
index = synth.daily_calendar(14600)
gid = (np.asarray(index.month) - 1).astype(np.int32)
cells = np.arange(C)
X, y, Xp = (synth.tas_field(n, 0, index, cells, C) for n in ("X_hist", "y_obs", "X_fut"))
exp, _ = c_oracle.bcsd_fit_predict(0, X, y, Xp, gid, gid, nthreads=64)
out, st = ctx.bcsd_fit_predict(0, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid)
out = out.to_host()
err = np.abs(out - exp)
bad = err > 1e-6
print("C", C, "max err", err.max(), "bad elements", bad.sum(), "of", bad.size)
if bad.any():
    t, c = np.nonzero(bad)
    print("bad cells (first 20):", np.unique(c)[:20], "count", len(np.unique(c)))
    print("bad groups:", np.unique(gid[t], return_counts=True))
    print("bad per tile (first 10 tiles):", np.bincount(c // 8)[:10])
    c0 = c[0]
    tt = t[c == c0]
    print("cell", c0, "bad t (first 20)", tt[:20], "n bad", len(tt))
