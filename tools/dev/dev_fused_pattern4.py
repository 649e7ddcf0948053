import sys, numpy as np
sys.path[:0] = ["scikit-downscale_amd", "oracle", "tests"]
import bcsd_oracle as bo
from skdownscale_amd import synth
from skdownscale_amd.engine import default_context
ctx = default_context()
C, T, Tp = 3, 14600, 3000
index = synth.daily_calendar(T); index_p = synth.daily_calendar(Tp)
gid = (np.asarray(index.month) - 1).astype(np.int32); gidp = (np.asarray(index_p.month) - 1).astype(np.int32)
cells = np.arange(C)
X = synth.tas_field("X_hist", 0, index, cells, C); y = synth.tas_field("y_obs", 0, index, cells, C)
Xp = synth.tas_field("X_fut", 0, index_p, cells, C)
st = ctx.bcsd_fit(0, X, y, gid, 12, True)
ref, _ = ctx.bcsd_predict(st, Xp, gidp)
out, _ = ctx.bcsd_fit_predict(0, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gidp)
out = out.to_host()
bad = np.abs(out - ref) > 1e-9
print("bad fraction", bad.mean(), "nan in out", np.isnan(out).sum())
for g in range(12):
    tt = np.flatnonzero(gidp == g)
    b = bad[tt][:, 0]
    print("group", g, "m", len(tt), "bad", b.sum(), "bad j (first 10)", np.flatnonzero(b)[:10], "diffs", (out - ref)[tt, 0][np.flatnonzero(b)[:4]])
