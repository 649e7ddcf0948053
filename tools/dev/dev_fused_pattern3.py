import sys, numpy as np
sys.path[:0] = ["scikit-downscale_amd", "oracle", "tests"]
import bcsd_oracle as bo
from skdownscale_amd import synth
from skdownscale_amd.engine import default_context
ctx = default_context()
C = 8
index = synth.daily_calendar(14600)
gid = (np.asarray(index.month) - 1).astype(np.int32)
cells = np.arange(C)
X, y, Xp = (synth.tas_field(n, 0, index, cells, C) for n in ("X_hist", "y_obs", "X_fut"))
out, _ = ctx.bcsd_fit_predict(0, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid)
out = out.to_host()
c, g = 0, 0
tt = np.flatnonzero(gid == g)
xg = Xp[tt, c]; yg = y[tt, c]; xh = X[tt, c]
xc = xh.mean(); yc = yg.mean()
roll = bo.rolling_mean_centered(xg); shift = roll - xc; u = xg - shift
ys = np.sort(yg); su = np.sort(u)
r = np.searchsorted(su, u, side="right") - 1
exp = shift + ys[r] - yc
got = out[tt, c]
print("max |exp-oracle path|", np.abs(exp - got).max())
# implied q and implied rank from the GPU output
q_got = got - shift + yc
r_got = np.array([np.argmin(np.abs(ys - v)) for v in q_got])
resid = np.abs(ys[r_got] - q_got)
print("implied-rank residual max", resid.max(), " (small => ranks wrong, large => shift wrong)")
bad = np.flatnonzero(np.abs(exp - got) > 1e-9)
print("n bad", len(bad), "first bad j", bad[:12])
print("rank exp", r[bad[:12]], "rank got", r_got[bad[:12]], "delta", (r_got - r)[bad[:12]])
print("delta stats: min", (r_got-r).min(), "max", (r_got-r).max(), "unique deltas", np.unique(r_got - r)[:20])
