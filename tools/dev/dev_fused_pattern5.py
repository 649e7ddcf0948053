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
for (c, g) in ((0, 0), (3, 5)):
    tt = np.flatnonzero(gid == g)
    xg = Xp[tt, c]; yg = y[tt, c]; xh = X[tt, c]
    xc = xh.mean(); yc = yg.mean()
    roll = bo.rolling_mean_centered(xg); shift = roll - xc; u = xg - shift
    ys = np.sort(yg); su = np.sort(u)
    r = np.searchsorted(su, u, side="right") - 1
    exp = shift + ys[r] - yc
    got = out[tt, c]
    bad = np.flatnonzero(np.abs(exp - got) > 1e-9)
    q_got = got - shift + yc
    r_got = np.array([np.argmin(np.abs(ys - v)) for v in q_got])
    resid = np.abs(ys[r_got] - q_got)
    print("cell", c, "group", g, "n bad", len(bad), "bad regs (j%21):", np.unique(bad % 21), "bad lanes count", len(np.unique(bad // 21)))
    print("   implied-rank residual at bad (max)", resid[bad].max() if len(bad) else 0, "rank delta at bad", (r_got - r)[bad][:10])
    # does the wrong value equal expected value of a neighbouring position?
    if len(bad):
        j = bad[0]
        cand = [k for k in range(max(0, j - 25), min(len(exp), j + 25)) if abs(exp[k] - got[j]) < 1e-9]
        print("   first bad j", j, "got matches exp at positions", cand, " shift-diff hypothesis:", got[j] - exp[j])
