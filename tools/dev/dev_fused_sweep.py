import sys, numpy as np
sys.path[:0] = ["scikit-downscale_amd", "oracle", "tests"]
from skdownscale_amd import synth
from skdownscale_amd.engine import default_context
ctx = default_context()
C, T = 5, 14600
index = synth.daily_calendar(T)
gid = (np.asarray(index.month) - 1).astype(np.int32)
cells = np.arange(C)
X = synth.tas_field("X_hist", 0, index, cells, C); y = synth.tas_field("y_obs", 0, index, cells, C)
st = ctx.bcsd_fit(0, X, y, gid, 12, True)
for Tp in (365, 1000, 3000, 7300, 12000, 14000, 14600, 15000, 15700):
    index_p = synth.daily_calendar(Tp)
    gidp = (np.asarray(index_p.month) - 1).astype(np.int32)
    Xp = synth.tas_field("X_fut", 0, index_p, cells, C)
    ref, _ = ctx.bcsd_predict(st, Xp, gidp)
    out, _ = ctx.bcsd_fit_predict(0, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gidp)
    d = np.abs(out.to_host() - ref)
    print("Tp", Tp, "m~", Tp // 12, "max diff", d.max(), "bad frac", (d > 1e-9).mean())
