import sys, numpy as np
sys.path[:0] = ["scikit-downscale_amd", "oracle", "tests"]
from skdownscale_amd import synth
from skdownscale_amd.engine import default_context
ctx = default_context()
C, T = 8, 14600
index = synth.daily_calendar(T)
gid = (np.asarray(index.month) - 1).astype(np.int32)
cells = np.arange(C)
for kind, name in ((1, "PR"), (0, "TAS")):
    if kind == 1:
        X, y, Xp = (synth.pr_field(n, 0, T, cells, C) for n in ("X_hist", "y_obs", "X_fut"))
    else:
        X, y, Xp = (synth.tas_field(n, 0, index, cells, C) for n in ("X_hist", "y_obs", "X_fut"))
    st = ctx.bcsd_fit(kind, X, y, gid, 12, True)
    ref, _ = ctx.bcsd_predict(st, Xp, gid)
    out, _ = ctx.bcsd_fit_predict(kind, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid)
    d = np.abs(out.to_host() - ref)
    print(name, "fused vs split max diff", d.max(), "bad frac", (d > 1e-9).mean())
    # repeat to check determinism
    out2, _ = ctx.bcsd_fit_predict(kind, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid)
    print(name, "fused run1 vs run2 max diff", np.abs(out2.to_host() - out.to_host()).max())
