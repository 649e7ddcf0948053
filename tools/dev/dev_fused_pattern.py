import sys, numpy as np
sys.path[:0] = ["scikit-downscale_amd", "oracle", "tests"]
from skdownscale_amd import synth
from skdownscale_amd.engine import default_context
ctx = default_context()
C = 8
index = synth.daily_calendar(14600)
gid = (np.asarray(index.month) - 1).astype(np.int32)
cells = np.arange(C)
X, y, Xp = (synth.tas_field(n, 0, index, cells, C) for n in ("X_hist", "y_obs", "X_fut"))
st = ctx.bcsd_fit(0, X, y, gid, 12, True)
ref, _ = ctx.bcsd_predict(st, Xp, gid)
out, _ = ctx.bcsd_fit_predict(0, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid)
out = out.to_host()
bad = np.abs(out - ref) > 1e-9
print("bad fraction", bad.mean())
for g in range(2):
    tt = np.flatnonzero(gid == g)
    for c in (0, 5):
        b = bad[tt, c]
        j = np.flatnonzero(b)
        print("group", g, "cell", c, "n", len(tt), "bad", len(j), "lanes with bad:", np.unique(j // 21)[:70], "regs with bad:", np.unique(j % 21))
        d = (out - ref)[tt, c]
        print("   first diffs", d[:8], "q-like?", )
e = st.export()
print("yclimo ref", e["y_climo"][0, :3])
