"""Segments of 1 281 .. 1 536 samples (K = 24 fused kernels: 20 - 36 spilled registers) against RANK + APPLY (K = 33) on the same data:
development library, SD_BCSD_FUSED=0 forces the pair.  T = 17 520 daily steps -> months of 1 344 .. 1 488 samples."""
import os
import sys
import time

sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scikit-downscale_amd")]
import numpy as np  # noqa: E402
from skdownscale_amd import synth  # noqa: E402
from skdownscale_amd.engine import default_context  # noqa: E402

ctx = default_context()
T, C = 17520, 50000
index = synth.daily_calendar(T)
gid = (np.asarray(index.month) - 1).astype(np.int32)
print("month lengths", np.bincount(gid))
for kind, name in ((0, "BcsdTemperature"), (1, "BcsdPrecipitation")):
    f = {}
    for n, s0 in (("X", 30), ("y", 31), ("Xp", 32)):
        d = ctx.empty((T, C))
        if kind == 0:
            ctx.synth_fill(d, synth.GAUSS, 0, s0, c_full=C, amp=3.0)
        else:
            ctx.synth_fill(d, synth.PRECIP, 0, s0, c_full=C, amp=40.0, p_dry=0.5)
        f[n] = d
    out = ctx.empty((T, C))
    ctx.bcsd_fit_predict(kind, f["X"], f["y"], gid, 12, f["Xp"], gid, out=out)
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.bcsd_fit_predict(kind, f["X"], f["y"], gid, 12, f["Xp"], gid, out=out)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ctx.prof_enable(False)
    print(name, "fused=" + os.environ.get("SD_BCSD_FUSED", "1"), "ms/step %.2f" % (dt * 1e3), {k: round(v["ms"] / 3, 2) for k, v in ctx.prof().items() if v["ms"] > 0.3})
    for d in list(f.values()) + [out]:
        d.free()
