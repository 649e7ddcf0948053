import sys, time
sys.path[:0] = ["/root/repo/scikit-downscale_amd"]
import numpy as np
from skdownscale_amd import synth
from skdownscale_amd.engine import default_context
ctx = default_context()
T, C = 14600, 4096
f = {}
for n, s0 in (("X", 30), ("y", 31), ("Xp", 32)):
    d = ctx.empty((T, C)); ctx.synth_fill(d, synth.GAUSS, 0, s0, c_full=C, amp=3.0); f[n] = d
out = ctx.empty((T, C))
for G in (1, 4):
    gid = (np.arange(T) * G // T).astype(np.int32)
    ctx.bcsd_fit_predict(0, f["X"], f["y"], gid, G, f["Xp"], gid, out=out)
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(2):
        ctx.bcsd_fit_predict(0, f["X"], f["y"], gid, G, f["Xp"], gid, out=out)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 2
    ctx.prof_enable(False)
    print("G", G, "ms/step", dt * 1e3, "cells/s", C / dt, {k: round(v["ms"] / 2, 2) for k, v in ctx.prof().items()})
