#!/bin/bash
# round 6: thresholded PureAnalog kinds (k = 30, thresh = 0) at 16 384 cells -- development script (bench.py has no threshold option: a direct call)
export SD_DOWNSCALE_LIB=${SD_DOWNSCALE_LIB:-$PWD/scikit-downscale_amd/lib/libsd_downscale.so}
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), "scikit-downscale_amd"))
from skdownscale_amd import _lib, synth
from skdownscale_amd.engine import default_context
C, T = 16384, 14600
ctx = default_context()
f = {}
for name, stream, kw in (("X", 20, {}), ("y", 20, dict(amp=2.0, stream2=21, amp2=1.0)), ("Xq", 22, {})):
    f[name] = ctx.synth_fill(ctx.empty((T, C)), synth.GAUSS, 9, stream, c_offset=0, c_full=C, **kw)
X3, Xq3 = ctx.wrap(f["X"].ptr, (T, 1, C)), ctx.wrap(f["Xq"].ptr, (T, 1, C))
out = ctx.empty((T, 3, C))
st = ctx.analog_fit(X3, f["y"])
for kind, name in ((_lib.ANALOG_WEIGHT, "weight_analogs"), (_lib.ANALOG_MEAN, "mean_analogs")):
    for thresh in (None, 0.0):
        ts = []
        for i in range(4):
            ctx.synchronize(); t0 = time.perf_counter()
            ctx.analog_predict(st, Xq3, 30, kind, thresh, out=out)
            ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{name} thresh={thresh}: predict {min(ts[1:]):.2f} ms per {C} cells", flush=True)
PY
