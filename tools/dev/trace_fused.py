"""Phase clocks of analog_f1_fused_kernel (development library, SD_FUSED_TRACE=1): s_memtime ticks per phase for the first cells of
workgroup 0.  usage: SD_DOWNSCALE_LIB=scikit-downscale_amd/lib/libsd_downscale_dev.so SD_FUSED_TRACE=1 python tools/dev/trace_fused.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scikit-downscale_amd"))
from skdownscale_amd import _lib, synth  # noqa: E402
from skdownscale_amd.engine import default_context  # noqa: E402

C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 14600
ctx = default_context()
f = {}
for name, stream, kw in (("X", 20, {}), ("y", 20, dict(amp=2.0, stream2=21, amp2=1.0)), ("Xq", 22, {})):
    f[name] = ctx.synth_fill(ctx.empty((T, C)), synth.GAUSS, 9, stream, c_offset=0, c_full=C, **kw)
X3, Xq3 = ctx.wrap(f["X"].ptr, (T, 1, C)), ctx.wrap(f["Xq"].ptr, (T, 1, C))
out = ctx.empty((T, 3, C))
for i in range(3):
    if i < 2:
        os.environ.pop("SD_FUSED_TRACE", None)
    else:
        os.environ["SD_FUSED_TRACE"] = "1"
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.analog_fit_predict(X3, f["y"], Xq3, 30, _lib.ANALOG_MEAN, out=out)
    ctx.synchronize()
    print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
