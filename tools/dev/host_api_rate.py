"""PCIe-inclusive rate of the host-buffer API (sd_bcsd_fit + sd_bcsd_predict on NumPy arrays): DESIGN.md section 6."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "scikit-downscale_amd"))
from skdownscale_amd import _lib  # noqa: E402
from skdownscale_amd.engine import default_context  # noqa: E402

C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 14600
rng = np.random.default_rng(0)
X, y, Xp = (280 + 10 * rng.standard_normal((T, C)) for _ in range(3))
gid = (np.arange(T) // 30 % 12).astype(np.int32)
ctx = default_context()
for rep in range(3):
    t0 = time.perf_counter()
    st = ctx.bcsd_fit(_lib.BCSD_TAS, X, y, gid, 12, True)
    out, status = ctx.bcsd_predict(st, Xp, gid)
    st.close()
    dt = time.perf_counter() - t0
print(json.dumps({"workload": f"BcsdTemperature host-buffer API (PCIe inclusive), {C} cells x {T} steps", "cells_per_s": C / dt,
                  "seconds": dt, "host_to_device_GBps": 3 * X.nbytes / dt / 1e9}))
