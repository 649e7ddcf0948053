"""bench.py's pointwise_end_to_end loop with the phases timed separately (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scikit-downscale_amd"))
from skdownscale_amd import BcsdTemperature, PointWiseDownscaler, synth
from skdownscale_amd.core import GridArray
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
n_cells, ny = 8192, 64
index = synth.daily_calendar(14600)
cells = np.arange(n_cells)
X, y, Xp = (synth.tas_field(name, 0, index, cells, 100000).reshape(len(index), ny, n_cells // ny) for name in ("X_hist", "y_obs", "X_fut"))
mk = lambda a: GridArray(a, ("time", "y", "x"), {"time": index})
Xg, yg, Xpg = mk(X), mk(y), mk(Xp)
for it in range(6):
    t0 = time.perf_counter()
    model = PointWiseDownscaler(BcsdTemperature(return_anoms=True))
    t1 = time.perf_counter()
    model.fit(Xg, yg)
    t2 = time.perf_counter()
    res = model.predict(Xpg)
    t3 = time.perf_counter()
    _ = np.asarray(res.values if hasattr(res, "values") else res)
    t4 = time.perf_counter()
    keep = (it % 2 == 0)
    if not keep:
        del model, res
    t5 = time.perf_counter()
    print(f"it {it}: ctor {1e3*(t1-t0):.1f} fit {1e3*(t2-t1):.1f} predict {1e3*(t3-t2):.1f} values {1e3*(t4-t3):.1f} del {1e3*(t5-t4):.1f} ms (objects {'kept' if keep else 'deleted'})")
