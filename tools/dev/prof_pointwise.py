"""Where the time of PointWiseDownscaler(BcsdTemperature()).fit().predict() on host grids goes (cProfile, 8 192 cells)."""
import cProfile, pstats, sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "scikit-downscale_amd"))
from skdownscale_amd import BcsdTemperature, PointWiseDownscaler, synth
from skdownscale_amd.core import GridArray

n_cells, ny = 8192, 64
index = synth.daily_calendar(14600)
cells = np.arange(n_cells)
X, y, Xp = (synth.tas_field(name, 0, index, cells, 100000).reshape(len(index), ny, n_cells // ny) for name in ("X_hist", "y_obs", "X_fut"))
mk = lambda a: GridArray(a, ("time", "y", "x"), {"time": index})
Xg, yg, Xpg = mk(X), mk(y), mk(Xp)
def once():
    m = PointWiseDownscaler(BcsdTemperature(return_anoms=True))
    t0 = time.perf_counter(); m.fit(Xg, yg); t1 = time.perf_counter(); r = m.predict(Xpg); _ = np.asarray(r.values); t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(3): print("fit %.3f s  predict %.3f s" % once())
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
