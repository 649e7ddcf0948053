"""Development helper (not a pytest file): exercise every register-sort K path in subprocesses."""
import subprocess
import sys

CASE = r'''
import sys, numpy as np, pandas as pd
sys.path[:0] = ["scikit-downscale_amd", "oracle", "tests"]
import bcsd_oracle as bo
from skdownscale_amd.engine import default_context
ctx = default_context()
T, Tp, C, mode = {T}, {Tp}, {C}, "{mode}"
rng = np.random.default_rng(T + Tp)
index = pd.date_range("1980-01-01", periods=T, freq="D"); index_p = pd.date_range("1980-01-01", periods=Tp, freq="D")
X, y, Xp = (15 + 8 * rng.standard_normal((n, C)) for n in (T, T, Tp))
gid, gid_p = bo.month_group_id(index), bo.month_group_id(index_p)
exp, est = bo.pointwise_fit_predict(0, X, y, Xp, gid, gid_p)
if mode == "fused":
    out, st = ctx.bcsd_fit_predict(0, ctx.to_device(X), ctx.to_device(y), gid, 12, ctx.to_device(Xp), gid_p)
    out = out.to_host()
else:
    s = ctx.bcsd_fit(0, X, y, gid, 12, True)
    out, st = ctx.bcsd_predict(s, Xp, gid_p)
err = np.abs(out - exp).max()
print("T", T, "Tp", Tp, "C", C, mode, "max err", err, "OK" if err < 1e-9 else "MISMATCH")
'''
import os
CASES = [(365, 365, 6), (1461, 1461, 6), (3650, 3650, 9), (8000, 8000, 8), (14600, 14600, 8), (14600, 20000, 5), (14600, 3000, 3), (20000, 14600, 4), (24000, 24000, 3)]
if os.environ.get('DEV_CASES') == 'k21':
    CASES = [(14600, 14600, 8), (14600, 3000, 3)]
for T, Tp, C in CASES:
    for mode in ("split", "fused"):
        r = subprocess.run([sys.executable, "-c", CASE.format(T=T, Tp=Tp, C=C, mode=mode)], capture_output=True, text=True)
        print(r.stdout.strip() or f"T {T} Tp {Tp} {mode}: CRASH rc={r.returncode}", (r.stderr.strip().splitlines() or [""])[-1][:200] if r.returncode else "")
