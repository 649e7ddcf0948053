"""Random sweep of the fused PureAnalog call (sd_analog_fit_predict*: analog_f1_fused_kernel and its hand-back / fall-back paths)
against fit -> predict, bit for bit: random series lengths around the widths of the tile-shaped fit, query counts, cell counts, k,
kinds, thresholds, continuous / quantised / zero-inflated data, masked cells, non-finite training samples and queries, host and
resident fields.  usage: python tools/dev/fuzz_analog_fused.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scikit-downscale_amd"))
from skdownscale_amd.engine import default_context  # noqa: E402


def same_bits(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def main(n_cases, seed):
    rng = np.random.default_rng(seed)
    ctx = default_context()
    t0 = time.time()
    stats = {"cases": 0, "fused_size": 0, "quantised": 0, "resident": 0, "thresh": 0}
    for it in range(n_cases):
        long_series = rng.random() < 0.75
        T = int(rng.integers(9217, 15361)) if long_series else int(rng.integers(40, 9217))
        Tq = int(rng.integers(1, 16385)) if rng.random() < 0.8 else int(rng.integers(16385, 21000))
        C = int(rng.integers(1, 24))
        k = int(rng.choice([1, 2, 5, 30, 64, 200]))
        k = min(k, T)
        style = rng.choice(["continuous", "continuous", "quantised", "zero_inflated"])
        X = rng.standard_normal((T, 1, C))
        if style == "quantised":
            X = np.round(X * 32) / 32
        if style == "zero_inflated":
            X = np.where(rng.random(X.shape) < 0.5, 0.0, np.abs(X))
        y = X[:, 0, :] + rng.standard_normal((T, C))
        Xq = 1.2 * rng.standard_normal((Tq, 1, C))
        if C > 2 and rng.random() < 0.3:
            X[0, 0, 1] = np.nan
        if C > 3 and rng.random() < 0.2:
            X[int(rng.integers(1, T)), 0, 2] = np.inf
        if C > 1 and rng.random() < 0.2:
            Xq[int(rng.integers(0, Tq)), 0, 0] = np.nan
        kind = int(rng.choice([3, 3, 3, 0, 2]))
        thresh = float(rng.normal()) if rng.random() < 0.15 else None
        resident = rng.random() < 0.4
        a = (ctx.to_device(X), ctx.to_device(y), ctx.to_device(Xq)) if resident else (X, y, Xq)
        st = ctx.analog_fit(a[0], a[1])
        ref, sref = ctx.analog_predict(st, a[2], k, kind, thresh=thresh)
        st.close()
        got, sgot = ctx.analog_fit_predict(a[0], a[1], a[2], k, kind, thresh=thresh)
        if resident:
            ref, got = ref.to_host(), got.to_host()
        what = f"case {it}: T={T} Tq={Tq} C={C} k={k} kind={kind} thresh={thresh} style={style} resident={resident}"
        assert sgot.tolist() == sref.tolist(), what
        assert same_bits(got, ref), what
        stats["cases"] += 1
        stats["fused_size"] += long_series and Tq <= 16384 and ((kind == 3 and thresh is None) or k == 1)
        stats["quantised"] += style != "continuous"
        stats["resident"] += resident
        stats["thresh"] += thresh is not None
    print(f"fuzz_analog_fused: {stats} all bit-identical, seed {seed}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
