#!/usr/bin/env python
"""Random sweep of analog_f1_mean_kernel's round-6 forms on the GPU: 'weight_analogs' (straight-line batches of 8 / 4 / 2 analogs + the
generic tail) and AnalogRegression (direct window sums for k <= 64, prefix differences above) against oracle/analog_oracle.py, any k,
with / without a threshold, continuous and tie-heavy data, short and long query series (time-ordered and value-ordered staging).
usage: python tools/dev/fuzz_mean_kernel.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scikit-downscale_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import analog_oracle as ao  # noqa: E402
from _cases import assert_close  # noqa: E402
from skdownscale_amd.engine import default_context  # noqa: E402


def main(n_cases=100, seed=0):
    ctx = default_context()
    rng = np.random.default_rng(seed)
    done = {"weight": 0, "weight_thresh": 0, "regression_direct": 0, "regression_prefix": 0}
    for it in range(n_cases):
        T = int(rng.integers(40, 3000)) if rng.random() < 0.7 else int(rng.integers(5000, 16385))
        Tq = int(rng.integers(1, 600)) if rng.random() < 0.6 else int(rng.integers(2048, 9000))
        C = int(rng.integers(1, 5))
        k = int(rng.integers(2, min(T, 90)))
        quant = rng.random() < 0.3
        X = rng.standard_normal((T, 1, C))
        Xq = 1.1 * rng.standard_normal((Tq, 1, C))
        if quant:
            X, Xq = np.round(X, 2), np.round(Xq, 2)
        y = 0.7 * X[:, 0, :] + rng.standard_normal((T, C))
        if rng.random() < 0.15:
            y = np.where(X[:, 0, :] > 0.2, y, 0.0)  # dry spells: constant analog sets
        st = ctx.analog_fit(X, y)
        sel = np.arange(Tq) if Tq <= 500 else np.unique(rng.integers(0, Tq, 500))
        thresh = None if rng.random() < 0.6 else 0.0
        out, _ = ctx.analog_predict(st, Xq, k, 2, thresh)  # weight_analogs
        exp = ao.pointwise_analog(X, y, Xq[sel], k, 2, thresh)
        assert_close(out[sel], exp, what=f"case {it} weight T={T} Tq={Tq} k={k} thresh={thresh} quant={quant}")
        done["weight" if thresh is None else "weight_thresh"] += 1
        if k >= 3 and not quant:
            out, _ = ctx.analogreg_predict(st, Xq, k)
            exp = ao.pointwise_analog(X, y, Xq[sel], k, 3, regression=True)
            assert_close(out[sel], exp, what=f"case {it} analogreg T={T} Tq={Tq} k={k}")
            done["regression_direct" if k <= 64 else "regression_prefix"] += 1
        st.close()
        if (it + 1) % 25 == 0:
            print(f"{it + 1} cases ok {done}", flush=True)
    print(f"fuzz_mean_kernel: {n_cases} cases ok (seed {seed}) {done}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
