"""Shared helpers: regenerate the inputs of the golden cases (tests/golden/make_golden.py)."""
from __future__ import annotations

import os

import numpy as np

from skdownscale_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def month_gid(index):
    return np.asarray(index.month, dtype=np.int32) - 1


def tas_inputs(g, start="1980-01-01"):
    C, T, Tp, seed, c_full = (int(g[k]) for k in ("C", "T", "Tp", "seed", "c_full"))
    index = synth.daily_calendar(T, start)
    index_p = synth.daily_calendar(Tp, start)
    cells = np.arange(C)
    X = synth.tas_field("X_hist", seed, index, cells, c_full)
    y = synth.tas_field("y_obs", seed, index, cells, c_full)
    Xp = synth.tas_field("X_fut", seed, index_p, cells, c_full)
    return index, index_p, X, y, Xp


def pr_inputs(g):
    C, T, Tp, seed, c_full = (int(g[k]) for k in ("C", "T", "Tp", "seed", "c_full"))
    cells = np.arange(C)
    X = synth.pr_field("X_hist", seed, T, cells, c_full)
    y = synth.pr_field("y_obs", seed, T, cells, c_full)
    Xp = synth.pr_field("X_fut", seed, Tp, cells, c_full)
    return synth.daily_calendar(T), synth.daily_calendar(Tp), X, y, Xp


def nasanex_inputs(g, case):
    """inputs of tests/golden/g13_nasanex.npz (make_golden.py:g13_nasanex): tas and positive pr fields on two calendars"""
    import pandas as pd

    index = pd.date_range(str(g[f"start{case}"]), str(g[f"end{case}"]))
    index_p = pd.date_range(str(g[f"pstart{case}"]), str(g[f"pend{case}"]))
    cells = np.arange(int(g[f"C{case}"]))
    seed, c_full = 20, 1000
    tas = tuple(synth.tas_field(n, seed, i, cells, c_full) for n, i in (("X_hist", index), ("y_obs", index), ("X_fut", index_p)))
    pr = (synth.pr_field("X_hist", seed, len(index), cells, c_full), synth.pr_field("y_obs", seed, len(index), cells, c_full) + 0.25,
          synth.pr_field("X_fut", seed, len(index_p), cells, c_full))
    return index, index_p, tas, pr


def detrend_inputs(g):
    """inputs of tests/golden/g15_detrend.npz (make_golden.py:g15_detrend): drifting tas fields and their positive twins"""
    import pandas as pd

    index = pd.date_range(str(g["start"]), str(g["end"]))
    index_p = pd.date_range(str(g["pstart"]), str(g["pend"]))
    cells = np.arange(int(g["C"]))
    T, Tp = len(index), len(index_p)
    drift, drift_p = 2e-4 * np.arange(T)[:, None] * (1 + cells), 5e-4 * np.arange(Tp)[:, None] * (1 + cells)
    X = synth.tas_field("X_hist", 22, index, cells, 1000) + drift
    y = synth.tas_field("y_obs", 22, index, cells, 1000) + 0.5 * drift
    Xp = synth.tas_field("X_fut", 22, index_p, cells, 1000) + drift_p
    return index, index_p, (X, y, Xp), (np.abs(X) + 0.1, np.abs(y) + 0.1, np.abs(Xp) + 0.1)


def analog_inputs(g):
    T, Tq, C, F, seed, c_full = (int(g[k]) for k in ("T", "Tq", "C", "F", "seed", "c_full"))
    return synth.analog_fields(seed, T, np.arange(C), c_full, n_query=Tq, n_features=F)


def assert_close(actual, expected, rtol=1e-6, scale=None, what=""):
    """north_star tolerance: 1e-6 relative (float64) with an absolute floor tied to the data scale
    (anomalies are differences of near-equal numbers)."""
    actual = np.asarray(actual)
    expected = np.asarray(expected)
    assert actual.shape == expected.shape, (what, actual.shape, expected.shape)
    nan_a, nan_e = np.isnan(actual), np.isnan(expected)
    assert np.array_equal(nan_a, nan_e), f"{what}: NaN pattern differs"
    fin = ~nan_e
    if scale is None:
        scale = float(np.std(expected[fin])) if fin.any() else 1.0
    atol = rtol * max(scale, 1e-300)
    err = np.abs(actual[fin] - expected[fin])
    tol = atol + rtol * np.abs(expected[fin])
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} mismatches, max err {err.max():.3e} (tol {tol[bad].min():.3e})"
    return float(err.max()) if err.size else 0.0
