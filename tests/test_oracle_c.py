"""CPU: the plain-C restatement (cpu_baseline 'port') reproduces the reference's golden vectors."""
import numpy as np
import pytest

import bcsd_oracle as bo
import c_oracle
from _cases import assert_close, load, month_gid, pr_inputs, tas_inputs

pytestmark = pytest.mark.skipif(not c_oracle.available(), reason="oracle/libsd_oracle.so not built (python __graft_entry__.py)")


@pytest.mark.parametrize("name", ["g1_tas_same", "g2_tas_long", "g2_tas_short", "g1_tas_small", "g3_pr_same", "g3_pr_long", "g3_pr_small"])
def test_c_oracle_golden(name):
    g = load(name)
    kind = 0 if str(g["kind"]) == "tas" else 1
    index, index_p, X, y, Xp = (tas_inputs if kind == 0 else pr_inputs)(g)
    for key, ra in (("out_anoms", True), ("out_abs", False)):
        out, st = c_oracle.bcsd_fit_predict(kind, X, y, Xp, month_gid(index), month_gid(index_p), return_anoms=ra, nthreads=2)
        assert_close(out, g[key], what=f"{name}/{key}")
        assert np.array_equal(st, g["status"])


def test_c_oracle_status_and_numpy_agreement():
    g = load("g7_masked")
    index, index_p, X, y, Xp = tas_inputs(g)
    X[0, 1] = np.nan
    X[0, 4] = np.nan
    y[0, 1] = np.nan
    X[100, 2] = np.nan
    out, st = c_oracle.bcsd_fit_predict(0, X, y, Xp, month_gid(index), month_gid(index_p))
    exp, est = bo.pointwise_fit_predict(0, X, y, Xp, month_gid(index), month_gid(index_p))
    assert np.array_equal(st, est) and np.array_equal(st, g["status_nan_inside"])
    assert_close(out, exp, rtol=1e-12, what="C vs numpy oracle")
    g = load("g3_pr_badclimo")
    index, _, X, y, Xp = pr_inputs(g)
    y[np.asarray(index.month) == 7, 1] = 0.0
    out, st = c_oracle.bcsd_fit_predict(1, X, y, Xp, month_gid(index), month_gid(index))
    assert np.array_equal(st, g["status"])
    assert_close(out, g["out_anoms"], what="badclimo")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_oracle_random_sweep_against_the_numpy_oracle(seed):
    """The C port (the bench's cpu_baseline) against the NumPy oracle on random sizes, group counts and layouts, both kinds,
    predict series shorter / longer than the fitted one, tie-heavy dyadic data, masked and non-finite cells, 1-3 threads."""
    rng = np.random.default_rng(seed)
    for it in range(14):
        kind = int(rng.integers(0, 2))
        G = int(rng.choice([1, 3, 12, 12]))
        T = int(rng.integers(G * 12, 3000))
        Tp = int(rng.choice([T, rng.integers(G * 3, 3500)]))
        C = int(rng.integers(1, 70))
        gid = rng.integers(0, G, T).astype(np.int32) if rng.random() < 0.3 else (np.arange(T) * G // T).astype(np.int32)
        gid[:G] = np.arange(G)
        gid_p = (np.arange(Tp) % G).astype(np.int32)
        q = float(rng.choice([1, 4, 16]))
        if rng.random() < 0.4:  # dyadic grid: exact sums, so ties do not depend on the summation order
            f = lambda n: np.round((10 + 3 * rng.standard_normal((n, C))) * q) / q  # noqa: E731
        else:
            f = lambda n: 10 + 3 * rng.standard_normal((n, C))  # noqa: E731
        X, y, Xp = f(T), f(T) + 20, f(Tp)
        if kind == 1:
            X, y, Xp = np.abs(X) * (rng.random(X.shape) > 0.4), np.abs(y) + 0.1, np.abs(Xp) * (rng.random(Xp.shape) > 0.4)
        if C > 3:
            X[0, 1] = y[0, 1] = Xp[0, 1] = np.nan          # masked cell
            (X if kind == 0 else y)[T // 2, 2] = np.inf    # non-finite sample inside a live cell
        ra = bool(rng.integers(0, 2))
        exp, est = bo.pointwise_fit_predict(kind, X, y, Xp, gid, gid_p, G=G, return_anoms=ra)
        out, st = c_oracle.bcsd_fit_predict(kind, X, y, Xp, gid, gid_p, G=G, return_anoms=ra, nthreads=int(rng.integers(1, 4)))
        what = f"seed {seed} case {it}: kind={kind} G={G} T={T} Tp={Tp} C={C} return_anoms={ra}"
        assert np.array_equal(st, est), (what, st, est)
        assert_close(out, exp, rtol=1e-12, what=what)
