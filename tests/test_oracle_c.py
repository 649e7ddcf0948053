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
