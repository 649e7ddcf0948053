"""Time-group key functions with the reference's names (groupers.py:11-16 of the reference).

The engine needs one integer group id per time step, shared by all cells; ``group_keys`` evaluates
a grouper on a pandas index once per call (the reference re-evaluates it 6x per cell through
``df.groupby(callable)``, ~60 % of its per-cell time -- SURVEY.md section 6).
"""
from __future__ import annotations

import numpy as np


def MONTH_GROUPER(x):
    return x.month


def DAY_GROUPER(x):
    return x.day


def group_keys(index, grouper):
    """Key of every time step, like ``df.groupby(grouper)`` applies ``grouper`` to each index label."""
    if grouper is MONTH_GROUPER and hasattr(index, "month"):
        return np.asarray(index.month)
    if grouper is DAY_GROUPER and hasattr(index, "day"):
        return np.asarray(index.day)
    return np.asarray([grouper(x) for x in index])
