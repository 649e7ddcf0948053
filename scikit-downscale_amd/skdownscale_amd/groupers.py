"""Time-group key functions with the reference's names (groupers.py:11-16 of the reference).

The engine needs one integer group id per time step, shared by all cells; ``group_keys`` evaluates
a grouper on a pandas index once per call (the reference re-evaluates it 6x per cell through
``df.groupby(callable)``, ~60 % of its per-cell time -- SURVEY.md section 6).
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd


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


class PaddedDOYGrouper:
    """Day-of-year groups padded by +-``offset`` days (groupers.py:19-89): iterating yields ``(doy, rows)`` for doy = 1..366,
    where ``rows`` are the samples of ``df`` whose day of year lies in the 2*offset+1 day window centred on ``doy`` --
    evaluated separately on the 366-day calendar of leap years and the 365-day calendar of the other years (windows wrap
    around the year end), leap-year rows first.  ``time_grouper='daily_nasa-nex'`` of the BCSD estimators swaps this class
    in (bcsd.py:36-38); the engine gets the same groups as one table through ``padded_doy_table``."""

    def __init__(self, df, offset=15):
        self.n = 1
        self.df = df
        self.max = 366
        self.offset = offset
        idx = df.index
        self.leap = "leap" if ((idx.month == 2) & (idx.day == 29)).any() else "noleap"
        self.df_leap = df[idx.is_leap_year]
        self.df_noleap = df[~idx.is_leap_year]

    def _window(self, n, ndays):
        """days of year of the window around day n on an ndays-day calendar, as the reference builds it from the wrapped
        calendar (groupers.py:36-63): offset days before, n itself, and the days after (one fewer when n lies beyond
        the calendar, i.e. n = 366 on the 365-day calendar)"""
        wrapped = np.pad(np.arange(1, ndays + 1), self.offset, mode="wrap")  # groupers.py:36-39
        i = n - 1
        return np.concatenate([wrapped[i:i + self.offset], [n], wrapped[n + self.offset:i + 2 * self.offset + 1]])

    def __iter__(self):
        self.n = 1
        return self

    def __next__(self):
        if self.n > self.max:
            raise StopIteration
        n, total = self.n, 2 * self.offset + 1
        days_leap, days_noleap = self._window(n, 366), self._window(n, 365)
        if len(set(days_leap)) != total and self.leap == "noleap":
            warnings.warn("leap days not included, day groups in leap years missing leap days")
        if len(set(days_noleap)) != total and n != 366:
            raise ValueError("no leap day groups do not contain the correct set of days")
        rows = pd.concat([self.df_leap[self.df_leap.index.dayofyear.isin(days_leap)],
                          self.df_noleap[self.df_noleap.index.dayofyear.isin(days_noleap)]])
        self.n += 1
        return n, rows

    def mean(self):
        """[366, 1] frame of the group means of the first column, indexed by day of year (inf where never set)"""
        means = np.full((self.max, 1), np.inf)
        for key, rows in self:
            means[key - 1] = rows.mean().values[0]
        return pd.DataFrame(means, index=np.arange(1, self.max + 1))


def padded_doy_table(index, offset=15):
    """The 366 groups of ``PaddedDOYGrouper`` on ``index`` as one table for the engine: ``order`` = row positions group
    by group (leap-year rows first inside a group, like the reference's ``pd.concat``), ``offsets[367]``.  Vectorised:
    no DataFrame is sliced."""
    idx = pd.DatetimeIndex(index)
    doy = np.asarray(idx.dayofyear)
    leap = np.asarray(idx.is_leap_year)
    rows = np.arange(len(idx))
    probe = PaddedDOYGrouper(pd.DataFrame({"v": np.zeros(0)}, index=pd.DatetimeIndex([])), offset=offset)
    order, offsets = [], [0]
    for n in range(1, 367):
        days_leap, days_noleap = probe._window(n, 366), probe._window(n, 365)
        if len(set(days_noleap)) != 2 * offset + 1 and n != 366:
            raise ValueError("no leap day groups do not contain the correct set of days")  # groupers.py:67-68
        sel = np.concatenate([rows[leap & np.isin(doy, days_leap)], rows[~leap & np.isin(doy, days_noleap)]])
        order.append(sel)
        offsets.append(offsets[-1] + len(sel))
    return np.concatenate(order).astype(np.int32), np.asarray(offsets, dtype=np.int64)
