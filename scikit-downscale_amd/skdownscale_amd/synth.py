"""Deterministic synthetic climate fields (host mirror of ``csrc/sd_synth.hip``).

The bench needs 100k-1M cell grids resident in HBM; they are generated *on the device* by a
counter-based integer hash so no PCIe transfer is needed, and this NumPy mirror regenerates any
(t-range, cell-subset) window bit-identically on the host so the parity tests can hand the same
numbers to the CPU oracle (SURVEY.md section 8(d) "Synthetic inputs").

Everything transcendental (the seasonal cycle, the calendar) lives in host-computed tables that
are passed to the device; the per-sample arithmetic is integer hashing plus a handful of
IEEE-754 double add/mul in a fixed order (the device file is compiled with -ffp-contract=off).

Sample definition, for global cell index ``c`` (0 <= c < c_full) and time step ``t``::

    h0   = splitmix64(seed ^ splitmix64(stream))
    ctr  = t * c_full + c
    x_j  = splitmix64(h0 ^ (4*ctr + j)),  j = 0..3
    u_j  = (x_j >> 11) * 2**-53                       in [0, 1)
  kind GAUSS:
    g    = (((u0 + u1) + (u2 + u3)) - 2.0) * sqrt(3)  (Irwin-Hall, unit variance)
    out  = (base[t] + cell_scale * (c % 101)) + amp * g
    (optional second stream: out = out + amp2 * g2, g2 drawn from ``stream2``)
  kind PRECIP (zero inflated, gamma-like):
    out  = 0.0 if u0 < p_dry else ((amp * u1) * u2) * u3
"""
from __future__ import annotations

import numpy as np

GAUSS = 0
PRECIP = 1

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
SQRT3 = 1.7320508075688772


def splitmix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def stream_key(seed: int, stream: int) -> np.uint64:
    return np.uint64(seed) ^ splitmix64(np.uint64(stream))


def uniforms(seed, stream, t_idx, c_idx, c_full):
    """u[4, len(t_idx), len(c_idx)] float64 in [0,1)."""
    h0 = splitmix64(stream_key(seed, stream))
    t = np.asarray(t_idx, dtype=np.uint64)[:, None]
    c = np.asarray(c_idx, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ctr4 = (t * np.uint64(c_full) + c) * np.uint64(4)
        out = np.empty((4,) + ctr4.shape, dtype=np.float64)
        for j in range(4):
            x = splitmix64(h0 ^ (ctr4 + np.uint64(j)))
            out[j] = (x >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return out


def fill(kind, seed, stream, t_idx, c_idx, c_full, base=None, amp=1.0, cell_scale=0.0, p_dry=0.0,
         stream2=None, amp2=0.0):
    """Host mirror of ``sd_synth_fill``; returns float64 [len(t_idx), len(c_idx)]."""
    t_idx = np.asarray(t_idx)
    c_idx = np.asarray(c_idx)
    u = uniforms(seed, stream, t_idx, c_idx, c_full)
    if kind == GAUSS:
        g = (((u[0] + u[1]) + (u[2] + u[3])) - 2.0) * SQRT3
        b = np.zeros(len(t_idx)) if base is None else np.asarray(base, dtype=np.float64)[t_idx]
        off = cell_scale * (c_idx % 101).astype(np.float64)
        out = (b[:, None] + off[None, :]) + amp * g
        if stream2 is not None:
            u2 = uniforms(seed, stream2, t_idx, c_idx, c_full)
            g2 = (((u2[0] + u2[1]) + (u2[2] + u2[3])) - 2.0) * SQRT3
            out = out + amp2 * g2
        return out
    elif kind == PRECIP:
        wet = ((amp * u[1]) * u[2]) * u[3]
        return np.where(u[0] < p_dry, 0.0, wet)
    raise ValueError(f"unknown synth kind {kind}")


# ----------------------------------------------------------------------------------------------
# Calendar / workload tables: 40-year daily series starting 1980-01-01.
# ----------------------------------------------------------------------------------------------

def daily_calendar(periods=14600, start="1980-01-01"):
    import pandas as pd

    return pd.date_range(start, periods=periods, freq="D")


def season_table(index, amplitude=10.0):
    """S[t] = amplitude * sin(2*pi*(doy-1)/365.25 - pi/2)  (host-computed, passed to the device)."""
    doy = np.asarray(index.dayofyear, dtype=np.float64)
    return amplitude * np.sin(2.0 * np.pi * (doy - 1.0) / 365.25 - 0.5 * np.pi)


def tas_tables(index):
    """base tables + amplitudes for the three BcsdTemperature fields (X_hist, y_obs, X_fut)."""
    S = season_table(index)
    return {
        "X_hist": dict(stream=0, base=15.0 + S, amp=3.0, cell_scale=0.05),
        "y_obs": dict(stream=1, base=13.0 + 1.2 * S, amp=4.0, cell_scale=0.05),
        "X_fut": dict(stream=2, base=17.0 + S, amp=3.5, cell_scale=0.05),
    }


PR_FIELDS = {
    "X_hist": dict(stream=10, amp=40.0, p_dry=0.55),
    "y_obs": dict(stream=11, amp=48.0, p_dry=0.45),
    "X_fut": dict(stream=12, amp=44.0, p_dry=0.50),
}

ANALOG_FIELDS = {
    "X": dict(stream=20, amp=1.0),
    "noise": dict(stream=21, amp=1.0),
    "Xq": dict(stream=22, amp=1.0),
}


def tas_field(name, seed, index, c_idx, c_full, t_idx=None):
    tab = tas_tables(index)[name]
    t_idx = np.arange(len(index)) if t_idx is None else t_idx
    return fill(GAUSS, seed, tab["stream"], t_idx, c_idx, c_full, base=tab["base"], amp=tab["amp"],
                cell_scale=tab["cell_scale"])


def pr_field(name, seed, n_times, c_idx, c_full):
    tab = PR_FIELDS[name]
    return fill(PRECIP, seed, tab["stream"], np.arange(n_times), c_idx, c_full, amp=tab["amp"], p_dry=tab["p_dry"])


def analog_fields(seed, n_times, c_idx, c_full, n_query=None, n_features=1):
    """X [T,F,C], y [T,C] (= 2*X[:,0] + noise), Xq [Tq,F,C] for the PureAnalog workload."""
    n_query = n_times if n_query is None else n_query
    t = np.arange(n_times)
    X = np.stack([fill(GAUSS, seed, ANALOG_FIELDS["X"]["stream"] + 100 * f, t, c_idx, c_full) for f in range(n_features)], axis=1)
    y = fill(GAUSS, seed, ANALOG_FIELDS["X"]["stream"], t, c_idx, c_full, amp=2.0,
             stream2=ANALOG_FIELDS["noise"]["stream"], amp2=1.0)
    tq = np.arange(n_query)
    Xq = np.stack([fill(GAUSS, seed, ANALOG_FIELDS["Xq"]["stream"] + 100 * f, tq, c_idx, c_full) for f in range(n_features)], axis=1)
    return X, y, Xq
