#!/usr/bin/env python3
"""Lane-level NumPy emulation of the wave merge sort's co-rank step (sd_sortnet.h / sd_bcsd_rs.hip merge_rounds):
checks the uniform-stride branch-free co-rank search against the definition, for every round / lane / random data.
Development aid (no GPU needed)."""
import numpy as np

def corank_ref(A, B, d):
    # smallest i in [max(0,d-LB), min(d,LA)] with i == hi or A[i] > B[d-1-i]
    LA, LB = len(A), len(B)
    lo, hi = max(0, d - LB), min(d, LA)
    i = lo
    while i < hi and A[i] <= B[d - 1 - i]:
        i += 1
    return i

def corank_uniform(row, a0, a1, b1, d, L):
    LA, LB = a1 - a0, b1 - a1
    lo0, hi0 = max(0, d - LB), min(d, LA)
    base = lo0
    ln = L + 1
    steps = 0
    while ln > 1:
        half = ln >> 1
        t = base + half
        ia, ib = a0 + t - 1, a1 + d - t
        av = row[ia] if 0 <= ia < len(row) else 0.0  # out-of-range LDS reads return garbage; masked by t <= hi0
        bv = row[ib] if 0 <= ib < len(row) else 0.0
        ok = (t <= hi0) and (av <= bv)
        base = t if ok else base
        ln -= half
        steps += 1
    return base, steps

def run(K, n, rng, ties=False):
    np_ = (n + K - 1) // K * K
    v = rng.standard_normal(np_)
    if ties:
        v = np.round(v * 3) / 3
    v[n:] = np.inf
    row = v.copy()
    for l in range(np_ // K):
        row[l * K:(l + 1) * K] = np.sort(row[l * K:(l + 1) * K])
    r = 0
    while (K << r) < np_:
        L = K << r
        new = row.copy()
        for lane in range(64):
            gl = lane & ((2 << r) - 1)
            base = (lane - gl) * K
            a0, a1, b1 = min(base, np_), min(base + L, np_), min(base + 2 * L, np_)
            LA, LB = a1 - a0, b1 - a1
            d0 = gl * K
            busy = d0 < LA + LB
            d = d0 if busy else LA + LB
            ref = corank_ref(row[a0:a1], row[a1:b1], d)
            got, steps = corank_uniform(row, a0, a1, b1, d, L)
            assert got == ref, (K, n, r, lane, got, ref)
            assert steps <= r + int(np.ceil(np.log2(K + 1))), (steps, r, K)
        # do the merge for the next round
        for g in range(0, np_, 2 * L):
            new[g:min(g + 2 * L, np_)] = np.sort(row[g:min(g + 2 * L, np_)], kind="stable")
        row = new
        r += 1
    assert np.array_equal(row, np.sort(v))

rng = np.random.default_rng(0)
for K in (5, 13, 19, 21, 33):
    for n in (1, 2, K, K + 1, 64 * K, 64 * K - 3, 31 * K + 7, 1240 if 64 * K >= 1240 else 64 * K - 1):
        if n > 64 * K: continue
        for ties in (False, True):
            run(K, n, rng, ties)
print("co-rank uniform search ok")
