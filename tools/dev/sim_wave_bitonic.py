"""Host simulation of the LDS-free wave sort of sd_wsort.h (64 lanes x K registers, blocked layout).

Lane l ends up owning sorted positions K*l .. K*l+K-1.  Values are positive floats; a lane may hold them negated
(sign s[l] = -1), in which case "sorted ascending as stored" means descending in truth.  The only cross-lane operation
is  new[i] = min(own[i], -partner[i])  (v_min_f32 with a negated DPP / swizzle operand): the + lane of a pair keeps the
minimum, the - lane the negated maximum.
"""
import numpy as np


def local_sort(v):
    v.sort(axis=1)  # stands for the register sorting network / the bitonic merger (checked separately)


def simulate(keys, K, check_bitonic=True):
    W = 64
    v = keys.reshape(W, K).astype(np.float64).copy()
    lane = np.arange(W)
    sign = np.where(lane & 1, -1.0, 1.0)  # level 1: B lanes (bit 0) negative
    v *= sign[:, None]
    local_sort(v)
    nx = 0
    for L in range(1, 7):
        m = 1 << (L - 1)
        # step 1: reverse compare, partner = lane ^ (2m - 1); A lanes (+, forward), B lanes (-, reversed)
        want = np.where(lane & m, -1.0, 1.0)
        assert np.all(sign == want), (L, sign)
        part = lane ^ (2 * m - 1)
        v = np.minimum(v, -v[part])
        nx += 1
        h = m >> 1
        while h >= 1:
            want = np.where(lane & h, -1.0, 1.0)
            flip = want != sign
            v[flip] *= -1.0
            sign = want
            part = lane ^ h
            v = np.minimum(v, -v[part])
            nx += 1
            h >>= 1
        # sign required by the next level (all + after the last), set before the local merge
        want = np.where(lane & (2 * m), -1.0, 1.0) if L < 6 else np.ones(W)
        flip = want != sign
        v[flip] *= -1.0
        sign = want
        if check_bitonic:
            for l in range(W):
                d = np.sign(np.diff(v[l]))
                d = d[d != 0]
                changes = int(np.sum(d[1:] != d[:-1]))
                # cyclic-bitonic: at most 2 direction changes
                assert changes <= 2, (L, l, v[l])
        local_sort(v)
    return v, nx


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for K in (1, 2, 5, 18, 19, 20, 21, 32):
        for trial in range(20):
            keys = rng.permutation(64 * K).astype(np.float64) + 1.0
            out, nx = simulate(keys, K)
            assert np.all(out > 0)
            assert np.array_equal(out.reshape(-1), np.arange(64 * K) + 1.0), K
        print("K", K, "ok; cross-lane stages", nx)
