"""Comparator networks used by csrc/sd_wsort.h, generated and checked on the host (0-1 principle).

sort_net(K)     Batcher odd-even merge sort pruned to K inputs (comparators touching the +inf padding dropped),
                then comparators that never act on any 0-1 input removed.
bitonic_net(K)  sorts every cyclic-bitonic sequence of length K: even lengths are half-cleaned (i, i + n/2) and split,
                odd lengths fall back to a full sorter of that length.
"""
import itertools
import numpy as np


def batcher(K):
    N = 1
    while N < K:
        N *= 2
    out = []
    p = 1
    while p < N:
        k = p
        while k >= 1:
            j = k % p
            while j + k < N:
                for i in range(k):
                    lo, hi = i + j, i + j + k
                    if hi < N and lo // (2 * p) == hi // (2 * p) and hi < K:
                        out.append((lo, hi))
                j += 2 * k
            k //= 2
        p *= 2
    return out


def all01(K):
    """all 0-1 inputs of length K as a bit-sliced array: x[i] is a uint64 array, bit t of word w = input t's element i"""
    n = 1 << K
    idx = np.arange(n, dtype=np.uint64)
    cols = [((idx >> np.uint64(i)) & np.uint64(1)).astype(np.uint8) for i in range(K)]
    return cols


def prune(net, cols):
    cols = [c.copy() for c in cols]
    kept = []
    for (a, b) in net:
        swap = cols[a] & (1 - cols[b])  # a=1, b=0 -> acts
        if swap.any():
            kept.append((a, b))
            lo = cols[a] & cols[b]
            hi = cols[a] | cols[b]
            cols[a], cols[b] = lo, hi
    return kept, cols


def is_sorted01(cols):
    ok = np.ones_like(cols[0], dtype=bool)
    for i in range(len(cols) - 1):
        ok &= cols[i] <= cols[i + 1]
    return bool(ok.all())


def sort_net(K):
    if K <= 1:
        return []
    net = batcher(K)
    if K <= 22:
        kept, cols = prune(net, all01(K))
        assert is_sorted01(cols)
        return kept
    return net


def cyclic_bitonic01(K):
    seqs = set()
    for start in range(K):
        for ln in range(K + 1):
            s = [0] * K
            for t in range(ln):
                s[(start + t) % K] = 1
            seqs.add(tuple(s))
            seqs.add(tuple(1 - v for v in s))
    arr = np.array(sorted(seqs), dtype=np.uint8)
    return [arr[:, i].copy() for i in range(K)]


def bitonic_net(K, base=0):
    if K <= 1:
        return []
    if K % 2 == 0:
        h = K // 2
        net = [(base + i, base + i + h) for i in range(h)]
        return net + bitonic_net(h, base) + bitonic_net(h, base + h)
    return [(base + a, base + b) for (a, b) in sort_net(K)]


def check_bitonic(K):
    net = bitonic_net(K)
    cols = cyclic_bitonic01(K)
    kept, cols = prune(net, cols)
    assert is_sorted01(cols), K
    return net, kept


def check_bitonic5_sort3():
    """the 10-instruction network of sd_wsort.h for cyclic-bitonic sequences of 5: two comparators + two 3-sorters"""
    cols = cyclic_bitonic01(5)
    def ce(a, b):
        lo, hi = cols[a] & cols[b], cols[a] | cols[b]
        cols[a], cols[b] = lo, hi
    def s3(a, b, c):
        n = cols[a].astype(int) + cols[b] + cols[c]
        cols[a], cols[b], cols[c] = (n >= 3).astype(np.uint8), (n >= 2).astype(np.uint8), (n >= 1).astype(np.uint8)
    ce(0, 2); ce(1, 3); s3(0, 1, 4); s3(2, 3, 4)
    assert is_sorted01(cols)


if __name__ == "__main__":
    check_bitonic5_sort3()
    print("bitonic-5 by two comparators + two 3-sorters: ok")
    for K in (3, 4, 5, 8, 10, 12, 16, 18, 20, 22, 24):
        s = sort_net(K) if K <= 22 else batcher(K)
        net, kept = check_bitonic(K)
        print(f"K={K:2d}: sorter {len(batcher(K))} -> {len(s)} comparators; bitonic merger {len(net)} -> {len(kept)} needed")
