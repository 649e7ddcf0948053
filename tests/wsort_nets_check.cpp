// Host check of the comparator networks the register wave sort ships (csrc/sd_wsort.h, constexpr tables): the per-lane
// sorter by the 0-1 principle over all 2^K inputs (K <= 20; larger K on 2^20 random 0-1 inputs plus the sorted / reversed
// patterns), the bitonic merger over every cyclic-bitonic 0-1 sequence, and the lane scheme (blocked bitonic sort over 64
// lane-blocks with min/max selected per lane) on random permutations.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "sd_wsort.h"

using sdws::CmpList;

static void apply(const CmpList& c, std::vector<int>& v) {
    for (int i = 0; i < c.n; ++i) {
        if (c.c[i] == sdws::kNone) {
            if (v[c.a[i]] > v[c.b[i]]) std::swap(v[c.a[i]], v[c.b[i]]);
        } else {
            int t[3] = {v[c.a[i]], v[c.b[i]], v[c.c[i]]};
            std::sort(t, t + 3);
            v[c.a[i]] = t[0];
            v[c.b[i]] = t[1];
            v[c.c[i]] = t[2];
        }
    }
}

template <int K>
int check() {
    int bad = 0;
    constexpr sdws::SortNet<K> snet{};
    constexpr sdws::BitonicNet<K> bnet{};
    std::mt19937 rng(K);
    const long cases = K <= 20 ? (1l << K) : (1l << 20);
    for (long x = 0; x < cases; ++x) {
        std::vector<int> v(K);
        const unsigned long bits = K <= 20 ? (unsigned long)x : ((unsigned long)rng() << 16) ^ rng();
        for (int i = 0; i < K; ++i) v[i] = (bits >> i) & 1;
        apply(snet.c, v);
        bad += !std::is_sorted(v.begin(), v.end());
    }
    // cyclic-bitonic 0-1 sequences: one run of ones (or of zeros) anywhere on the ring
    for (int start = 0; start < K; ++start)
        for (int len = 0; len <= K; ++len)
            for (int inv = 0; inv < 2; ++inv) {
                std::vector<int> v(K, inv);
                for (int t = 0; t < len; ++t) v[(start + t) % K] = 1 - inv;
                apply(bnet.c, v);
                bad += !std::is_sorted(v.begin(), v.end());
            }
    // the lane scheme of wave_sort on random permutations (host model of the cross-lane stages)
    for (int trial = 0; trial < 8; ++trial) {
        std::vector<std::vector<int>> k(64, std::vector<int>(K));
        std::vector<int> perm(64 * K);
        for (int i = 0; i < 64 * K; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < K; ++i) k[l][i] = perm[l * K + i];
        for (int l = 0; l < 64; ++l) apply(snet.c, k[l]);
        auto stage = [&](int X, int bit, bool rev) {
            auto old = k;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < K; ++i) {
                    const int a = old[l][i], b = old[l ^ X][rev ? K - 1 - i : i];
                    k[l][i] = ((l >> bit) & 1) ? std::max(a, b) : std::min(a, b);
                }
        };
        for (int L = 1; L <= 6; ++L) {
            const int M = 1 << (L - 1);
            stage(2 * M - 1, L - 1, true);
            for (int h = M / 2, b = L - 2; h >= 1; h >>= 1, --b) stage(h, b, false);
            for (int l = 0; l < 64; ++l) apply(bnet.c, k[l]);
        }
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < K; ++i) bad += k[l][i] != l * K + i;
    }
    std::printf("K=%d: sorter %d ops, bitonic merger %d ops: %s\n", K, snet.c.n, bnet.c.n, bad ? "FAILED" : "ok");
    return bad;
}

int main() {
    int bad = 0;
    bad += check<4>();
    bad += check<8>();
    bad += check<12>();
    bad += check<16>();
    bad += check<20>();
    bad += check<24>();
    return bad ? 1 : 0;
}
