// Wave-level sort of 32-bit keys held in registers: 64 lanes x K keys, no LDS storage at all.
//
// Layout: blocked -- lane l holds K keys; afterwards lane l owns sorted positions K*l .. K*l + K-1 (ascending in
// k[0..K)).  The network is a bitonic sort over 64 *blocks*:
//
//   * every lane sorts its K keys with a comparator network (Batcher's odd-even merge sort pruned to K inputs);
//   * level L = 1..6 merges pairs of sorted runs of m = 2^(L-1) lanes.  Step 1 compares element i of lane l with
//     element K-1-i of lane l ^ (2m-1) (the partner run read backwards): the lower run keeps the minima and the upper
//     run the maxima, both as bitonic sequences.  Steps 2.. are half-cleaners between lanes l and l ^ h, h = m/2 .. 1
//     (same element index); they leave every lane with a bitonic sequence of K keys, which a local bitonic merger
//     sorts.  21 cross-lane stages and 6 local mergers in all.
//   * a cross-lane stage costs two vector instructions per key: fetch the partner's key (DPP move for lane ^ 1, 2, 3, 7,
//     8, 15; ds_swizzle -- the LDS crossbar, no LDS memory -- for lane ^ 4, 16, 31; ds_bpermute for lane ^ 63) and
//     v_med3_u32(own, partner, sel) with sel = 0 in the lanes that keep the minimum and 0xffffffff in those that keep
//     the maximum: median(a, b, 0) = min(a, b), median(a, b, ~0) = max(a, b).  No lane-dependent sign conventions, no
//     divergence; all keys stay in true ascending order.
//   * the local merger sorts cyclic-bitonic sequences: even lengths are half-cleaned and split; the odd lengths 3 and 5
//     (K = 12, 24; K = 20) end in 3-sorters (v_min3 / v_med3 / v_max3_u32), other odd lengths in a full sorter.
//
// K = 20: 202 + 6 * 80 + 21 * 20 * 2 = ~1 520 32-bit instructions per 1 280 keys, against ~2 800 mostly double-rate instructions and 410 LDS instructions of the f64 merge sort of
// sd_wave.h.  tools/dev/nets.py checks the networks (0-1 principle), tools/dev/sim_wave_bitonic.py the lane scheme.
#pragma once
#include <hip/hip_runtime.h>

namespace sdws {

constexpr int kMaxCmp = 400;

// a list of operations on registers: compare-exchange (a, b) -- min to a, max to b -- or, when c != kNone, a 3-sorter
// (a, b, c) -- min to a, median to b, max to c: v_min3_u32, v_med3_u32, v_max3_u32, three instructions for what three
// compare-exchanges (six instructions) do
constexpr unsigned char kNone = 0xff;
struct CmpList {
    int n = 0;
    unsigned char a[kMaxCmp] = {}, b[kMaxCmp] = {}, c[kMaxCmp] = {};
    constexpr void add(int lo, int hi) {
        a[n] = (unsigned char)lo;
        b[n] = (unsigned char)hi;
        c[n] = kNone;
        ++n;
    }
    constexpr void add3(int lo, int mid, int hi) {
        a[n] = (unsigned char)lo;
        b[n] = (unsigned char)mid;
        c[n] = (unsigned char)hi;
        ++n;
    }
};

// Batcher's odd-even merge sort for the inputs base .. base + K-1 (comparators that would touch the +inf padding up
// to the next power of two are no-ops and dropped)
constexpr void batcher_into(CmpList& out, int K, int base) {
    int N = 1;
    while (N < K) N <<= 1;
    for (int p = 1; p < N; p <<= 1)
        for (int k = p; k >= 1; k >>= 1)
            for (int j = k % p; j + k < N; j += 2 * k)
                for (int i = 0; i < k; ++i) {
                    const int lo = i + j, hi = i + j + k;
                    if (hi < N && lo / (2 * p) == hi / (2 * p) && hi < K) out.add(base + lo, base + hi);
                }
}

template <int K>
struct SortNet {
    CmpList c;
    constexpr SortNet() { batcher_into(c, K, 0); }
};

// sorts every cyclic-bitonic sequence of K keys
template <int K>
struct BitonicNet {
    CmpList c;
    constexpr BitonicNet() {
        int sb[64] = {}, sl[64] = {};
        int sp = 0;
        sb[sp] = 0;
        sl[sp] = K;
        ++sp;
        while (sp > 0) {
            --sp;
            const int base = sb[sp], len = sl[sp];
            if (len <= 1) continue;
            if (len % 2 == 0) {
                const int h = len / 2;
                for (int i = 0; i < h; ++i) c.add(base + i, base + i + h);
                sb[sp] = base + h;
                sl[sp] = h;
                ++sp;
                sb[sp] = base;
                sl[sp] = h;
                ++sp;
            } else if (len == 3) {
                c.add3(base, base + 1, base + 2);
            } else if (len == 5) {
                // found by exhaustive search over the cyclic-bitonic 0-1 inputs (tools/dev/nets.py checks it): 10 instructions
                c.add(base, base + 2);
                c.add(base + 1, base + 3);
                c.add3(base, base + 1, base + 4);
                c.add3(base + 2, base + 3, base + 4);
            } else {
                batcher_into(c, len, base);
            }
        }
    }
};

#ifdef __HIPCC__
__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
// median of three; the compiler matches max(min(a, b), min(max(a, b), c)) to v_med3_u32
__device__ __forceinline__ unsigned med3(unsigned a, unsigned b, unsigned c) { return umax(umin(a, b), umin(umax(a, b), c)); }

// the value lane (l ^ X) holds in v
template <int X>
__device__ __forceinline__ unsigned lane_xor(unsigned v, int addr63) {
#ifdef SD_WSORT_SWZ
    // A/B: partners below lane ^ 32 through the LDS crossbar instead of DPP moves (one vector instruction per key and stage less,
    // one LDS-pipe instruction more); SD_WSORT_SWZ is a bit mask over X
    if constexpr (X < 32 && ((SD_WSORT_SWZ >> (X == 1 ? 0 : X == 2 ? 1 : X == 3 ? 2 : X == 7 ? 3 : X == 8 ? 4 : X == 15 ? 5 : 6)) & 1))
        return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (X << 10) | 0x1F);
    else
#endif
    if constexpr (X == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    else if constexpr (X == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (X == 3) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
    else if constexpr (X == 7) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (X == 15) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    else if constexpr (X == 8) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);   // row_ror:8
    else if constexpr (X == 4) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);               // and 0x1f, xor 4
    else if constexpr (X == 16) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);              // and 0x1f, xor 16
    else if constexpr (X == 31) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x7C1F);              // and 0x1f, xor 31
    else {
        static_assert(X == 63, "unsupported lane permutation");
        return (unsigned)__builtin_amdgcn_ds_bpermute(addr63, (int)v);
    }
}

template <int K, typename Net>
__device__ __forceinline__ void apply_net(unsigned (&k)[K], const Net& net) {
#pragma unroll
    for (int c = 0; c < net.c.n; ++c) {
        if (net.c.c[c] == kNone) {
            const unsigned lo = umin(k[net.c.a[c]], k[net.c.b[c]]);
            const unsigned hi = umax(k[net.c.a[c]], k[net.c.b[c]]);
            k[net.c.a[c]] = lo;
            k[net.c.b[c]] = hi;
        } else {
            // written out: the compiler re-associates min(min(x, z), y) to share min(x, y) with its v_med3 pattern and then
            // emits five instructions instead of three
            const unsigned x = k[net.c.a[c]], y = k[net.c.b[c]], z = k[net.c.c[c]];
            unsigned lo, mid, hi;
            asm("v_min3_u32 %0, %3, %4, %5\n\tv_med3_u32 %1, %3, %4, %5\n\tv_max3_u32 %2, %3, %4, %5"
                : "=&v"(lo), "=&v"(mid), "=&v"(hi)
                : "v"(x), "v"(y), "v"(z));
            k[net.c.a[c]] = lo;
            k[net.c.b[c]] = mid;
            k[net.c.c[c]] = hi;
        }
    }
}

// Registers written by the inline assembly above may be read by a DPP move right away, and the compiler's hazard
// recognizer does not look inside inline assembly (a VALU write needs two wait states before a DPP read of the same
// register): every key passes through one "s_nop 1" after a network that uses 3-sorters.
constexpr bool uses_sort3(const CmpList& c) {
    for (int i = 0; i < c.n; ++i)
        if (c.c[i] != kNone) return true;
    return false;
}
template <int K>
__device__ __forceinline__ void dpp_fence(unsigned (&k)[K]) {
    if constexpr (!uses_sort3(BitonicNet<K>{}.c)) {
        return;
    } else if constexpr (K == 20) {
        asm volatile("s_nop 1"
                     : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(k[5]), "+v"(k[6]), "+v"(k[7]), "+v"(k[8]), "+v"(k[9]),
                       "+v"(k[10]), "+v"(k[11]), "+v"(k[12]), "+v"(k[13]), "+v"(k[14]), "+v"(k[15]), "+v"(k[16]), "+v"(k[17]), "+v"(k[18]),
                       "+v"(k[19]));
    } else if constexpr (K == 12) {
        asm volatile("s_nop 1"
                     : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(k[5]), "+v"(k[6]), "+v"(k[7]), "+v"(k[8]), "+v"(k[9]),
                       "+v"(k[10]), "+v"(k[11]));
    } else {
        static_assert(K == 24, "add the operand list for this K");
        asm volatile("s_nop 1"
                     : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(k[5]), "+v"(k[6]), "+v"(k[7]), "+v"(k[8]), "+v"(k[9]),
                       "+v"(k[10]), "+v"(k[11]), "+v"(k[12]), "+v"(k[13]), "+v"(k[14]), "+v"(k[15]), "+v"(k[16]), "+v"(k[17]), "+v"(k[18]),
                       "+v"(k[19]), "+v"(k[20]), "+v"(k[21]), "+v"(k[22]), "+v"(k[23]));
    }
}

// one cross-lane stage: partner = lane ^ X; the lane with bit BIT of its id clear keeps the minima.  REV: element i
// meets element K-1-i of the partner (the first stage of a level), otherwise element i.
template <int K, int X, int BIT, bool REV>
__device__ __forceinline__ void cross_stage(unsigned (&k)[K], int lane, int addr63) {
    const unsigned sel = (unsigned)(-((lane >> BIT) & 1));
    unsigned t[K];
#pragma unroll
    for (int i = 0; i < K; ++i) t[i] = lane_xor<X>(k[i], addr63);
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = med3(k[i], t[REV ? K - 1 - i : i], sel);
}

template <int K, int L>
__device__ __forceinline__ void merge_level(unsigned (&k)[K], int lane, int addr63) {
    constexpr int M = 1 << (L - 1);
    constexpr BitonicNet<K> bnet{};
    cross_stage<K, 2 * M - 1, L - 1, true>(k, lane, addr63);
    if constexpr (M >= 32) cross_stage<K, 16, 4, false>(k, lane, addr63);
    if constexpr (M >= 16) cross_stage<K, 8, 3, false>(k, lane, addr63);
    if constexpr (M >= 8) cross_stage<K, 4, 2, false>(k, lane, addr63);
    if constexpr (M >= 4) cross_stage<K, 2, 1, false>(k, lane, addr63);
    if constexpr (M >= 2) cross_stage<K, 1, 0, false>(k, lane, addr63);
    apply_net<K>(k, bnet);
    dpp_fence<K>(k);
}

// Sorts the wave's 64 * K keys; `lanes_used` (wave-uniform) = number of leading lanes that hold data: lanes beyond
// must hold keys >= every key of the lanes in use (pads), and merge levels that could only move pads are skipped.
template <int K>
__device__ __forceinline__ void wave_sort(unsigned (&k)[K], int lane, int lanes_used = 64) {
    constexpr SortNet<K> snet{};
    const int addr63 = (lane ^ 63) << 2;
    apply_net<K>(k, snet);
    if (lanes_used > 1) merge_level<K, 1>(k, lane, addr63);
    if (lanes_used > 2) merge_level<K, 2>(k, lane, addr63);
    if (lanes_used > 4) merge_level<K, 3>(k, lane, addr63);
    if (lanes_used > 8) merge_level<K, 4>(k, lane, addr63);
    if (lanes_used > 16) merge_level<K, 5>(k, lane, addr63);
    if (lanes_used > 32) merge_level<K, 6>(k, lane, addr63);
}
#endif  // __HIPCC__

}  // namespace sdws
