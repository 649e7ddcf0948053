// Wave-level ranking of 64 x K 32-bit keys by counting, for keys that are roughly evenly spread over their range.
//
// A lane holds the keys of K consecutive positions of its segment (blocked layout).  rank[i] = number of keys of the
// segment smaller than q[i] -- the position the key would take in the sorted segment -- without sorting anything:
//
//   A  a histogram over the top 11 key bits (2 048 bins, two 16-bit counts per LDS word, 4 KB) is zeroed;
//   B  every key bumps its bin with a returning LDS atomic (ds_add_rtn_u32): the value that comes back is the key's
//      arrival number among the keys of that bin;
//   C  the lanes turn the counts into exclusive prefix sums in place (32 bins per lane, one wave scan over the lane
//      totals) and find the largest bin population cmax;
//   D  every key is stored at qrow[prefix(bin) + arrival number]: the row now holds the keys grouped by bin, bins in
//      ascending order, the members of a bin in arrival order;
//   E  every key reads the members of its own bin -- qrow[prefix(bin) + t], t = 0 .. cmax-1 -- and counts those smaller
//      than itself.  Reads past the end of the bin see members of higher bins (or the 0xffffffff pads behind the row):
//      they compare greater and need no bounds test.  T members are read unconditionally, the wave loops further only
//      when some bin holds more than T keys.
//   F  equal keys would share a rank: the ranks of a segment are a permutation of 0 .. n-1 exactly when they sum to
//      n (n-1) / 2 (ties only ever lower the sum), which one wave reduction checks.
//
// About 35 vector and 9 LDS instructions per key, against ~80 + 6 for the register sorting network of sd_wsort.h and
// ~140 + 20 for the f64 merge sort of sd_wave.h; the price is that the cost depends on the data (cmax), so callers
// quantise with a clamp on outliers and fall back when cmax exceeds kMaxBin.
//
// LDS: 4 096 B histogram + 4 * (n + kMaxBin + 1) B keys, wave-private.
#pragma once
#include <hip/hip_runtime.h>

namespace sdwr {

constexpr int kBins = 2048;
constexpr int kHistBytes = kBins * 2;
constexpr int kMaxBin = 32;  // largest bin population served (pads behind the key row)

typedef __attribute__((address_space(3))) unsigned lds_u32_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u128_t;

__device__ __forceinline__ unsigned lds_ld(unsigned addr) { return *reinterpret_cast<lds_u32_t*>((uintptr_t)addr); }
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { *reinterpret_cast<lds_u32_t*>((uintptr_t)addr) = v; }
__device__ __forceinline__ unsigned lds_add_rtn(unsigned addr, unsigned v) {
    return __hip_atomic_fetch_add(reinterpret_cast<lds_u32_t*>((uintptr_t)addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wfence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp0(unsigned v) {  // lanes without a source (or masked out) read 0
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v) {
    v += dpp0<0x111, 0xF>(v);  // row_shr:1
    v += dpp0<0x112, 0xF>(v);  // row_shr:2
    v += dpp0<0x114, 0xF>(v);  // row_shr:4
    v += dpp0<0x118, 0xF>(v);  // row_shr:8
    v += dpp0<0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
    v += dpp0<0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)v, o, 64);
        v = v > t ? v : t;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
    v = wave_inclusive_sum(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ws: LDS byte address of the wave's workspace (16-byte aligned).  n: keys of the segment (positions K*lane + i < n take
// part; the others get rank = their position).  Returns the largest bin population; ranks are valid only if it is
// <= kMaxBin and *ok comes back true (no two equal keys).
#ifdef SD_WRANK_STAMPS
#define SD_STAMP(k) stamps[k] = __builtin_readcyclecounter()
#else
#define SD_STAMP(k)
#endif
template <int K, int T>
__device__ __forceinline__ int wave_rank(const unsigned (&q)[K], int n, int lane, unsigned ws, unsigned (&rank)[K], bool* ok
#ifdef SD_WRANK_STAMPS
                                         , unsigned long long* stamps
#endif
) {
    const unsigned qrow = ws + kHistBytes;
    const unsigned dummy = qrow + 4u * (unsigned)(n + kMaxBin);
    SD_STAMP(0);
    // ---- A
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        lds_u128_t* h = reinterpret_cast<lds_u128_t*>((uintptr_t)(ws + 64u * (unsigned)lane));
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = z;
        if (lane < kMaxBin) lds_st(qrow + 4u * (unsigned)(n + lane), 0xffffffffu);
    }
    wfence();
    SD_STAMP(1);
    // ---- B
    unsigned arr[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const bool in = K * lane + i < n;
        const unsigned a = ws + ((q[i] >> 20) & 0xFFCu);
        const unsigned sh = (q[i] >> 17) & 16u;
        const unsigned old = lds_add_rtn(a, in ? (1u << sh) : 0u);
        arr[i] = (old >> sh) & 0xffffu;
    }
    wfence();
    SD_STAMP(2);
    // ---- C
    unsigned cmax;
    {
        lds_u128_t* h = reinterpret_cast<lds_u128_t*>((uintptr_t)(ws + 64u * (unsigned)lane));
        unsigned c[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = h[j];
            c[4 * j] = v.x;
            c[4 * j + 1] = v.y;
            c[4 * j + 2] = v.z;
            c[4 * j + 3] = v.w;
        }
        unsigned run = 0, mlo = 0, mhi = 0;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const unsigned lo = c[d] & 0xffffu, hi = c[d] >> 16;
            mlo = mlo > lo ? mlo : lo;
            mhi = mhi > hi ? mhi : hi;
            const unsigned e1 = run + lo;
            c[d] = run | (e1 << 16);
            run = e1 + hi;
        }
        const unsigned excl = wave_inclusive_sum(run) - run;
        const unsigned add = excl * 0x10001u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = {c[4 * j] + add, c[4 * j + 1] + add, c[4 * j + 2] + add, c[4 * j + 3] + add};
            h[j] = v;
        }
        cmax = wave_max_u32(mlo > mhi ? mlo : mhi);
    }
    wfence();
    SD_STAMP(3);
    // ---- D
    unsigned base[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const bool in = K * lane + i < n;
        const unsigned a = ws + ((q[i] >> 20) & 0xFFCu);
        const unsigned sh = (q[i] >> 17) & 16u;
        const unsigned e = (lds_ld(a) >> sh) & 0xffffu;
        rank[i] = e;
        base[i] = qrow + 4u * e;
        lds_st(in ? base[i] + 4u * arr[i] : dummy, q[i]);  // positions past the segment write to a spare slot
    }
    wfence();
    SD_STAMP(4);
    // ---- E
#pragma unroll
    for (int i = 0; i < K; ++i) {
        unsigned mbr[T];
#pragma unroll
        for (int t = 0; t < T; ++t) mbr[t] = lds_ld(base[i] + 4u * (unsigned)t);
#pragma unroll
        for (int t = 0; t < T; ++t) rank[i] += mbr[t] < q[i] ? 1u : 0u;
    }
    if ((int)cmax > T) {  // wave-uniform, rare: a crowded bin somewhere in the segment
        const int lim = (int)cmax < kMaxBin ? (int)cmax : kMaxBin;
#pragma unroll 1
        for (int t = T; t < lim; ++t) {
#pragma unroll
            for (int i = 0; i < K; ++i) rank[i] += lds_ld(base[i] + 4u * (unsigned)t) < q[i] ? 1u : 0u;
        }
    }
    SD_STAMP(5);
    // ---- F
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = K * lane + i;
        rank[i] = j < n ? rank[i] : (unsigned)j;
        s += j < n ? rank[i] : 0u;
    }
    *ok = wave_sum_u32(s) == (unsigned)n * (unsigned)(n - 1) / 2u;
    wfence();
    SD_STAMP(6);
    return (int)cmax;
}

}  // namespace sdwr
