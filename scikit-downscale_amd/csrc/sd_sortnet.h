// Compile-time sorting networks on registers + a workgroup-level merge sort built from them (device code,
// shared by the BCSD register-sort kernels and the analog fit).
#pragma once
#include <hip/hip_runtime.h>

namespace sdsort {

// min/max straight to the hardware instructions: __builtin_fmin/fmax make the compiler canonicalise every
// freshly loaded operand first (one extra v_max_f64 x,x per element and merge round).  NaNs never reach the
// sort of a cell whose result is kept (such cells are flagged non-finite and overwritten with NaN).
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// ---- Batcher odd-even merge sort network for K registers (built at compile time) ----------------
template <int K>
struct Net {
    static constexpr int N = K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : 64;
    static constexpr int kMax = 600;
    int n = 0;
    unsigned char a[kMax] = {}, b[kMax] = {};
    constexpr Net() {
        for (int p = 1; p < N; p <<= 1)
            for (int k = p; k >= 1; k >>= 1)
                for (int j = k % p; j + k < N; j += 2 * k)
                    for (int i = 0; i < k; ++i) {
                        const int lo = i + j, hi = i + j + k;
                        if (hi < N && lo / (2 * p) == hi / (2 * p) && hi < K) {  // comparators touching the +inf padding are no-ops
                            a[n] = (unsigned char)lo;
                            b[n] = (unsigned char)hi;
                            ++n;
                        }
                    }
    }
};

template <int K>
__device__ __forceinline__ void sort_registers(double (&v)[K]) {
    constexpr Net<K> net{};
#pragma unroll
    for (int c = 0; c < net.n; ++c) {
        const double lo = vmin(v[net.a[c]], v[net.b[c]]);
        const double hi = vmax(v[net.a[c]], v[net.b[c]]);
        v[net.a[c]] = lo;
        v[net.b[c]] = hi;
    }
}

constexpr int ceil_log2(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }

// ---- bitonic merger for K registers (built at compile time) --------------------------------------
// A lane's merge window is loaded as [A ascending | +inf filler | B descending] into w[0..K): a bitonic
// sequence.  Conceptually it is padded with -inf up to the next power of two N and pushed through the
// standard N-input bitonic merger; comparators against a known -inf are resolved at compile time
// (pure register renaming), so only ~2K real comparators remain.  out[s] names the register that
// holds the s-th smallest value afterwards.
template <int K>
struct MergeNet {
    static constexpr int N = K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : 64;
    int n = 0;
    unsigned char a[200] = {}, b[200] = {}, out[K] = {};
    constexpr MergeNet() {
        int reg[N] = {};
        bool ninf[N] = {};
        for (int s = 0; s < N; ++s) {
            reg[s] = s < K ? s : 0;
            ninf[s] = s >= K;
        }
        for (int h = N / 2; h >= 1; h >>= 1)
            for (int i = 0; i < N; ++i) {
                if (i & h) continue;
                const int j = i + h;
                if (ninf[i]) continue;  // min(-inf, x) stays put
                if (ninf[j]) {          // (x, -inf) -> (-inf, x): rename
                    reg[j] = reg[i];
                    ninf[j] = false;
                    ninf[i] = true;
                    continue;
                }
                a[n] = (unsigned char)reg[i];
                b[n] = (unsigned char)reg[j];
                ++n;
            }
        for (int s = 0; s < K; ++s) out[s] = (unsigned char)reg[N - K + s];
    }
};


// ---- workgroup-level merge sort: runs of K per thread -> buf[0..np) fully sorted ---------------------
// Same scheme as the wave-level sort of sd_bcsd_rs.hip (co-rank by binary search, windows merged in registers
// by the pruned bitonic merger), but the merge groups grow past one wave, so rounds are separated by
// workgroup barriers and a thread learns its neighbour's co-rank through `xch` (LDS, nthr + 1 ints).  While a merge
// group still fits one wave (rounds 0..5) the hardware's in-order LDS service per wave makes a wavefront-scope
// fence sufficient.
// np = slots being sorted, a multiple of K (pads = +inf are ordinary elements); v[] = the thread's run.
__device__ __forceinline__ void group_sync(bool within_wave) {
    if (within_wave) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// the merge rounds from round `first` on: buf holds sorted runs of K << first slots (first = 0: the threads' own runs; first = 6:
// runs of 64 * K produced elsewhere, e.g. by the wave sorts of analog_tile_sort_kernel)
template <int K>
__device__ __forceinline__ void block_merge_rounds(double* buf, int np, int* xch, int tid, int nthr, int first) {
    constexpr MergeNet<K> net{};
#pragma unroll 1
    for (int r = first; (K << r) < np; ++r) {
        const int L = K << r;
        const bool in_wave = (2 << r) <= 64;      // this round's merge groups do not cross a wave
        const bool next_in_wave = (4 << r) <= 64;  // ... nor do the next round's
        const int gl = tid & ((2 << r) - 1);  // thread within its merge group
        const int base = (tid - gl) * K;
        const int a0 = base < np ? base : np;
        const int a1 = base + L < np ? base + L : np;
        const int b1 = base + 2 * L < np ? base + 2 * L : np;
        const int LA = a1 - a0, LB = b1 - a1;
        const int d0 = gl * K;
        const bool busy = d0 < LA + LB;  // this thread owns K outputs of the pair (LA + LB is a multiple of K)
        const int d = busy ? d0 : LA + LB;
        // co-rank: smallest i with A[i] > B[d-1-i]; ties go to A
        int lo = d - LB > 0 ? d - LB : 0, hi = d < LA ? d : LA;
        const int nsteps = r + ceil_log2(K + 1);
        const double* pa0 = buf + a0;
        const double* pb0 = buf + a1 + d - 1;
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const int mid = (lo + hi) >> 1;  // lo == hi (finished): reads stay inside buf, updates are no-ops
            const bool le = (pa0[mid] <= pb0[-mid]) && (lo < hi);
            lo = le ? mid + 1 : lo;
            hi = le ? hi : mid;
        }
        xch[tid] = lo;
        group_sync(in_wave);
        const int inext = xch[tid + 1 < nthr ? tid + 1 : tid];
        const int ihi = (d + K >= LA + LB) ? LA : inext;  // co-rank of the end of this thread's window
        const int acnt = ihi - lo;                         // elements taken from A; K - acnt from B
        double w[K];
        if (busy) {
            const double* pa = buf + a0 + lo;
            const double* pq = buf + a1 + (d - lo) + (K - acnt) - 1 + acnt;
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const double* src = s < acnt ? pa : pq - 2 * s;  // (pq - 2s)[s] == pq[-s]
                w[s] = src[s];
            }
#pragma unroll
            for (int c = 0; c < net.n; ++c) {
                const double mn = vmin(w[net.a[c]], w[net.b[c]]);
                const double mx = vmax(w[net.a[c]], w[net.b[c]]);
                w[net.a[c]] = mn;
                w[net.b[c]] = mx;
            }
        }
        group_sync(in_wave);
        if (busy) {
            double* dst = buf + a0 + d;
#pragma unroll
            for (int s = 0; s < K; ++s) dst[s] = w[net.out[s]];
        }
        group_sync(next_in_wave);
    }
    __syncthreads();
}

template <int K>
__device__ __forceinline__ void block_merge_sort(double (&v)[K], double* buf, int np, int* xch, int tid, int nthr) {
    sort_registers<K>(v);
    if (K * tid < np) {
        double* dst = buf + K * tid;
#pragma unroll
        for (int i = 0; i < K; ++i) dst[i] = v[i];
    }
    group_sync(true);  // round 0 reads only the two runs of a lane pair
    block_merge_rounds<K>(buf, np, xch, tid, nthr, 0);
}

}  // namespace sdsort
