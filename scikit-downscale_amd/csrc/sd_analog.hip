// GARD analog models (PureAnalog / AnalogRegression), batched over the cell axis.
//
// Reference (file:line under skdownscale/pointwise_models/gard.py): AnalogBase.fit 58-87 (KDTree),
// PureAnalog.predict 273-364, AnalogRegression.predict/_predict_one_step 152-224 (thresh=None).
// KDTree.query is restated as: k training rows with the smallest reduced distance
// rdist = sum_f (q_f - x_f)^2 (accumulated f = 0..F-1, no FMA), ascending by (rdist, index).
//
// fit   : mask / finite check, cell-major copies Xc[C][F][T], yc[C][T] (tiled LDS transpose); for F == 1
//         additionally the sorted view of a cell: xs[C][T] (values by (x, index)), xi[C][T] (their training
//         indices), yx[C][T] (y in that order) and pq[C][T+1][2] (prefix sums of the centred yx and its
//         squares) -- analog_sort2_kernel: workgroup merge sort (sd_sortnet.h); rx[C][T+1], the cross term of
//         the one-feature regression, is added by analog_rx_kernel on the first AnalogRegression call.  For
//         F > 1 a copy of the training points sorted by feature 0 (ps, indices xi).
// predict, F == 1 (one persistent workgroup per cell, queries and outputs through cell-major staging):
//   analog_f1_mean3_kernel  mean_analogs without a threshold: window search over xs in LDS, then the two prefix-sum
//                           components staged through the same LDS array (three generations per cell);
//   analog_f1_mean_kernel   a single analog, AnalogRegression, weighted / thresholded kinds: window search over xs
//                           in LDS, prefix sums or the window of yx read from memory (single pass);
//   analog_f1_window_kernel the other PureAnalog kinds: k-NN window over xs, statistics from yx, both
//                           LDS-resident per value range;
//   analog_f1_predict_kernel / f1_walk_query  exact (rdist, index)-ordered two-pointer walk: 'sample_analogs',
//                           neighbour outputs, and any query whose window has a tie on its boundary.
// predict, F > 1: analog_slab_predict_kernel (one wave per 64 queries sorted by feature 0, scalar-loaded training
//   points from the feature-0 sorted copy, only the reachable slab is scanned, top-k heap in LDS);
//   analog_bf2_predict_kernel (same scanner over the whole set in index order); analog_bf_predict_kernel
//   (LDS-staged tiles, lists in global scratch) for k > 208.
// Epilogues: PureAnalog statistics (gard.py:303-346), per-query least squares (gard.py:194-224).
#include <algorithm>
#include <cstdlib>

#include "sd_internal.h"
#include "sd_lsq.h"
#include "sd_sortnet.h"
#include "sd_wave.h"

namespace {

constexpr int kMaxF = sdlsq::kMaxF;

__device__ __forceinline__ bool sd_finite(double v) { return (__double_as_longlong(v) & 0x7ff0000000000000ll) != 0x7ff0000000000000ll; }

// XCD-aware persistent mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of
// cells and let its workgroups take adjacent cells at the same time, so 8-byte column reads of
// neighbouring cells merge into full lines in that XCD's L2.
__device__ __forceinline__ int64_t first_cell(int64_t C, int64_t* step, int64_t* end) {
    const int nb = gridDim.x, b = blockIdx.x;
    if (nb % 8 != 0) {
        *step = nb;
        *end = C;
        return b;
    }
    const int64_t cx = (C + 7) / 8;
    const int x = b % 8, j = b / 8;
    *step = nb / 8;
    *end = (x + 1) * cx < C ? (x + 1) * cx : C;
    return x * cx + j;
}

// ------------------------------------------------------------------------------------------------
// fit kernels
// ------------------------------------------------------------------------------------------------

// [R, C] (ld) -> [C][R] transpose through a 32x33 LDS tile, with mask / finite bookkeeping.
// plane f of X: rows are t*F + f.
__global__ void __launch_bounds__(256) analog_transpose_kernel(const double* __restrict__ src, int64_t ld, int64_t T, int F,
                                                               int f, int64_t C, double* __restrict__ dst /* [C][F][T] */,
                                                               int32_t* status, int set_mask) {
    __shared__ double tile[32][33];
    const int64_t t0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int64_t t = t0 + r, c = c0 + tx;
        double v = 0.0;
        if (t < T && c < C) {
            v = src[(t * F + f) * ld + c];
            if (set_mask && t == 0 && f == 0 && v != v) atomicOr(&status[c], SDI_MASKED);
            if (!sd_finite(v)) atomicOr(&status[c], SDI_NONFINITE);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t c = c0 + r, t = t0 + tx;
        if (t < T && c < C) dst[(c * F + f) * T + t] = tile[tx][r];
    }
}

// cell-major staging [C][3][Tq] -> output field [Tq, 3, ld] through a 32x33 LDS tile (grid: cells/32, Tq/32, 3)
// prob_from_pred: the probability plane of the staging buffer was not written; the column is 1 where the prediction is
// not NaN (no threshold: gard.py:346), NaN where it is
__global__ void __launch_bounds__(256) analog_untranspose_kernel(const double* __restrict__ oc, int64_t Tq, int64_t C,
                                                                 double* __restrict__ out, int64_t ld, int prob_from_pred) {
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    const int j = prob_from_pred ? 2 * (int)blockIdx.z : (int)blockIdx.z;  // grid z: 2 planes (pred [+ prob], err) or all 3
    for (int r = ty; r < 32; r += 8) {
        const int64_t c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < C && t < Tq) ? oc[(c * 3 + j) * Tq + t] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t t = t0 + r, c = c0 + tx;
        if (t < Tq && c < C) {
            const double v = tile[tx][r];
            out[(t * 3 + j) * ld + c] = v;
            if (prob_from_pred && j == 0) out[(t * 3 + 1) * ld + c] = v != v ? v : 1.0;
        }
    }
}

// F == 1: per-cell sort of (x, index) ascending, lexicographic.  One workgroup per cell, keys and
// 16-bit indices in LDS, truncated standard-form bitonic network (see sd_bcsd.hip).
__global__ void __launch_bounds__(1024) analog_sort_kernel(const double* __restrict__ Xc, const double* __restrict__ yc,
                                                           int64_t T, int64_t C, double* __restrict__ xs,
                                                           int32_t* __restrict__ xi, double* __restrict__ yx) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* key = reinterpret_cast<double*>(smem_raw);
    uint16_t* idx = reinterpret_cast<uint16_t*>(key + T);
    const int n = (int)T;
    int N = 1;
    while (N < n) N <<= 1;
    const int half = N >> 1;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* x = Xc + c * T;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            key[i] = x[i];
            idx[i] = (uint16_t)i;
        }
        __syncthreads();
        for (int size = 2; size <= N; size <<= 1) {
            const int hs = size >> 1;
            for (int stride = hs, first = 1; stride >= 1; stride >>= 1, first = 0) {
                for (int i = threadIdx.x; i < half; i += blockDim.x) {
                    int lo, hi;
                    if (first) {
                        const int blk = i / hs, off = i - blk * hs;
                        lo = blk * size + off;
                        hi = blk * size + size - 1 - off;
                    } else {
                        const int blk = i / stride, off = i - blk * stride;
                        lo = blk * 2 * stride + off;
                        hi = lo + stride;
                    }
                    if (hi < n) {
                        const double a = key[lo], b = key[hi];
                        const uint16_t ia = idx[lo], ib = idx[hi];
                        if (b < a || (b == a && ib < ia)) {
                            key[lo] = b; key[hi] = a;
                            idx[lo] = ib; idx[hi] = ia;
                        }
                    }
                }
                __syncthreads();
            }
        }
        const double* yy = yc + c * T;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            xs[c * T + i] = key[i];
            xi[c * T + i] = idx[i];
            if (yx) yx[c * T + i] = yy[idx[i]];
        }
        __syncthreads();
    }
}

// F == 1, fast form of the same result: two workgroup-level merge sorts of plain float64 keys (sd_sortnet.h).
//   1. sort x                       -> xs
//   2. every training sample finds lb = first position of its value in xs (binary search).  Without equal
//      values in the cell lb is the sorted position: y and the index are scattered through LDS.  Otherwise the
//      keys lb * 65536 + index are distinct integers < 2^32 (exact in float64) whose order is exactly the
//      lexicographic (x, index) order; sorting them yields xi, and yx = y[xi].
// One 1024-thread workgroup per cell, K consecutive samples per thread, T <= 1024 * K.
// keys_only: only xs and xi are produced (feature 0 of an F > 1 training set, or of a query series: x_stride is the
// distance between the series of consecutive cells); non-finite keys sort as 0 (their cell / query is flagged elsewhere,
// NaNs must not enter the min/max networks).
constexpr long long kTagMask = 0x3fff;      // 14 bits: series of up to 16 384 samples
constexpr unsigned kTagPadHi = 0x7fe00000u;  // upper word of the pad keys (>= 8.98e307: beyond any data the fast path accepts)

// ---- fit, F == 1, tile-shaped first stage ------------------------------------------------------------------------------
// analog_tile_sort_kernel<K>: one 512-thread workgroup = 8 adjacent cells x one chunk of 64 * K consecutive time steps, read as
// 64-byte row fragments of the time-major fields (the geometry of the BCSD kernels, sd_wave.h).  It does what the two staging
// transposes of X and y did (the cell-major copies the state keeps, with the mask / finite bookkeeping of
// analog_transpose_kernel) and, while the tile is on chip, sorts every cell's chunk of tagged keys with the wave sort: the
// sorted runs of 64 * K keys go to a scratch field and analog_sort2_kernel only has to merge them (rounds 6 ..), which is less
// than half of its work (measured: 9.8 instead of 23.1 ms per 100 000 cells with the register sort and rounds 0 .. 5 skipped).
// Keys are those of analog_sort2_kernel<K, true>: (x with -0.0 -> +0.0, non-finite -> 0) with the training index in the 14 low
// mantissa bits; slots past the series are pads (kTagPadHi, index).  A cell that holds a value in the pad range is reported in
// odd_flags (it takes the exact two-sort kernel, like in the single-kernel path).
template <int K>
__global__ void __launch_bounds__(sdw::kThreads, 4) analog_tile_sort_kernel(const double* __restrict__ X, const double* __restrict__ y,
                                                                            int64_t ld, int64_t T, int64_t C, int nchunks,
                                                                            double* __restrict__ Xc, double* __restrict__ yc,
                                                                            double* __restrict__ runs, int64_t runs_stride,
                                                                            int32_t* status, int32_t* odd_flags) {
    using namespace sdw;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int CHUNK = kWave * K;
    constexpr int NR = (CHUNK + kRowsPerPass - 1) / kRowsPerPass;  // rows a thread loads of one tile
    constexpr int RS = CHUNK + 2 + ((4 - (CHUNK + 2) % 4) + 2) % 4;  // row stride: >= CHUNK + 1 slots, RS % 4 == 2 (see sd_bcsd_rs_row_stride)
    double* const tile = reinterpret_cast<double*>(smem_raw) + kHeadDoubles;  // (no row at LDS address 0: sd_wave.h keeps "address - 8" positions)
    // workgroup -> (tile, chunk): XCD-aware like xcd_tile_of_block (tile-fastest inside an XCD)
    const int64_t ntiles = (C + kW - 1) / kW;
    int64_t tile_id;
    int q;
    xcd_tile_of_block(blockIdx.x, ntiles, &tile_id, &q);
    if (tile_id >= ntiles || q >= nchunks) return;
    const int64_t c0 = tile_id * kW;
    const int64_t r0 = (int64_t)q * CHUNK;
    const int nq = (int)(T - r0 < CHUNK ? T - r0 : CHUNK);  // valid rows of this chunk (> 0)
    const int tid = tid_now();
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t cpair = c0 + 2 * cp;
    const bool vec = (ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && cpair + 1 < C;
    // ---- both tiles are requested at once ----
    double x0[NR], x1[NR], y0[NR], y1[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = rr + k * kRowsPerPass;
        const int64_t row = r0 + (r < nq ? r : 0);
        const double* px = X + row * ld + cpair;
        if (vec) {
            const double2 v = *reinterpret_cast<const double2*>(px);
            x0[k] = v.x;
            x1[k] = v.y;
        } else {
            x0[k] = cpair < C ? px[0] : 0.0;
            x1[k] = cpair + 1 < C ? px[1] : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = rr + k * kRowsPerPass;
        const int64_t row = r0 + (r < nq ? r : 0);
        const double* py = y + row * ld + cpair;
        if (vec) {
            const double2 v = *reinterpret_cast<const double2*>(py);
            y0[k] = v.x;
            y1[k] = v.y;
        } else {
            y0[k] = cpair < C ? py[0] : 0.0;
            y1[k] = cpair + 1 < C ? py[1] : 0.0;
        }
    }
    // ---- X tile -> rows; mask (core.py:35-37: first sample of X is NaN) and finite bookkeeping ----
    {
        double* d0 = tile + (2 * cp) * RS;
        double* d1 = d0 + RS;
        int bits0 = 0, bits1 = 0;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rr + k * kRowsPerPass;
            if (r < nq) {
                if (r0 + r == 0) {
                    bits0 |= x0[k] != x0[k] ? SDI_MASKED : 0;
                    bits1 |= x1[k] != x1[k] ? SDI_MASKED : 0;
                }
                bits0 |= sd_finite(x0[k]) ? 0 : SDI_NONFINITE;
                bits1 |= sd_finite(x1[k]) ? 0 : SDI_NONFINITE;
                bits0 |= sd_finite(y0[k]) ? 0 : SDI_NONFINITE;
                bits1 |= sd_finite(y1[k]) ? 0 : SDI_NONFINITE;
                d0[r] = x0[k];
                d1[r] = x1[k];
            }
        }
        if (bits0 && cpair < C) atomicOr(&status[cpair], bits0);
        if (bits1 && cpair + 1 < C) atomicOr(&status[cpair + 1], bits1);
    }
    __syncthreads();
    const int64_t c = c0 + wave;
    const bool cell_ok = c < C;
    double* const row = tile + wave * RS;
    {
        // cell-major copy of the chunk (coalesced: the wave writes 512 consecutive bytes per step)
        if (cell_ok) {
            double* dst = Xc + c * T + r0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = lane + i * kWave;
                if (j < nq) dst[j] = row[j];
            }
        }
        // tagged keys of the K consecutive samples this lane owns
        double v[K];
        bool odd = false;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int jl = K * lane + i;  // (lane stride K is odd: conflict-free)
            const int64_t j = r0 + jl;    // training index
            double xv = row[jl < nq ? jl : 0];
            xv = sd_finite(xv) ? xv : 0.0;
            const long long b = __double_as_longlong(xv + 0.0);  // -0.0 -> +0.0: they tie as values
            odd |= jl < nq && (unsigned)((b >> 32) & 0x7fffffff) >= kTagPadHi;
            const long long key = jl < nq ? ((b & ~(long long)kTagMask) | (long long)j) : (((long long)kTagPadHi << 32) | (long long)j);
            v[i] = __longlong_as_double(key);
        }
        if (odd && cell_ok) atomicOr(&odd_flags[c], 1);
        wave_fence();
        sort_segment<K>(v, row, CHUNK, lane);  // every slot of the chunk is an element: pads sort behind the data
        if (cell_ok) {
            double* dst = runs + c * runs_stride + r0;
#pragma unroll
            for (int i = 0; i < K; ++i) dst[lane + i * kWave] = row[lane + i * kWave];
        }
    }
    __syncthreads();
    // ---- y tile -> rows -> cell-major copy ----
    {
        double* d0 = tile + (2 * cp) * RS;
        double* d1 = d0 + RS;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rr + k * kRowsPerPass;
            if (r < nq) {
                d0[r] = y0[k];
                d1[r] = y1[k];
            }
        }
    }
    __syncthreads();
    if (cell_ok) {
        double* dst = yc + c * T + r0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = lane + i * kWave;
            if (j < nq) dst[j] = row[j];
        }
    }
}

// TAGGED = true: the index-tag pass (below); cells it cannot serve are appended to `worklist` and the TAGGED = false
// instance (two sorts, any data) walks that list afterwards.  TAGGED = false with worklist == nullptr: every cell.
template <int K, bool TAGGED>
__global__ void __launch_bounds__(1024) analog_sort2_kernel(const double* __restrict__ Xc, int64_t x_stride, int keys_only,
                                                            const double* __restrict__ yc,
                                                            int64_t T, int64_t C, double* __restrict__ xs,
                                                            int32_t* __restrict__ xi, double* __restrict__ yx,
                                                            double* __restrict__ pq_all, double* __restrict__ ybar_all,
                                                            int32_t* worklist, int32_t* work_count,
                                                            const double* __restrict__ runs, int np_runs,
                                                            const int32_t* __restrict__ odd_flags) {
    // runs != nullptr (TAGGED only): the keys arrive as sorted runs of 64 * K slots, np_runs slots per cell
    // (analog_tile_sort_kernel): only the merge rounds 6 .. are left
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int np = (TAGGED && runs != nullptr) ? np_runs : (n + K - 1) / K * K;
    double* buf = reinterpret_cast<double*>(smem_raw);     // np + 1 doubles
    int* xch = reinterpret_cast<int*>(buf + np + 1);        // nthr + 1 ints (also 3 x 16 doubles of reduction scratch)
    double* red = reinterpret_cast<double*>(xch);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const int64_t nitems = (!TAGGED && worklist != nullptr) ? (int64_t)*work_count : C;
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int64_t c = (!TAGGED && worklist != nullptr) ? (int64_t)worklist[item] : item;
        const double* x = Xc + c * x_stride;
        auto load_x = [&]() {
            // coalesced load (all K + 1 requests of a thread in flight together), blocked reads afterwards
            double xv[K + 1];
#pragma unroll
            for (int t = 0; t <= K; ++t) {
                const int i = tid + t * nthr;
                xv[t] = i < n ? x[i] : inf;
            }
#pragma unroll
            for (int t = 0; t <= K; ++t) {
                const int i = tid + t * nthr;
                if (i <= np) buf[i] = (i < n && !sd_finite(xv[t])) ? 0.0 : xv[t];
            }
        };
        __syncthreads();
        if (!(TAGGED && runs != nullptr)) load_x();
        __syncthreads();
        if constexpr (TAGGED) {
            // ---- fast path: the training index rides through the sort in the 14 low mantissa bits of the key.  The sorted
            // order is then (upper 50 bits of x, index); it equals the (x, index) order whenever no two neighbouring sorted
            // keys share their upper 50 bits (checked: equal values, values closer than 2^-38 relative, and cells whose
            // magnitudes reach the pad range take the two-sort path below).  The tags of the sorted keys are xi, and xs / yx
            // are x / y gathered through them from LDS: 2 x K random LDS reads per thread instead of the 14 x K of the
            // first-position search.
            bool odd = false;
            if (runs != nullptr) {
                const double* rc = runs + c * (int64_t)np_runs;
                double kv[K + 1];
#pragma unroll
                for (int t2 = 0; t2 <= K; ++t2) {
                    const int i = tid + t2 * nthr;
                    kv[t2] = i < np ? rc[i] : inf;
                }
#pragma unroll
                for (int t2 = 0; t2 <= K; ++t2) {
                    const int i = tid + t2 * nthr;
                    if (i <= np) buf[i] = kv[t2];
                }
                odd = odd_flags[c] != 0;
                __syncthreads();
                sdsort::block_merge_rounds<K>(buf, np, xch, tid, nthr, 6);
            } else {
                double t[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const int j = K * tid + i;
                    const long long b = __double_as_longlong(buf[j < np ? j : np] + 0.0);  // (lane stride K is odd: conflict-free); -0.0 -> +0.0: they tie as values
                    odd |= j < n && (unsigned)((b >> 32) & 0x7fffffff) >= kTagPadHi;
                    const long long key = j < n ? ((b & ~(long long)kTagMask) | (long long)j) : (((long long)kTagPadHi << 32) | (long long)j);
                    t[i] = __longlong_as_double(key);
                }
                __syncthreads();
                sdsort::block_merge_sort<K>(t, buf, np, xch, tid, nthr);
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * tid + i;
                if (j + 1 < n) odd |= ((__double_as_longlong(buf[j]) ^ __double_as_longlong(buf[j + 1])) >> 14) == 0;
            }
            if (__syncthreads_or(odd) == 0) {
                unsigned short tg[K];
#pragma unroll
                for (int s2 = 0; s2 < K; ++s2) {
                    const int pos = tid + s2 * nthr;
                    tg[s2] = pos < n ? (unsigned short)(__double_as_longlong(buf[pos]) & kTagMask) : 0;
                    if (pos < n) xi[c * T + pos] = (int)tg[s2];
                }
                __syncthreads();  // every tag is in registers: the array is free
                {
                    double xv[K];
#pragma unroll
                    for (int s2 = 0; s2 < K; ++s2) {
                        const int pos = tid + s2 * nthr;
                        xv[s2] = pos < n ? x[pos] : 0.0;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < K; ++s2) {
                        const int pos = tid + s2 * nthr;
                        if (pos < n) buf[pos] = sd_finite(xv[s2]) ? xv[s2] : 0.0;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int s2 = 0; s2 < K; ++s2) {
                    const int pos = tid + s2 * nthr;
                    if (pos < n) xs[c * T + pos] = buf[tg[s2]];
                }
                if (keys_only) continue;
                __syncthreads();
                const double* yy = yc + c * T;
                double ysum = 0.0;
                {
                    double yv[K];
#pragma unroll
                    for (int s2 = 0; s2 < K; ++s2) {
                        const int pos = tid + s2 * nthr;
                        yv[s2] = pos < n ? yy[pos] : 0.0;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < K; ++s2) {
                        const int pos = tid + s2 * nthr;
                        if (pos < n) buf[pos] = yv[s2];
                        ysum += yv[s2];
                    }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) ysum += __shfl_xor(ysum, o, 64);
                if (lane == 0) red[wave] = ysum;  // (xch is free: the sort is over)
                __syncthreads();
#pragma unroll
                for (int s2 = 0; s2 < K; ++s2) {
                    const int pos = tid + s2 * nthr;
                    if (pos < n) yx[c * T + pos] = buf[tg[s2]];
                }
                double tot = 0.0;
                for (int w = 0; w < 16; ++w) tot += red[w];
                if (tid == 0) ybar_all[c] = tot / (double)n;
                continue;
            }
            if (tid == 0) worklist[atomicAdd(work_count, 1)] = (int32_t)c;  // left to the two-sort instance
            continue;
        }
        double v[K], orig[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            v[i] = buf[j < np ? j : np];  // lane stride K is odd: conflict-free
            orig[i] = v[i];
        }
        __syncthreads();
        sdsort::block_merge_sort<K>(v, buf, np, xch, tid, nthr);
        for (int i = tid; i < n; i += nthr) xs[c * T + i] = buf[i];
        // any two equal training values in this cell?  (then the order inside a tie run needs the second sort)
        bool tie = false;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            tie |= j + 1 < n && buf[j] == buf[j + 1];
        }
        const bool ties = __syncthreads_or(tie) != 0;
        // lb = number of sorted values < x (branch-free binary search; strides that are multiples of 16
        // doubles are shortened by one: see the rank search in sd_bcsd_rs.hip)
        int lb[K];
        {
            int pos[K];
#pragma unroll
            for (int i = 0; i < K; ++i) pos[i] = -1;  // index of the last element known to be < x
#pragma unroll 1
            for (int len = n; len > 1;) {
                int half = len >> 1;
                if ((half & 15) == 0) --half;
                len -= half;
#pragma unroll
                for (int i = 0; i < K; ++i) pos[i] += buf[pos[i] + half] < orig[i] ? half : 0;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) lb[i] = pos[i] + 1 + (buf[pos[i] + 1] < orig[i] ? 1 : 0);
        }
        __syncthreads();
        const double* yy = yc + c * T;
        if (!ties) {
            // distinct values: lb is the sorted position itself -> scatter the index, then y, through LDS
            int* ibuf = reinterpret_cast<int*>(buf);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * tid + i;
                if (j < n) ibuf[lb[i]] = j;
            }
            __syncthreads();
            for (int i = tid; i < n; i += nthr) xi[c * T + i] = ibuf[i];
            if (keys_only) continue;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * tid + i;
                if (j < n) buf[lb[i]] = yy[j];
            }
        } else {
            double key2[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * tid + i;
                key2[i] = j < n ? (double)lb[i] * 65536.0 + (double)j : inf;
            }
            sdsort::block_merge_sort<K>(key2, buf, np, xch, tid, nthr);
            double ya[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = tid + i * nthr;  // coalesced positions
                ya[i] = 0.0;
                if (j < n) {
                    const int idx = (int)((unsigned)buf[j] & 0xffffu);
                    xi[c * T + j] = idx;
                    if (!keys_only) ya[i] = yy[idx];
                }
            }
            if (keys_only) continue;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = tid + i * nthr;
                if (j < n) buf[j] = ya[i];
            }
        }
        __syncthreads();
        // buf[0..n) = y in sorted-x order: write it and its centred exclusive prefix sums (see analog_prefix_kernel)
        for (int i = tid; i < n; i += nthr) yx[c * T + i] = buf[i];
        double yv[K];
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            yv[i] = j < n ? buf[j] : 0.0;
            s += yv[i];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();  // (xch is free again)
        if (lane == 0) red[wave] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += red[w];
        const double ybar = tot / (double)n;
        if (tid == 0) ybar_all[c] = ybar;
        if (pq_all == nullptr) continue;  // the prefix sums are built when a kernel first needs them (ensure_prefix_sums)
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            const double d = j < n ? yv[i] - ybar : 0.0;
            yv[i] = d;
            a += d;
            b += d * d;
        }
        double ia = a, ib = b;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
            if (lane >= o) {
                ia += ta;
                ib += tb;
            }
        }
        __syncthreads();
        if (lane == 63) {
            red[16 + wave] = ia;
            red[32 + wave] = ib;
        }
        __syncthreads();
        double ra = ia - a, rb = ib - b;  // exclusive prefix at this thread's first element
        for (int w = 0; w < wave; ++w) {
            ra += red[16 + w];
            rb += red[32 + w];
        }
        double2* pq = reinterpret_cast<double2*>(pq_all) + c * (T + 1);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            if (j <= n) pq[j] = make_double2(ra, rb);
            ra += yv[i];
            rb += yv[i] * yv[i];
        }
        if (K * tid + K == n) pq[n] = make_double2(ra, rb);  // n = 1024 * K: no thread starts at position n
    }
}

struct Sort2Args {
    const double* X;   // series of cell c at X + c * x_stride
    int64_t x_stride;
    int keys_only;     // 1: only xs / xi
    const double* y;
    int64_t T, C;
    double* xs;
    int32_t* xi;
    double *yx, *pq, *ybar;
    // presorted runs of 64 * K tagged keys per cell from analog_tile_sort_kernel (np_runs slots per cell), or null
    const double* runs = nullptr;
    int np_runs = 0;
    const int32_t* odd_flags = nullptr;
};

template <int K>
int launch_sort2(sd_ctx* ctx, const Sort2Args& a) {
    int np = (int)((a.T + K - 1) / K * K);
    if (a.runs != nullptr && a.np_runs > np) np = a.np_runs;
    const size_t lds = sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_sort2_kernel<K, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_sort2_kernel<K, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int nb = (int)std::min<int64_t>(a.C, (int64_t)ctx->cu_count * 4);
    // index-tag pass first (series of up to 16 384 samples, no prefix sums asked for), then the cells it handed back
    const bool tagged = a.T <= kTagMask + 1 && a.pq == nullptr && a.C < ((int64_t)1 << 31) && sd_dev_env("SD_ANALOG_NOTAGS") == nullptr;
    sd_scratch list;
    int32_t* worklist = nullptr;
    int32_t* work_count = nullptr;
    if (tagged) {
        SD_HIP(list.alloc(ctx, sizeof(int32_t) * (size_t)(a.C + 1)));
        work_count = list.as<int32_t>();
        worklist = work_count + 1;
        SD_HIP(hipMemsetAsync(work_count, 0, sizeof(int32_t), ctx->stream));
        SD_LAUNCH(ctx, "analog_sort2_kernel", (analog_sort2_kernel<K, true>), dim3(nb), dim3(1024), lds, a.X, a.x_stride, a.keys_only, a.y,
                  a.T, a.C, a.xs, a.xi, a.yx, a.pq, a.ybar, worklist, work_count, tagged ? a.runs : nullptr, a.np_runs, a.odd_flags);
    }
    SD_LAUNCH(ctx, "analog_sort2_exact_kernel", (analog_sort2_kernel<K, false>), dim3(tagged ? std::min(nb, 256) : nb), dim3(1024), lds, a.X,
              a.x_stride, a.keys_only, a.y, a.T, a.C, a.xs, a.xi, a.yx, a.pq, a.ybar, worklist, work_count, (const double*)nullptr, 0,
              (const int32_t*)nullptr);
    if (tagged) SD_HIP(hipStreamSynchronize(ctx->stream));  // the list goes back to the block cache
#ifdef SD_DEV
    if (tagged && sd_dev_env("SD_ANALOG_COUNT")) {
        int32_t h = 0;
        SD_HIP(hipMemcpy(&h, work_count, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "analog sort: %d of %lld cells took the exact kernel (presorted runs: %d)\n", h, (long long)a.C, a.runs != nullptr);
    }
#endif
    return SD_OK;
}

// The tile-shaped first stage of the F == 1 fit (analog_tile_sort_kernel): writes the cell-major copies Xc / yc, the mask / finite
// status bits and the sorted runs.  Instantiated for the widths of the 40-year daily series and its neighbours.
bool tile_sort_applies(int K, int64_t T, int64_t C, size_t lds_max) {
    if (K != 13 && K != 15 && K != 17) return false;
    const int64_t chunk = 64 * K, nchunks = (T + chunk - 1) / chunk;
    if (T > kTagMask + 1 || C >= ((int64_t)1 << 31) || nchunks > 16) return false;
    if (sizeof(double) * (size_t)(nchunks * chunk + 1) + sizeof(int) * 1025 > lds_max) return false;
    return sd_dev_env("SD_ANALOG_NOTILE") == nullptr && sd_dev_env("SD_ANALOG_NOTAGS") == nullptr;
}

template <int K>
int launch_tile_sort_k(sd_ctx* ctx, const double* X, const double* y, int64_t ld, int64_t T, int64_t C, double* Xc, double* yc, double* runs,
                       int64_t runs_stride, int32_t* status, int32_t* odd_flags) {
    constexpr int CHUNK = 64 * K;
    constexpr int RS = CHUNK + 2 + ((4 - (CHUNK + 2) % 4) + 2) % 4;
    const int nchunks = (int)((T + CHUNK - 1) / CHUNK);
    const size_t lds = sizeof(double) * ((size_t)sdw::kW * RS + sdw::kHeadDoubles);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_tile_sort_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t ntiles = (C + sdw::kW - 1) / sdw::kW, tx = (ntiles + 7) / 8;
    const int64_t nblocks = 8 * tx * nchunks;
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "analog fit: grid too large");
    SD_LAUNCH(ctx, "analog_tile_sort_kernel", analog_tile_sort_kernel<K>, dim3((unsigned)nblocks), dim3(sdw::kThreads), lds, X, y, ld, T, C, nchunks,
              Xc, yc, runs, runs_stride, status, odd_flags);
    return SD_OK;
}
int launch_tile_sort(sd_ctx* ctx, int K, const double* X, const double* y, int64_t ld, int64_t T, int64_t C, double* Xc, double* yc, double* runs,
                     int64_t runs_stride, int32_t* status, int32_t* odd_flags) {
    switch (K) {
        case 13: return launch_tile_sort_k<13>(ctx, X, y, ld, T, C, Xc, yc, runs, runs_stride, status, odd_flags);
        case 15: return launch_tile_sort_k<15>(ctx, X, y, ld, T, C, Xc, yc, runs, runs_stride, status, odd_flags);
        case 17: return launch_tile_sort_k<17>(ctx, X, y, ld, T, C, Xc, yc, runs, runs_stride, status, odd_flags);
    }
    return sd_set_error(SD_ERR_INVALID, "analog tile sort: width %d not instantiated", K);
}

int launch_sort2_width(sd_ctx* ctx, int K, const Sort2Args& a) {
    switch (K) {
        case 5: return launch_sort2<5>(ctx, a);
        case 9: return launch_sort2<9>(ctx, a);
        case 13: return launch_sort2<13>(ctx, a);
        case 15: return launch_sort2<15>(ctx, a);
        case 17: return launch_sort2<17>(ctx, a);
        case 19: return launch_sort2<19>(ctx, a);
    }
    return sd_set_error(SD_ERR_INVALID, "analog sort: width %d not instantiated", K);
}

// widths instantiated for the fast sort: T <= 1024 * K and the keys must fit the LDS
int sort2_width(int64_t T, size_t lds_max) {
    const int widths[] = {5, 9, 13, 15, 17, 19};
    for (int K : widths) {
        const int64_t np = (T + K - 1) / K * K;
        if (T <= (int64_t)1024 * K && T <= 65535 && sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025 <= lds_max) return K;
    }
    return 0;
}

// F == 1: exclusive prefix sums of the centred analog values in sorted-x order, pq[c][i] = (sum_{j<i} d_j,
// sum_{j<i} d_j^2) with d = yx - mean(y).  The mean and standard deviation of any window of k consecutive analogs
// then cost two 16-byte loads (centring keeps the running sums small: no cancellation for the differences).
// One 1024-thread workgroup per cell: serial partial sums per thread, wave shuffles + LDS for the offsets.
__global__ void __launch_bounds__(1024) analog_prefix_kernel(const double* __restrict__ yx_all, int64_t T, int64_t C,
                                                             double* __restrict__ pq_all, double* __restrict__ ybar_all,
                                                             int keep_ybar /* 1: centre on the ybar_all given */) {
    __shared__ double wsum[2][16];
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + nthr - 1) / nthr;  // consecutive elements per thread
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* yx = yx_all + c * T;
        double2* pq = reinterpret_cast<double2*>(pq_all) + c * (T + 1);
        const int beg = tid * per < n ? tid * per : n, end = beg + per < n ? beg + per : n;
        // mean of y
        double s = 0.0;
        for (int i = beg; i < end; ++i) s += yx[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();
        if (lane == 0) wsum[0][wave] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += wsum[0][w];
        const double ybar = keep_ybar ? ybar_all[c] : tot / (double)n;
        if (tid == 0 && !keep_ybar) ybar_all[c] = ybar;
        // per-thread totals of d and d^2, exclusive scan across the workgroup
        double a = 0.0, b = 0.0;
        for (int i = beg; i < end; ++i) {
            const double d = yx[i] - ybar;
            a += d;
            b += d * d;
        }
        double ia = a, ib = b;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
            if (lane >= o) {
                ia += ta;
                ib += tb;
            }
        }
        __syncthreads();
        if (lane == 63) {
            wsum[0][wave] = ia;
            wsum[1][wave] = ib;
        }
        __syncthreads();
        double oa = 0.0, ob = 0.0;
        for (int w = 0; w < wave; ++w) {
            oa += wsum[0][w];
            ob += wsum[1][w];
        }
        double ra = oa + (ia - a), rb = ob + (ib - b);  // exclusive prefix at this thread's first element
        for (int i = beg; i < end; ++i) {
            pq[i] = make_double2(ra, rb);
            const double d = yx[i] - ybar;
            ra += d;
            rb += d * d;
        }
        if (end == n && beg < n) pq[n] = make_double2(ra, rb);
        if (n == 0 && tid == 0) pq[0] = make_double2(0.0, 0.0);
    }
}

// F == 1, one-feature AnalogRegression: rx[c][i] = sum_{j<i} (xs_j - mean(x)) (yx_j - mean(y)), the cross term of the
// window regression (analog_f1_mean_kernel), computed on the first regression call on a state.  The products are
// formed with coalesced reads into LDS, scanned there (odd number of consecutive elements per thread: conflict-free)
// and stored coalesced.
__global__ void __launch_bounds__(1024) analog_rx_kernel(const double* __restrict__ xs_all, const double* __restrict__ yx_all,
                                                         const double* __restrict__ ybar_all, int64_t T, int64_t C,
                                                         double* __restrict__ rx_all, double* __restrict__ xbar_all) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* e = reinterpret_cast<double*>(smem_raw);  // n + 1 doubles
    __shared__ double wsum[16];
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int per = ((n + nthr - 1) / nthr) | 1;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* xs = xs_all + c * T;
        const double* yx = yx_all + c * T;
        double s = 0.0;
        for (int i = tid; i < n; i += nthr) s += xs[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();
        if (lane == 0) wsum[wave] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        const double xbar = tot / (double)n, ybar = ybar_all[c];
        if (tid == 0) xbar_all[c] = xbar;
        for (int i = tid; i < n; i += nthr) e[i] = (xs[i] - xbar) * (yx[i] - ybar);
        __syncthreads();
        const int beg = tid * per < n ? tid * per : n, end = beg + per < n ? beg + per : n;
        double a = 0.0;
        for (int i = beg; i < end; ++i) a += e[i];
        double ia = a;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double ta = __shfl_up(ia, o, 64);
            if (lane >= o) ia += ta;
        }
        if (lane == 63) wsum[wave] = ia;  // (all reads of wsum above are behind the barrier before the products)
        __syncthreads();
        double ra = ia - a;
        for (int w = 0; w < wave; ++w) ra += wsum[w];
        for (int i = beg; i < end; ++i) {
            const double t = e[i];
            e[i] = ra;
            ra += t;
        }
        if (end == n && (beg < n || tid * per == n)) e[n] = ra;
        if (n == 0 && tid == 0) e[0] = 0.0;
        __syncthreads();
        double* rx = rx_all + c * (T + 1);
        for (int i = tid; i <= n; i += nthr) rx[i] = e[i];
    }
}

// ------------------------------------------------------------------------------------------------
// epilogues (run by the thread that owns the query; lists are [k][nthr] in scratch)
// ------------------------------------------------------------------------------------------------
struct PredictArgs;
__device__ __forceinline__ void put_out(const PredictArgs& pa, int64_t tq, int64_t c, double pred, double prob, double err);

struct PredictArgs {
    int k, kind, has_thresh;
    double thresh;
    const int32_t* sample;  // device [Tq, ld_s] or null
    int64_t ld_s;
    double* out;            // [Tq,3,ld_out]; windowed path: cell-major staging [C][3][Tq] (oc_Tq > 0)
    int64_t ld_out;
    int64_t oc_Tq;          // > 0: out is the cell-major staging buffer of a Tq-long query series
    int64_t* inds;          // [Tq,k,ld_out] or null
    double* dist;           // [Tq,k,ld_out] or null
    int32_t* one_class;     // per-cell status words (predict side): SDI_ONE_CLASS is set here
};

__device__ __forceinline__ void put_out(const PredictArgs& pa, int64_t tq, int64_t c, double pred, double prob, double err) {
    if (pa.oc_Tq > 0) {  // consecutive queries of a cell are consecutive in memory: coalesced across the workgroup
        double* o = pa.out + c * 3 * pa.oc_Tq + tq;
        o[0] = pred;
        o[pa.oc_Tq] = prob;
        o[2 * pa.oc_Tq] = err;
    } else {
        pa.out[(tq * 3 + 0) * pa.ld_out + c] = pred;
        pa.out[(tq * 3 + 1) * pa.ld_out + c] = prob;
        pa.out[(tq * 3 + 2) * pa.ld_out + c] = err;
    }
}

__device__ __forceinline__ double nan_to_num(double v) {
    if (v != v) return 0.0;
    if (v == __longlong_as_double(0x7ff0000000000000ll)) return 1.7976931348623157e308;
    if (v == __longlong_as_double(0xfff0000000000000ll)) return -1.7976931348623157e308;
    return v;
}

// PureAnalog statistics for one query (gard.py:301-346).  a[i] = analog values in neighbour order,
// read through `av(i)`; rd(i) = reduced distance.
template <typename AV, typename RD>
__device__ void pure_analog_stats(const PredictArgs& pa, int k, int kind, int sample_i, AV av, RD rd, double* pred,
                                  double* prob, double* err) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double sum = 0.0, wsum = 0.0, awsum = 0.0;
    int nexc = 0;
    bool any_masked = false;
    for (int i = 0; i < k; ++i) {
        const double a = av(i);
        const bool exc = !pa.has_thresh || a > pa.thresh;  // gard.py:307
        nexc += exc ? 1 : 0;
        any_masked |= !exc;
        sum += a;
        if (kind == SD_ANALOG_WEIGHT) {
            const double d = sqrt(rd(i));
            const double w = 1.0 / (d == 0.0 ? 1e-20 : d);  // gard.py:322-323
            wsum += w;
            awsum += a * w;
        }
    }
    double p;
    if (kind == SD_ANALOG_BEST) p = av(0);                               // gard.py:311
    else if (kind == SD_ANALOG_SAMPLE) p = av(sample_i);                 // gard.py:313-317
    else if (kind == SD_ANALOG_WEIGHT) p = any_masked ? nan : awsum / wsum;  // gard.py:319-327 (NaN-masked average)
    else p = any_masked ? nan : sum / (double)k;                         // gard.py:329-333
    if (pa.has_thresh) {
        p = nan_to_num(p);  // gard.py:341
        *prob = (double)nexc / (double)k;  // gard.py:343
    } else {
        *prob = 1.0;  // gard.py:346
    }
    if (any_masked) {
        *err = nan;  // gard.py:342 plain .std() of a NaN-masked row
    } else {
        const double mean = sum / (double)k;
        double ss = 0.0;
        for (int i = 0; i < k; ++i) {
            const double d = av(i) - mean;
            ss += d * d;
        }
        *err = sqrt(ss / (double)k);  // ddof = 0 (gard.py:342,345)
    }
    *pred = p;
}

// AnalogRegression for one query (gard.py:194-224): centred normal equations over the analogs selected by `use`
// (all of them without a threshold; those above it otherwise, gard.py:215: ne of them, ne >= 1).
template <typename XV, typename YV, typename USE>
__device__ void analog_regression(int k, int F, XV xv /* (i,f) */, YV yv /* (i) */, USE use /* (i) */, int ne, const double* q,
                                  double* pred, double* err) {
    double xm[kMaxF], A[kMaxF][kMaxF + 1], coef[kMaxF];
    double ym = 0.0;
    for (int f = 0; f < F; ++f) xm[f] = 0.0;
    for (int i = 0; i < k; ++i) {
        if (!use(i)) continue;
        ym += yv(i);
        for (int f = 0; f < F; ++f) xm[f] += xv(i, f);
    }
    ym /= (double)ne;
    for (int f = 0; f < F; ++f) xm[f] /= (double)ne;
    for (int f = 0; f < F; ++f)
        for (int g = 0; g <= F; ++g) A[f][g] = 0.0;
    for (int i = 0; i < k; ++i) {
        if (!use(i)) continue;
        const double dy = yv(i) - ym;
        for (int f = 0; f < F; ++f) {
            const double df = xv(i, f) - xm[f];
            for (int g = f; g < F; ++g) A[f][g] += df * (xv(i, g) - xm[g]);
            A[f][F] += df * dy;
        }
    }
    for (int f = 0; f < F; ++f)
        for (int g = 0; g < f; ++g) A[f][g] = A[g][f];
    sdlsq::minnorm_solve(F, A, coef);  // like LinearRegression's lstsq (gard.py:215-217)
    double icpt = ym;
    for (int f = 0; f < F; ++f) icpt -= xm[f] * coef[f];
    double p = icpt;
    for (int f = 0; f < F; ++f) p += q[f] * coef[f];
    double ss = 0.0;
    for (int i = 0; i < k; ++i) {
        if (!use(i)) continue;
        double yh = icpt;
        for (int f = 0; f < F; ++f) yh += xv(i, f) * coef[f];
        const double d = yv(i) - yh;
        ss += d * d;
    }
    *pred = p;
    *err = sqrt(ss / (double)ne);  // root_mean_squared_error (gard.py:218-219)
}

// LogisticRegression() of sklearn (L2 penalty, C = 1, intercept not penalised; gard.py:177, 204-212) on the k analogs of a
// query: labels t_i = (y_i > thresh), both classes present.  Exact minimiser of
//     sum_i [log(1 + exp(z_i)) - t_i z_i] + |w|^2 / 2,   z_i = w . x_i + b
// by damped Newton steps (Cholesky of the (F+1) x (F+1) Hessian, step halved until the objective does not increase);
// sklearn stops its L-BFGS at a gradient of 1e-4 of the mean loss, i.e. within ~1e-3 of this optimum.  Returns z(q).
template <typename XV, typename TV>
__device__ double logistic_at_query(int k, int F, XV xv /* (i,f) */, TV tv /* (i) -> 0/1 */, const double* q) {
    const int n = F + 1;
    double th[kMaxF + 1], g[kMaxF + 1], d[kMaxF + 1], trial[kMaxF + 1], H[kMaxF + 1][kMaxF + 1];
    for (int a = 0; a < n; ++a) th[a] = 0.0;
    auto objective = [&](const double* t) {
        double f = 0.0;
        for (int i = 0; i < k; ++i) {
            double z = t[F];
            for (int a = 0; a < F; ++a) z += t[a] * xv(i, a);
            f += sdlsq::softplus(z) - (tv(i) ? z : 0.0);
        }
        for (int a = 0; a < F; ++a) f += 0.5 * t[a] * t[a];
        return f;
    };
    double f = objective(th);
    for (int it = 0; it < 60; ++it) {
        for (int a = 0; a < n; ++a) {
            g[a] = a < F ? th[a] : 0.0;
            for (int b = 0; b < n; ++b) H[a][b] = (a == b && a < F) ? 1.0 : 0.0;
        }
        for (int i = 0; i < k; ++i) {
            double z = th[F];
            for (int a = 0; a < F; ++a) z += th[a] * xv(i, a);
            const double sg = sdlsq::sigmoid(z), r = sg - (tv(i) ? 1.0 : 0.0), w = sg * (1.0 - sg);
            for (int a = 0; a < n; ++a) {
                const double xa = a < F ? xv(i, a) : 1.0;
                g[a] += r * xa;
                for (int b = 0; b <= a; ++b) H[a][b] += w * xa * (b < F ? xv(i, b) : 1.0);
            }
        }
        double gmax = 0.0;
        for (int a = 0; a < n; ++a) gmax = fmax(gmax, fabs(g[a]));
        if (gmax <= 1e-12 * (double)k) break;
        for (int a = 0; a < n; ++a) {
            H[a][a] += 1e-12;
            g[a] = -g[a];
        }
        if (!sdlsq::chol_solve(n, H, g, d)) break;
        double step = 1.0, fn = f;
        for (;;) {
            for (int a = 0; a < n; ++a) trial[a] = th[a] + step * d[a];
            fn = objective(trial);
            if (fn <= f || step < 1e-10) break;
            step *= 0.5;
        }
        for (int a = 0; a < n; ++a) th[a] = trial[a];
        f = fn;
    }
    double z = th[F];
    for (int a = 0; a < F; ++a) z += th[a] * q[a];
    return z;
}

// mode 0 = PureAnalog, 1 = AnalogRegression.  Lists in scratch: sd[i*nthr + tid], si[...].
template <typename IT>
__device__ void finish_query(int mode, const PredictArgs& pa, int F, int64_t T, int64_t c, int64_t tq, const double* q,
                             const double* __restrict__ Xc_cell, const double* __restrict__ yc_cell,
                             const double* sd, const IT* si, int nthr, bool cell_active) {
    const int tid = threadIdx.x;
    const int k = pa.k;
    double pred, prob = 1.0, err;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (!cell_active) {
        pred = prob = err = nan;
    } else if (mode == 0) {
        const int s = (pa.kind == SD_ANALOG_SAMPLE && pa.sample) ? pa.sample[tq * pa.ld_s + c] : 0;
        pure_analog_stats(
            pa, k, pa.kind, s < 0 ? 0 : (s >= k ? k - 1 : s), [&](int i) { return yc_cell[si[(int64_t)i * nthr + tid]]; },
            [&](int i) { return sd[(int64_t)i * nthr + tid]; }, &pred, &prob, &err);
    } else {
        auto xv = [&](int i, int f) { return Xc_cell[(int64_t)f * T + si[(int64_t)i * nthr + tid]]; };
        auto yv = [&](int i) { return yc_cell[si[(int64_t)i * nthr + tid]]; };
        if (pa.has_thresh) {  // gard.py:201-219
            auto exc = [&](int i) { return yv(i) > pa.thresh; };
            int ne = 0;
            for (int i = 0; i < k; ++i) ne += exc(i) ? 1 : 0;
            if (ne == 0) {
                // every analog at or below the threshold: the reference's LogisticRegression.fit raises (one class only)
                if (pa.one_class) atomicOr(&pa.one_class[c], SDI_ONE_CLASS);
                pred = prob = err = nan;
            } else {
                if (ne < k) prob = 1.0 - sdlsq::sigmoid(logistic_at_query(k, F, xv, exc, q));  // predict_proba(X)[0, 0] (gard.py:210)
                analog_regression(k, F, xv, yv, exc, ne, q, &pred, &err);
            }
        } else {
            analog_regression(k, F, xv, yv, [](int) { return true; }, k, q, &pred, &err);
        }
    }
    put_out(pa, tq, c, pred, prob, err);
    if (cell_active && pa.inds)
        for (int i = 0; i < k; ++i) pa.inds[(tq * k + i) * pa.ld_out + c] = si[(int64_t)i * nthr + tid];
    if (cell_active && pa.dist)
        for (int i = 0; i < k; ++i) pa.dist[(tq * k + i) * pa.ld_out + c] = sqrt(sd[(int64_t)i * nthr + tid]);
}

// ------------------------------------------------------------------------------------------------
// F == 1 predict: sorted training values in LDS, binary search + two-pointer walk
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) analog_f1_predict_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                 int64_t Tq, int64_t T, int64_t C,
                                                                 const double* __restrict__ xs_all,
                                                                 const int32_t* __restrict__ xi_all,
                                                                 const double* __restrict__ Xc,
                                                                 const double* __restrict__ yc,
                                                                 const int32_t* __restrict__ fit_status, int32_t* status,
                                                                 double* scratch_d, int32_t* scratch_i, PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int n = (int)T, k = pa.k;
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    int64_t step, end;
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const int32_t* xi = xi_all + c * T;
        __syncthreads();
        if (active)
            for (int i = tid; i < n; i += nthr) xs[i] = xs_all[c * T + i];
        __syncthreads();
        for (int64_t tq = tid; tq < Tq; tq += nthr) {
            const double q = Xq[tq * ld + c];
            bool ok = active;
            if (active && !sd_finite(q)) {
                atomicOr(&status[c], SDI_NONFINITE);
                ok = false;
            }
            if (ok) {
                // r = first sorted position with x > q ; left part ends at r - 1
                int lo = 0, hi = n;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (xs[mid] <= q) lo = mid + 1; else hi = mid;
                }
                int r = lo;          // next right candidate
                int le = lo - 1;     // last element of the current left run (-1: exhausted)
                int rs = 0, cur = 0; // current left run [rs, le], next to take = cur (ascending index order)
                if (le >= 0) {
                    rs = le;
                    while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
                    cur = rs;
                }
                for (int i = 0; i < k; ++i) {
                    double dl = 0.0, dr = 0.0;
                    const bool hl = le >= 0, hr = r < n;
                    if (hl) { const double d = q - xs[le]; dl = d * d; }
                    if (hr) { const double d = q - xs[r]; dr = d * d; }
                    bool take_left;
                    if (hl && hr) take_left = dl < dr || (dl == dr && xi[cur] < xi[r]);
                    else take_left = hl;
                    if (take_left) {
                        sd[(int64_t)i * nthr + tid] = dl;
                        si[(int64_t)i * nthr + tid] = xi[cur];
                        if (++cur > le) {
                            le = rs - 1;
                            if (le >= 0) {
                                rs = le;
                                while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
                                cur = rs;
                            }
                        }
                    } else {
                        sd[(int64_t)i * nthr + tid] = dr;
                        si[(int64_t)i * nthr + tid] = xi[r];
                        ++r;
                    }
                }
            }
            finish_query(mode, pa, 1, T, c, tq, &q, Xc + c * T, yc + c * T, sd, si, nthr, ok);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// F == 1 predict, window form.  In one dimension the k nearest training values are k consecutive
// entries of the sorted view unless a tie sits on the boundary, so a query costs one binary search
// for the window start (2 LDS reads per step) and one pass over yx[L .. L+k) -- the analog values
// in sorted-x order, k consecutive doubles.  No neighbour lists, no gathers.  Used for PureAnalog
// kinds best / weight / mean when neither indices nor distances are requested; a query whose window
// is not strictly separated from its outside neighbours (exact distance ties, tie runs cut by the
// left boundary: KDTree order then depends on the training index) is answered by the exact
// (rdist, index)-ordered walk below, as are 'sample_analogs' and AnalogRegression.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sq_dist(double q, double x) {
    const double d = q - x;
    return d * d;
}

__device__ void f1_walk_query(int mode, const PredictArgs& pa, int n, int64_t T, int64_t c, int64_t tq, double q,
                              const double* xs /* LDS */, const int32_t* __restrict__ xi, const double* __restrict__ Xc_cell,
                              const double* __restrict__ yc_cell, double* sd, int32_t* si, int nthr) {
    const int tid = threadIdx.x, k = pa.k;
    // r = first sorted position with x > q ; left part ends at r - 1
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] <= q) lo = mid + 1; else hi = mid;
    }
    int r = lo;          // next right candidate
    int le = lo - 1;     // last element of the current left run (-1: exhausted)
    int rs = 0, cur = 0; // current left run [rs, le], next to take = cur (ascending index order)
    if (le >= 0) {
        rs = le;
        while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
        cur = rs;
    }
    for (int i = 0; i < k; ++i) {
        double dl = 0.0, dr = 0.0;
        const bool hl = le >= 0, hr = r < n;
        if (hl) dl = sq_dist(q, xs[le]);
        if (hr) dr = sq_dist(q, xs[r]);
        bool take_left;
        if (hl && hr) take_left = dl < dr || (dl == dr && xi[cur] < xi[r]);
        else take_left = hl;
        if (take_left) {
            sd[(int64_t)i * nthr + tid] = dl;
            si[(int64_t)i * nthr + tid] = xi[cur];
            if (++cur > le) {
                le = rs - 1;
                if (le >= 0) {
                    rs = le;
                    while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
                    cur = rs;
                }
            }
        } else {
            sd[(int64_t)i * nthr + tid] = dr;
            si[(int64_t)i * nthr + tid] = xi[r];
            ++r;
        }
    }
    finish_query(mode, pa, 1, T, c, tq, &q, Xc_cell, yc_cell, sd, si, nthr, true);
}

constexpr int kWinQ = 2;      // queries a thread answers together (independent dependency chains)
constexpr int kWinBatch = 8;  // analog values read together per query

// The sorted view of a cell is processed in `npass` value ranges so that both xs and yx of a range (plus k
// entries of margin on either side) sit in LDS: the window search and the k analog values of a query are LDS
// reads, HBM/L2 only see the query and the three outputs.  A query belongs to the range that holds its value;
// its k nearest neighbours are at most k positions away from there.
__global__ void __launch_bounds__(1024) analog_f1_window_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                int64_t Tq, int64_t T, int64_t C, int npass,
                                                                const double* __restrict__ xs_all,
                                                                const int32_t* __restrict__ xi_all,
                                                                const double* __restrict__ yx_all,
                                                                const double* __restrict__ Xc, const double* __restrict__ yc,
                                                                const int32_t* __restrict__ fit_status, int32_t* status,
                                                                double* scratch_d, int32_t* scratch_i, PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int n = (int)T, k = pa.k;
    const int seg = (n + npass - 1) / npass;
    const int cap = seg + 2 * k + 1;                        // local entries per pass (upper bound)
    double* xs = reinterpret_cast<double*>(smem_raw);       // cap + 1 doubles (sentinel)
    double* yl = xs + cap + 1;                              // cap doubles
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    int64_t step, end;
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const int32_t* xi = xi_all + c * T;
        const double* xg = xs_all + c * T;
        const double* yx = yx_all + c * T;
        for (int p = 0; p < npass; ++p) {
            const int b0 = p * seg < n ? p * seg : n, b1 = (p + 1) * seg < n ? (p + 1) * seg : n;
            if (b0 >= b1) break;
            const int g0 = b0 - k > 0 ? b0 - k : 0, g1 = b1 + k < n ? b1 + k : n;  // global range held in LDS
            const int nl = g1 - g0;
            // value range of this pass: [vlo, vhi), open-ended at the ends of the sorted view
            const double vlo = (p == 0 || !active) ? -inf : xg[b0];
            const double vhi = (b1 >= n || !active) ? inf : xg[b1];
            __syncthreads();
            if (active)
                for (int i = tid; i < nl; i += nthr) {
                    xs[i] = xg[g0 + i];
                    yl[i] = yx[g0 + i];
                }
            if (tid == 0) xs[nl] = inf;
            __syncthreads();
            int nsteps = 0;  // fixed trip count of the window search: every lane and query runs the same loop
            while ((1 << nsteps) < nl - k + 1) ++nsteps;
            for (int64_t tq0 = tid; tq0 < Tq; tq0 += (int64_t)nthr * kWinQ) {
                double q[kWinQ];
                bool has[kWinQ], ok[kWinQ], mine[kWinQ];
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    const int64_t tq = tq0 + (int64_t)j * nthr;
                    has[j] = tq < Tq;
                    q[j] = has[j] ? Xq[c * ld + tq] : 0.0;  // cell-major copy of the queries (ld = Tq)
                }
                bool any = false;
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    ok[j] = active && has[j] && sd_finite(q[j]);
                    // inactive cells and non-finite queries are reported (NaN outputs) in the first pass
                    mine[j] = has[j] && (ok[j] ? (q[j] >= vlo && q[j] < vhi) || (q[j] == inf) : p == 0);
                    if (mine[j] && active && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                    if (!ok[j]) q[j] = 0.0;
                    any |= mine[j];
                }
                if (!any) continue;
                // window start: smallest L with rdist(L) <= rdist(L + k) (rdist is unimodal along the sorted view)
                int lo[kWinQ], hi[kWinQ];
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    lo[j] = 0;
                    hi[j] = nl - k;
                }
#pragma unroll 1
                for (int s = 0; s < nsteps; ++s) {
#pragma unroll
                    for (int j = 0; j < kWinQ; ++j) {
                        const int mid = (lo[j] + hi[j]) >> 1;
                        const bool act = lo[j] < hi[j];
                        const bool right = sq_dist(q[j], xs[mid]) > sq_dist(q[j], xs[mid + k]);
                        lo[j] = (act && right) ? mid + 1 : lo[j];
                        hi[j] = (act && !right) ? mid : hi[j];
                    }
                }
                bool unique[kWinQ];
                // sums over the window, shifted by its first element (x0, a0) so that no cancellation occurs;
                // PureAnalog: s1 = sum(a), s2 = sum(a^2), weights; AnalogRegression: wsum/awsum/sxx hold sum(x), sum(x*a), sum(x^2)
                double x0[kWinQ], a0[kWinQ], s1[kWinQ], s2[kWinQ], wsum[kWinQ], awsum[kWinQ], sxx[kWinQ];
                int nexc[kWinQ];
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    const int L = lo[j];
                    const double dL = sq_dist(q[j], xs[L]), dR = sq_dist(q[j], xs[L + k - 1]);
                    const double worst = dL > dR ? dL : dR;
                    // the outside neighbours must be strictly farther; at an edge of the LDS range that is not an
                    // edge of the sorted view the neighbour is unknown -> exact walk
                    const bool sep_l = L == 0 ? g0 == 0 : sq_dist(q[j], xs[L - 1]) > worst;
                    const bool sep_r = L + k == nl ? g1 == n : sq_dist(q[j], xs[L + k]) > worst;
                    unique[j] = sep_l && sep_r;
                    s1[j] = s2[j] = wsum[j] = awsum[j] = sxx[j] = 0.0;
                    nexc[j] = 0;
                    x0[j] = xs[L];
                    a0[j] = yl[L];
                }
                const bool need_x = mode == 1 || pa.kind == SD_ANALOG_WEIGHT;
                for (int i0 = 0; i0 < k; i0 += kWinBatch) {
#pragma unroll
                    for (int j = 0; j < kWinQ; ++j)
#pragma unroll
                        for (int b = 0; b < kWinBatch; ++b) {
                            const int i = i0 + b;
                            if (i < k) {
                                const double ai = yl[lo[j] + i];
                                const double e = ai - a0[j];
                                s1[j] += e;
                                s2[j] += e * e;
                                nexc[j] += (!pa.has_thresh || ai > pa.thresh) ? 1 : 0;  // gard.py:307
                                if (need_x) {
                                    const double xv = xs[lo[j] + i];
                                    if (mode == 1) {
                                        const double dx = xv - x0[j];
                                        wsum[j] += dx;
                                        awsum[j] += dx * e;
                                        sxx[j] += dx * dx;
                                    } else {
                                        // w = 1 / distance (gard.py:322-323); sqrt((q-x)^2) == |q-x| in IEEE arithmetic.
                                        // Reciprocal by v_rcp_f64 + two Newton steps (< 1 ulp; the tolerance is 1e-6).
                                        double d = __builtin_fabs(q[j] - xv);
                                        d = d == 0.0 ? 1e-20 : d;
                                        double r = __builtin_amdgcn_rcp(d);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                                        wsum[j] += r;
                                        awsum[j] += ai * r;
                                    }
                                }
                            }
                        }
                }
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    if (!mine[j]) continue;
                    const int64_t tq = tq0 + (int64_t)j * nthr;
                    double pred = nan, prob = nan, err = nan;
                    if (ok[j]) {
                        const int L = lo[j];
                        double best_a = a0[j];
                        if (mode == 0 && pa.kind == SD_ANALOG_BEST && unique[j]) {
                            // nearest element = one of the two around the insertion point of q inside the window;
                            // equal distances or equal values there leave the choice to the training index -> walk
                            int m = 0;  // first window entry with x >= q
                            for (int len = k; len > 0;) {
                                const int half = len >> 1;
                                if (xs[L + m + half] < q[j]) { m += half + 1; len -= half + 1; } else len = half;
                            }
                            const double dl = m > 0 ? sq_dist(q[j], xs[L + m - 1]) : inf;
                            const double dr = m < k ? sq_dist(q[j], xs[L + m]) : inf;
                            if (dl == dr) unique[j] = false;
                            const int b = dl < dr ? m - 1 : m;
                            if (dl < dr ? (b > 0 && xs[L + b - 1] == xs[L + b]) : (b + 1 < k && xs[L + b + 1] == xs[L + b]))
                                unique[j] = false;
                            best_a = yl[L + (b < k ? b : k - 1)];
                        }
                        if (!unique[j]) {
                            f1_walk_query(mode, pa, n, T, c, tq, q[j], xg, xi, Xc + c * T, yc + c * T, sd, si, nthr);
                            continue;
                        }
                        const bool any_masked = nexc[j] != k;
                        const double kk = (double)k;
                        const double m1 = s1[j] / kk;
                        if (mode == 1) {
                            // one-feature OLS on the k analogs (gard.py:194-224): centred sums, slope 0 when all x are equal
                            const double mx = wsum[j] / kk;
                            const double vxx = sxx[j] - kk * mx * mx, vxy = awsum[j] - kk * mx * m1;
                            const double slope = vxx > 0.0 ? vxy / vxx : 0.0;
                            const double xm = x0[j] + mx, ym = a0[j] + m1;
                            const double icpt = ym - xm * slope;
                            pred = icpt + q[j] * slope;
                            double ss = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double r = yl[L + i] - (icpt + xs[L + i] * slope);
                                ss += r * r;
                            }
                            prob = 1.0;
                            err = sqrt(ss / kk);  // root_mean_squared_error (gard.py:218-219)
                        } else {
                            if (pa.kind == SD_ANALOG_BEST) pred = best_a;                                         // gard.py:311
                            else if (pa.kind == SD_ANALOG_WEIGHT) pred = any_masked ? nan : awsum[j] / wsum[j];   // gard.py:319-327
                            else pred = any_masked ? nan : a0[j] + m1;                                            // gard.py:329-333
                            if (pa.has_thresh) {
                                pred = nan_to_num(pred);      // gard.py:341
                                prob = (double)nexc[j] / kk;   // gard.py:343
                            } else {
                                prob = 1.0;  // gard.py:346
                            }
                            if (!any_masked) {
                                const double var = s2[j] / kk - m1 * m1;
                                err = sqrt(var > 0.0 ? var : 0.0);  // ddof = 0 (gard.py:342,345)
                            }
                        }
                    }
                    put_out(pa, tq, c, pred, prob, err);
                }
            }
        }
    }
}

// F == 1, single pass over the queries with only the sorted training values LDS-resident.  'mean_analogs' without a
// threshold, a single analog and AnalogRegression (mode 1, k >= 3) take the window statistics from the prefix sums
// pq / rx (analog_prefix_kernel): the window search plus two (regression: three) pairs of prefix loads per query.
// 'weight_analogs' and the thresholded kinds read the k consecutive analog values of the window from memory.
// Tie handling as in analog_f1_window_kernel.
__global__ void __launch_bounds__(1024) analog_f1_mean_kernel(int mode, const double* __restrict__ Xq /* [C][Tq] */, int64_t Tq,
                                                              int64_t T, int64_t C, const double* __restrict__ xs_all,
                                                              const int32_t* __restrict__ xi_all,
                                                              const double* __restrict__ pq_all,
                                                              const double* __restrict__ ybar_all,
                                                              const double* __restrict__ rx_all,
                                                              const double* __restrict__ xbar_all,
                                                              const double* __restrict__ yx_all, const double* __restrict__ Xc,
                                                              const double* __restrict__ yc,
                                                              const int32_t* __restrict__ fit_status, int32_t* status,
                                                              double* scratch_d, int32_t* scratch_i, PredictArgs pa, int qsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);  // n sorted values + one +inf sentinel
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int n = (int)T, k = pa.k;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    int nsteps = 0;  // window refinement: the range p - k .. p holds at most k + 1 candidates
    while ((1 << nsteps) < (k + 1 < n - k + 1 ? k + 1 : n - k + 1)) ++nsteps;
    const double kk = (double)k;
    // qsplit workgroups of one XCD share a cell (each answers 1/qsplit of its queries), so that the XCD works on
    // fewer cells at a time and the prefix sums of those cells (read at two random places per query) stay in its L2.
    // Pays for the regression (three prefix arrays: pq, rx), not for the plain mean (the extra LDS fills cost more)
    int64_t step, end, c0;
    int part = 0;
    if (qsplit > 1) {  // (the launcher guarantees gridDim.x % (8 * qsplit) == 0)
        const int64_t cx = (C + 7) / 8;
        const int x = blockIdx.x % 8, j = blockIdx.x / 8;
        part = j % qsplit;
        step = gridDim.x / 8 / qsplit;
        end = (x + 1) * cx < C ? (x + 1) * cx : C;
        c0 = x * cx + j / qsplit;
    } else {
        c0 = first_cell(C, &step, &end);
    }
    const int64_t qchunk = (Tq + qsplit - 1) / qsplit, q_beg = part * qchunk, q_end = q_beg + qchunk < Tq ? q_beg + qchunk : Tq;
    for (int64_t c = c0; c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const double* xg = xs_all + c * T;
        const double2* pq = reinterpret_cast<const double2*>(pq_all) + c * (T + 1);
        const double ybar = ybar_all[c];
        const double* rx = rx_all + c * (T + 1);
        const double xbar = mode == 1 ? xbar_all[c] : 0.0;
        // AnalogRegression: residual sums below this are left to direct summation (the prefix differences carry an
        // absolute error of ~1e-16 of the cell total)
        const double ss_floor = mode == 1 ? 1e-4 * kk * (pq[n].y / (double)n) : 0.0;
        __syncthreads();
        if (active)
            for (int i = tid; i < n; i += nthr) xs[i] = xg[i];
        if (tid == 0) xs[n] = inf;
        __syncthreads();
        for (int64_t tq0 = q_beg + tid; tq0 < q_end; tq0 += (int64_t)nthr * kWinQ) {
            double q[kWinQ];
            bool has[kWinQ], ok[kWinQ];
#pragma unroll
            for (int j = 0; j < kWinQ; ++j) {
                const int64_t tq = tq0 + (int64_t)j * nthr;
                has[j] = tq < q_end;
                q[j] = has[j] ? Xq[c * Tq + tq] : 0.0;
                ok[j] = active && has[j] && sd_finite(q[j]);
                if (active && has[j] && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                if (!ok[j]) q[j] = 0.0;
            }
            // p = number of training values < q (one LDS read per step; strides that are multiples of 16 doubles are
            // shortened by one, see the rank search in sd_bcsd_rs.hip).  Without ties the k nearest values are a window
            // [L, L + k) with p - k <= L <= p: the smallest L of that range with rdist(L) <= rdist(L + k), log2(k + 1)
            // more steps of two reads (rdist is unimodal along xs).  With ties the separation test below sends the query
            // to the exact walk.
            int lo[kWinQ], hi[kWinQ];
            {
                int pos[kWinQ];
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) pos[j] = -1;  // index of the last value known to be < q
#pragma unroll 1
                for (int len = n; len > 1;) {
                    int half = len >> 1;
                    if ((half & 15) == 0) --half;
                    len -= half;
#pragma unroll
                    for (int j = 0; j < kWinQ; ++j) pos[j] += xs[pos[j] + half] < q[j] ? half : 0;
                }
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    const int p = pos[j] + 1 + (xs[pos[j] + 1] < q[j] ? 1 : 0);
                    lo[j] = p - k > 0 ? p - k : 0;
                    hi[j] = p < n - k ? p : n - k;
                }
            }
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    const int mid = (lo[j] + hi[j]) >> 1;
                    const bool act = lo[j] < hi[j];
                    const bool right = sq_dist(q[j], xs[mid]) > sq_dist(q[j], xs[mid + k]);
                    lo[j] = (act && right) ? mid + 1 : lo[j];
                    hi[j] = (act && !right) ? mid : hi[j];
                }
            }
#pragma unroll
            for (int j = 0; j < kWinQ; ++j) {
                if (!has[j]) continue;
                const int64_t tq = tq0 + (int64_t)j * nthr;
                double pred = nan, prob = nan, err = nan;
                if (ok[j]) {
                    const int L = lo[j];
                    const double dL = sq_dist(q[j], xs[L]), dR = sq_dist(q[j], xs[L + k - 1]);
                    const double worst = dL > dR ? dL : dR;
                    const bool sep_l = L == 0 || sq_dist(q[j], xs[L - 1]) > worst;
                    const bool sep_r = L + k == n || sq_dist(q[j], xs[L + k]) > worst;
                    if (!(sep_l && sep_r)) {
                        f1_walk_query(mode, pa, n, T, c, tq, q[j], xg, xi_all + c * T, Xc + c * T, yc + c * T, sd, si, nthr);
                        continue;
                    }
                    if (mode == 1) {
                        // one-feature OLS on the k analogs (gard.py:194-224), slope 0 when all x are equal.  The x sums
                        // come from the LDS window, the y and cross sums from the prefix differences:
                        //   sum (x - xm)(y - ym) = [rx] + (xbar - xm) [p],  sum (y - ym)^2 = [q] - k m1^2
                        const double x0 = xs[L];
                        double sx = 0.0, sxx = 0.0;
                        for (int i = 0; i < k; ++i) {
                            const double dx = xs[L + i] - x0;
                            sx += dx;
                            sxx += dx * dx;
                        }
                        const double2 a = pq[L], b = pq[L + k];
                        const double s1 = b.x - a.x, m1 = s1 / kk, mx = sx / kk, xm = x0 + mx;
                        const double vxx = sxx - kk * mx * mx, vyy = (b.y - a.y) - kk * m1 * m1;
                        const double vxy = (rx[L + k] - rx[L]) + (xbar - xm) * s1;
                        const double slope = vxx > 0.0 ? vxy / vxx : 0.0;
                        double ss = vyy - slope * vxy;
                        pred = (ybar + m1) + (q[j] - xm) * slope;
                        if (!(ss > ss_floor)) {
                            // (nearly) exact fit or constant analogs: the sums directly, as analog_f1_window_kernel
                            const double* yl = yx_all + c * T + L;
                            const double a0 = yl[0];
                            double t1 = 0.0, txy = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double e = yl[i] - a0;
                                t1 += e;
                                txy += (xs[L + i] - x0) * e;
                            }
                            const double n1 = t1 / kk;
                            const double wxy = txy - kk * mx * n1;
                            const double sl = vxx > 0.0 ? wxy / vxx : 0.0;
                            const double icpt = (a0 + n1) - xm * sl;
                            pred = icpt + q[j] * sl;
                            ss = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double r = yl[i] - (icpt + xs[L + i] * sl);
                                ss += r * r;
                            }
                        }
                        prob = 1.0;
                        err = sqrt(ss / kk);  // root_mean_squared_error (gard.py:218-219)
                    } else if (k == 1) {
                        // a single analog (best_analog, or n_analogs = 1: gard.py:291-296): the value itself, no spread
                        const double a1 = yx_all[c * T + L];
                        const bool exc = !pa.has_thresh || a1 > pa.thresh;  // gard.py:307
                        pred = (pa.kind == SD_ANALOG_BEST || exc) ? a1 : 0.0;  // mean / weight of a masked analog: NaN -> 0 (gard.py:341)
                        prob = pa.has_thresh ? (exc ? 1.0 : 0.0) : 1.0;       // gard.py:343, 346
                        err = exc ? 0.0 : nan;                                // gard.py:342, 345
                    } else if (pa.kind == SD_ANALOG_MEAN && !pa.has_thresh) {
                        const double2 a = pq[L], b = pq[L + k];
                        const double m1 = (b.x - a.x) / kk;           // mean of the centred analogs
                        const double var = (b.y - a.y) / kk - m1 * m1;
                        pred = ybar + m1;                            // gard.py:329-333
                        prob = 1.0;                                  // gard.py:346
                        err = sqrt(var > 0.0 ? var : 0.0);           // ddof = 0 (gard.py:345)
                    } else {
                        // weights and / or a threshold need every analog: the window of yx is read from memory (k
                        // consecutive values, cache-resident), the training values come from LDS
                        const double* yl = yx_all + c * T + L;
                        const double a0 = yl[0];
                        double s1 = 0.0, s2 = 0.0, wsum = 0.0, awsum = 0.0;
                        int nexc = 0;
                        for (int i0 = 0; i0 < k; i0 += kWinBatch) {
                            double ab[kWinBatch];
#pragma unroll
                            for (int b = 0; b < kWinBatch; ++b) ab[b] = i0 + b < k ? yl[i0 + b] : 0.0;
#pragma unroll
                            for (int b = 0; b < kWinBatch; ++b) {
                                const int i = i0 + b;
                                if (i < k) {
                                    const double ai = ab[b], e = ai - a0;
                                    s1 += e;
                                    s2 += e * e;
                                    nexc += (!pa.has_thresh || ai > pa.thresh) ? 1 : 0;  // gard.py:307
                                    if (pa.kind == SD_ANALOG_WEIGHT) {
                                        // w = 1 / distance (gard.py:322-323): v_rcp_f64 + two Newton steps (< 1 ulp)
                                        double d = __builtin_fabs(q[j] - xs[L + i]);
                                        d = d == 0.0 ? 1e-20 : d;
                                        double r = __builtin_amdgcn_rcp(d);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                                        wsum += r;
                                        awsum += ai * r;
                                    }
                                }
                            }
                        }
                        const bool any_masked = nexc != k;
                        const double m1 = s1 / kk;
                        if (pa.kind == SD_ANALOG_WEIGHT) pred = any_masked ? nan : awsum / wsum;  // gard.py:319-327
                        else pred = any_masked ? nan : a0 + m1;                                    // gard.py:329-333
                        if (pa.has_thresh) {
                            pred = nan_to_num(pred);     // gard.py:341
                            prob = (double)nexc / kk;    // gard.py:343
                        } else {
                            prob = 1.0;  // gard.py:346
                        }
                        if (!any_masked) {
                            const double var = s2 / kk - m1 * m1;
                            err = sqrt(var > 0.0 ? var : 0.0);  // ddof = 0 (gard.py:342,345)
                        }
                    }
                }
                put_out(pa, tq, c, pred, prob, err);
            }
        }
    }
}

// F == 1, PureAnalog 'mean_analogs' without a threshold (the BASELINE configuration) or a single analog.  One 1024-thread
// workgroup per cell; a thread keeps the window starts of its (up to kPhQ) queries in registers and the LDS array
// (n + 1 doubles: all the LDS a workgroup can have at the BASELINE length) is filled three times per cell:
//   1. sorted training values: every query finds its window of k nearest values (two branch-free bisections: position
//      among the values, then window start among the k + 1 candidates), windows with a tie on their boundary take the
//      exact walk;
//   2. exclusive prefix sums of the centred analog values d = yx - mean(y), computed here from yx (blocked partial sums,
//      wave scans) -> window means;
//   3. exclusive prefix sums of d^2 -> spreads.
// Every byte of the state is fetched once, coalesced: xs and yx (8 + 8 bytes per training sample; the fitted state holds
// no prefix sums for this path); every fill keeps all of a thread's loads in flight together (register staging).  The
// workgroup is alone on its CU (LDS), so its memory phases and its LDS phases do not overlap; warming L2 for the next
// phase with early one-word-per-line loads was tried and made the kernel 13 % slower (the lines are gone again before the
// fill: 32 workgroups per XCD stream ~11 MB through a 4 MB L2) and doubled its counted fetch traffic.
// With skip_prob the exceedance-probability column is not written: it is 1 wherever the prediction is not NaN
// (gard.py:346) and the staging transpose fills it in.
constexpr int kPhQ = 16;  // queries per thread and LDS generation (1024 threads: series up to 16 384 queries per pass)

// a wave-uniform double, pinned to scalar registers (the allocator otherwise keeps such values in vector registers and,
// in this kernel, spills them)
__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
typedef __attribute__((address_space(1))) double global_f64;  // (pointers that travel inside PredictArgs are generic otherwise)

// x / k for an integer-valued k with rk = RN(1 / k): quotient estimate, exact remainder, one correction (correctly rounded;
// the hardware-assisted IEEE division costs ~10x as many instructions and this kernel needs two per query)
__device__ __forceinline__ double div_by(double x, double kk, double rk) {
    const double q = x * rk;
    const double r = __builtin_fma(-q, kk, x);
    return __builtin_fma(r, rk, q);
}

template <int PER>
__global__ void __launch_bounds__(1024) analog_f1_mean3_kernel(const double* __restrict__ Xq /* [C][Tq] */, int64_t Tq, int64_t T,
                                                               int64_t C, const double* __restrict__ xs_all,
                                                               const int32_t* __restrict__ xi_all,
                                                               const double* __restrict__ ybar_all,
                                                               const double* __restrict__ yx_all, const double* __restrict__ Xc,
                                                               const double* __restrict__ yc,
                                                               const int32_t* __restrict__ fit_status, int32_t* status,
                                                               double* scratch_d, int32_t* scratch_i, PredictArgs pa, int skip_prob,
                                                               long long* trace /* development library: phase clocks of block 0 */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ double wsum[16];
    double* buf = reinterpret_cast<double*>(smem_raw);  // n + 1 doubles
    const int nthr = blockDim.x;
    const int n = (int)T, k = pa.k;
    const int per = (n + nthr - 1) / nthr;  // consecutive samples per thread in the prefix sums (<= PER)
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    // The thread id is re-read behind an opaque barrier in every phase: otherwise the compiler computes the dozens of
    // per-sample indices, predicates and LDS addresses of all phases once, ahead of the cell loop, and spills them.
#define SD_TID()                                        \
    int tid = (int)threadIdx.x;                         \
    asm volatile("" : "+v"(tid));                       \
    const int lane = tid & 63, wave = tid >> 6;         \
    (void)lane;                                         \
    (void)wave
    const double kk = uniform_f64((double)k), rk = uniform_f64(1.0 / (double)k);
    const int M = n - k > 0 ? n - k : 0;  // window starts 0 .. M
    int nsteps = 0;  // window refinement: the range p - k .. p holds at most k + 1 candidates
    while ((1 << nsteps) < (k + 1 < M + 1 ? k + 1 : M + 1)) ++nsteps;
    int64_t step, end;
#ifdef SD_DEV
    int traced = 0;
#define SD_STAMP(slot)                                                                                                              \
    do {                                                                                                                            \
        if (trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && traced < 8) trace[traced * 16 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SD_STAMP(slot) do { } while (0)
#endif
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const double* xg = xs_all + c * T;
        const double* yx = yx_all + c * T;
        const double ybar = uniform_f64(ybar_all[c]);
        for (int64_t q0 = 0; q0 < Tq; q0 += (int64_t)kPhQ * nthr) {
            // ---- generation 1: sorted training values -> window start of every query
            __syncthreads();
            SD_STAMP(0);
            {
                SD_TID();
                double xv[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    xv[i] = (active && j < n) ? xg[j] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    if (j < n) buf[j] = xv[i];
                }
                if (tid == 0) buf[n] = inf;
            }
            __syncthreads();
            SD_STAMP(1);
            SD_TID();
            // the queries
            const double* qrow = Xq + c * Tq + q0;  // (uniform) queries of this pass
            const int nq = (int)(Tq - q0 < (int64_t)kPhQ * nthr ? Tq - q0 : (int64_t)kPhQ * nthr);
            double qv[kPhQ];
            unsigned hasmask = 0u;
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) {
                const int j = tid + i * nthr;
                qv[i] = 0.0;
                if (j < nq) {
                    qv[i] = qrow[j];
                    hasmask |= 1u << i;
                }
            }
            SD_STAMP(2);
            unsigned Lw2[kPhQ / 2];  // window starts, two 16-bit values per word
#define SD_LW(i) ((int)(((i) & 1) ? (Lw2[(i) >> 1] >> 16) : (Lw2[(i) >> 1] & 0xffffu)))
            unsigned okmask = 0u, nanmask = 0u, walkmask = 0u;  // bit i: prefix-sum statistics / NaN output / exact walk
#pragma unroll
            for (int i0 = 0; i0 < kPhQ; i0 += 2) {
                double q[2];
                bool has[2], ok[2];
                int lo[2], hi[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    has[j] = (hasmask >> (i0 + j)) & 1u;
                    q[j] = qv[i0 + j];
                    ok[j] = active && has[j] && sd_finite(q[j]);
                    if (active && has[j] && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                    if (!ok[j]) q[j] = 0.0;
                }
                // position of the query among the sorted values (branch-free bisection; strides that are multiples of 16
                // doubles are shortened by one: see the rank search in sd_bcsd_rs.hip), then the start of the window of k
                // nearest values among the k + 1 candidates around it
                int pos[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) pos[j] = -1;  // index of the last value known to be < q
#pragma unroll 1
                for (int len = n; len > 1;) {
                    int half = len >> 1;
                    if ((half & 15) == 0) --half;
                    len -= half;
#pragma unroll
                    for (int j = 0; j < 2; ++j) pos[j] += buf[pos[j] + half] < q[j] ? half : 0;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int p = pos[j] + 1 + (buf[pos[j] + 1] < q[j] ? 1 : 0);
                    lo[j] = p - k > 0 ? p - k : 0;
                    hi[j] = p < M ? p : M;
                }
#pragma unroll 1
                for (int s = 0; s < nsteps; ++s) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int mid = (lo[j] + hi[j]) >> 1;
                        const bool act = lo[j] < hi[j];
                        const bool right = sq_dist(q[j], buf[mid]) > sq_dist(q[j], buf[mid + k]);
                        lo[j] = (act && right) ? mid + 1 : lo[j];
                        hi[j] = (act && !right) ? mid : hi[j];
                    }
                }
                Lw2[i0 >> 1] = (unsigned)lo[0] | ((unsigned)lo[1] << 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = i0 + j;
                    if (!has[j]) continue;
                    if (!ok[j]) {
                        nanmask |= 1u << i;
                        continue;
                    }
                    const int L = lo[j];
                    const double dL = sq_dist(q[j], buf[L]), dR = sq_dist(q[j], buf[L + k - 1]);
                    const double worst = dL > dR ? dL : dR;
                    const bool sep_l = L == 0 || sq_dist(q[j], buf[L - 1]) > worst;
                    const bool sep_r = L + k >= n || sq_dist(q[j], buf[L + k]) > worst;
                    if (sep_l && sep_r) okmask |= 1u << i;
                    else walkmask |= 1u << i;  // a tie on the window boundary (or a bracket that missed)
                }
            }
            SD_STAMP(3);
#pragma unroll 1
            for (int i = 0; walkmask >> i; ++i)  // exact (rdist, index)-ordered walk; writes its own output
                if ((walkmask >> i) & 1u) {
                    const int64_t tq = q0 + tid + (int64_t)i * nthr;
                    f1_walk_query(0, pa, n, T, c, tq, Xq[c * Tq + tq], xg, xi_all + c * T, Xc + c * T, yc + c * T, sd, si, nthr);
                }
            // ---- y in sorted-x order
            __syncthreads();
            SD_STAMP(4);
            {
                SD_TID();
                double yv[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    yv[i] = j < n ? yx[j] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    if (j < n) buf[j] = yv[i];
                }
            }
            __syncthreads();
            SD_STAMP(5);
            // (uniform) staging rows of this cell and pass: predictions, probabilities, spreads
            global_f64* const orow = (global_f64*)(pa.out + c * 3 * pa.oc_Tq + q0);
            global_f64* const prow = orow + pa.oc_Tq;
            global_f64* const erow = prow + pa.oc_Tq;
            if (k == 1) {
                // a single analog (best_analog, or n_analogs = 1: gard.py:291-296)
#pragma unroll
                for (int i = 0; i < kPhQ; ++i) {
                    const int idx = tid + i * nthr;
                    const bool okq = (okmask >> i) & 1u;
                    if (okq || ((nanmask >> i) & 1u)) {
                        const double a1 = buf[okq ? SD_LW(i) : 0];
                        const bool exc = !pa.has_thresh || a1 > pa.thresh;  // gard.py:307
                        orow[idx] = !okq ? nan : (pa.kind == SD_ANALOG_BEST || exc) ? a1 : 0.0;  // masked mean / weight: NaN -> 0 (gard.py:341)
                        prow[idx] = !okq ? nan : pa.has_thresh ? (exc ? 1.0 : 0.0) : 1.0;          // gard.py:343, 346
                        erow[idx] = !okq ? nan : exc ? 0.0 : nan;                                  // gard.py:342, 345
                    }
                }
                continue;
            }
            // ---- generation 2: exclusive prefix sums of d = yx - mean(y) -> window means
            const int beg = per * tid;
            double d[PER];
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = beg + i;
                d[i] = (i < per && j < n) ? buf[j] - ybar : 0.0;
                a += d[i];
                b += d[i] * d[i];
            }
            double ia = a, ib = b;  // inclusive scans inside the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
                if (lane >= o) {
                    ia += ta;
                    ib += tb;
                }
            }
            __syncthreads();  // every thread has read its block of yx
            if (lane == 63) wsum[wave] = ia;
            __syncthreads();
            double ra = ia - a;  // exclusive prefix at this thread's first sample
            for (int w = 0; w < wave; ++w) ra += wsum[w];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = beg + i;
                if (i < per && j <= n) buf[j] = ra;
                ra += d[i];
            }
            if (tid == nthr - 1 && beg + per == n) buf[n] = ra;  // n = nthr * per: no thread starts at position n
            __syncthreads();
            SD_STAMP(6);
            double m1[kPhQ];
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) m1[i] = (okmask >> i) & 1u ? div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) : 0.0;
            // ---- generation 3: exclusive prefix sums of d^2 -> spreads, outputs
            __syncthreads();
            if (lane == 63) wsum[wave] = ib;
            __syncthreads();
            double rb = ib - b;
            for (int w = 0; w < wave; ++w) rb += wsum[w];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = beg + i;
                if (i < per && j <= n) buf[j] = rb;
                rb += d[i] * d[i];
            }
            if (tid == nthr - 1 && beg + per == n) buf[n] = rb;
            __syncthreads();
            SD_STAMP(7);
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) {
                const int idx = tid + i * nthr;
                double pred, err;
                if ((okmask >> i) & 1u) {
                    const double var = div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) - m1[i] * m1[i];
                    pred = ybar + m1[i];                 // gard.py:329-333
                    err = sqrt(var > 0.0 ? var : 0.0);   // gard.py:345
                } else if ((nanmask >> i) & 1u) {
                    pred = err = nan;
                } else {
                    continue;
                }
                orow[idx] = pred;
                erow[idx] = err;
                if (!skip_prob) prow[idx] = pred != pred ? nan : 1.0;  // gard.py:346
            }
            SD_STAMP(8);
            SD_STAMP(9);
#ifdef SD_DEV
            ++traced;
#endif
        }
    }
#undef SD_STAMP
#undef SD_LW
#undef SD_TID
}

// ------------------------------------------------------------------------------------------------
// fit + predict of the BASELINE case in one kernel (sd_analog_fit_predict_dev: gard.py:58-87 with 273-364 on the same call)
// ------------------------------------------------------------------------------------------------
// The per-cell workgroup that has merged the sorted runs of analog_tile_sort_kernel answers the cell's queries before it
// leaves: the tail of analog_sort2_kernel<K, true> followed by analog_f1_mean3_kernel, with the sorted view handed over on
// chip.  No fitted state exists: xs / xi / yx (20 bytes written and 16 read back per training sample) never travel.  The
// tags of the sorted keys (= xi) are parked in LDS behind the key array; x and y are gathered through them twice, once into
// the sorted order of each.  Arithmetic, summation orders and the window search are those of the two kernels, so the result
// is bit-identical to fit -> predict.  Cells the fast paths cannot decide -- the tag pass fails (equal or nearly equal
// training values), or a query's window is not strictly separated (the exact walk: inlined here it costs the kernel 60 more
// spilled registers and 3 ms per 16 384 cells for a case continuous data never produces) -- are appended to `worklist`; the
// host answers them with the split path.  Pointers are relative to the chunk of cells of this
// launch, `cell0` is the grid index of its first cell.
template <int K>
__global__ void __launch_bounds__(1024) analog_f1_fused_kernel(const double* __restrict__ runs, int np,
                                                               const int32_t* __restrict__ odd_flags,
                                                               const double* __restrict__ Xc, const double* __restrict__ yc,
                                                               const double* __restrict__ Xq /* [C][Tq] */, int64_t Tq, int64_t T,
                                                               int64_t C, const int32_t* __restrict__ fit_status, int32_t* status,
                                                               int32_t* worklist, int32_t* work_count, int64_t cell0,
                                                               PredictArgs pa, int skip_prob) {
    constexpr int PER = K;  // consecutive samples per thread in the prefix sums: ceil(n / 1024) <= K
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ double wsum[16];
    double* buf = reinterpret_cast<double*>(smem_raw);          // np + 1 doubles
    int* xch = reinterpret_cast<int*>(buf + np + 1);            // 1025 ints (co-ranks of the merge rounds; reduction scratch)
    double* red = reinterpret_cast<double*>(xch);
    unsigned short* tagl = reinterpret_cast<unsigned short*>(xch + 1026);  // n tags: training index of the sorted position
    const int nthr = blockDim.x;
    const int n = (int)T, k = pa.k;
    const int per = (n + nthr - 1) / nthr;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
#define SD_TID()                                        \
    int tid = (int)threadIdx.x;                         \
    asm volatile("" : "+v"(tid));                       \
    const int lane = tid & 63, wave = tid >> 6;         \
    (void)lane;                                         \
    (void)wave
#define SD_LW(i) ((int)(((i) & 1) ? (Lw2[(i) >> 1] >> 16) : (Lw2[(i) >> 1] & 0xffffu)))
    const double kk = uniform_f64((double)k), rk = uniform_f64(1.0 / (double)k);
    const int M = n - k > 0 ? n - k : 0;
    int nsteps = 0;
    while ((1 << nsteps) < (k + 1 < M + 1 ? k + 1 : M + 1)) ++nsteps;
    int64_t step, end;
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        global_f64* const orow = (global_f64*)(pa.out + c * 3 * pa.oc_Tq);  // (uniform) staging rows: predictions, probabilities, spreads
        global_f64* const prow = orow + pa.oc_Tq;
        global_f64* const erow = prow + pa.oc_Tq;
        if (fit_status[c] != 0) {
            // masked / non-finite training series: every query answers NaN (what the query phase below does for such a cell)
            for (int j = (int)threadIdx.x; j < (int)Tq; j += nthr) {
                orow[j] = nan;
                erow[j] = nan;
                if (!skip_prob || k == 1) prow[j] = nan;
            }
            continue;
        }
        // ---- the sorted runs of 64 * K tagged keys -> merge rounds 6 .. (analog_sort2_kernel<K, true>)
        __syncthreads();
        {
            SD_TID();
            const double* rc = runs + c * (int64_t)np;
            double kv[K + 1];
#pragma unroll
            for (int t2 = 0; t2 <= K; ++t2) {
                const int i = tid + t2 * nthr;
                kv[t2] = i < np ? rc[i] : inf;
            }
#pragma unroll
            for (int t2 = 0; t2 <= K; ++t2) {
                const int i = tid + t2 * nthr;
                if (i <= np) buf[i] = kv[t2];
            }
        }
        __syncthreads();
        {
            SD_TID();
            sdsort::block_merge_rounds<K>(buf, np, xch, tid, nthr, 6);
            bool odd = odd_flags[c] != 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * tid + i;
                if (j + 1 < n) odd |= ((__double_as_longlong(buf[j]) ^ __double_as_longlong(buf[j + 1])) >> 14) == 0;
            }
            if (__syncthreads_or(odd) != 0) {
                if (tid == 0) worklist[atomicAdd(work_count, 1)] = (int32_t)(cell0 + c);
                continue;
            }
            // tags -> LDS; x in training order -> buf; gathered through the tags -> registers -> buf = xs
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) tagl[pos] = (unsigned short)(__double_as_longlong(buf[pos]) & kTagMask);
            }
            __syncthreads();
            const double* x = Xc + c * T;
            double xv[K];
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                xv[s2] = pos < n ? x[pos] : 0.0;
            }
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = sd_finite(xv[s2]) ? xv[s2] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                xv[s2] = pos < n ? buf[tagl[pos]] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = xv[s2];
            }
            if (tid == 0) buf[n] = inf;
        }
        __syncthreads();
        // ---- generation 1 (analog_f1_mean3_kernel): sorted training values -> window start of every query
        SD_TID();
        const double* qrow = Xq + c * Tq;
        const int nq = (int)Tq;  // (one pass: the host sends Tq <= kPhQ * 1024 here)
        // (the queries of a thread are fetched two pairs ahead of their use instead of all at once: 8 registers instead of 32)
        double qn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * nthr;
            qn[i] = j < nq ? qrow[j] : 0.0;
        }
        unsigned Lw2[kPhQ / 2];
        unsigned okmask = 0u, nanmask = 0u, walkmask = 0u;
#pragma unroll
        for (int i0 = 0; i0 < kPhQ; i0 += 2) {
            double q[2];
            bool has[2], ok[2];
            int lo[2], hi[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                has[j] = tid + (i0 + j) * nthr < nq;
                q[j] = qn[j];
                qn[j] = qn[j + 2];
                const int jn = tid + (i0 + j + 4) * nthr;
                qn[j + 2] = (i0 + j + 4 < kPhQ && jn < nq) ? qrow[jn] : 0.0;
                ok[j] = has[j] && sd_finite(q[j]);
                if (has[j] && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                if (!ok[j]) q[j] = 0.0;
            }
            int pos[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) pos[j] = -1;
#pragma unroll 1
            for (int len = n; len > 1;) {
                int half = len >> 1;
                if ((half & 15) == 0) --half;
                len -= half;
#pragma unroll
                for (int j = 0; j < 2; ++j) pos[j] += buf[pos[j] + half] < q[j] ? half : 0;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = pos[j] + 1 + (buf[pos[j] + 1] < q[j] ? 1 : 0);
                lo[j] = p - k > 0 ? p - k : 0;
                hi[j] = p < M ? p : M;
            }
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int mid = (lo[j] + hi[j]) >> 1;
                    const bool act = lo[j] < hi[j];
                    const bool right = sq_dist(q[j], buf[mid]) > sq_dist(q[j], buf[mid + k]);
                    lo[j] = (act && right) ? mid + 1 : lo[j];
                    hi[j] = (act && !right) ? mid : hi[j];
                }
            }
            Lw2[i0 >> 1] = (unsigned)lo[0] | ((unsigned)lo[1] << 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = i0 + j;
                if (!has[j]) continue;
                if (!ok[j]) {
                    nanmask |= 1u << i;
                    continue;
                }
                const int L = lo[j];
                const double dL = sq_dist(q[j], buf[L]), dR = sq_dist(q[j], buf[L + k - 1]);
                const double worst = dL > dR ? dL : dR;
                const bool sep_l = L == 0 || sq_dist(q[j], buf[L - 1]) > worst;
                const bool sep_r = L + k >= n || sq_dist(q[j], buf[L + k]) > worst;
                if (sep_l && sep_r) okmask |= 1u << i;
                else walkmask |= 1u << i;
            }
        }
        if (__syncthreads_or(walkmask != 0u) != 0) {  // some window needs the exact walk: the whole cell goes to the split path
            if (tid == 0) worklist[atomicAdd(work_count, 1)] = (int32_t)(cell0 + c);
            continue;
        }
        // ---- y in training order -> buf (and its mean, summed as analog_sort2_kernel does); gathered -> buf = yx
        double ybar;
        {
            const double* yy = yc + c * T;
            double yv[K];
            double ysum = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                yv[s2] = pos < n ? yy[pos] : 0.0;
            }
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = yv[s2];
                ysum += yv[s2];
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ysum += __shfl_xor(ysum, o, 64);
            if (lane == 0) red[wave] = ysum;
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                yv[s2] = pos < n ? buf[tagl[pos]] : 0.0;
            }
            double tot = 0.0;
            for (int w = 0; w < 16; ++w) tot += red[w];
            ybar = uniform_f64(tot / (double)n);
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = yv[s2];
            }
        }
        __syncthreads();
        if (k == 1) {
            // a single analog (best_analog, or n_analogs = 1: gard.py:291-296)
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) {
                const int idx = tid + i * nthr;
                const bool okq = (okmask >> i) & 1u;
                if (okq || ((nanmask >> i) & 1u)) {
                    const double a1 = buf[okq ? SD_LW(i) : 0];
                    const bool exc = !pa.has_thresh || a1 > pa.thresh;  // gard.py:307
                    orow[idx] = !okq ? nan : (pa.kind == SD_ANALOG_BEST || exc) ? a1 : 0.0;
                    prow[idx] = !okq ? nan : pa.has_thresh ? (exc ? 1.0 : 0.0) : 1.0;
                    erow[idx] = !okq ? nan : exc ? 0.0 : nan;
                }
            }
            continue;
        }
        // ---- generation 2: exclusive prefix sums of d = yx - mean(y) -> window means
        const int beg = per * tid;
        double d[PER];
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = beg + i;
            d[i] = (i < per && j < n) ? buf[j] - ybar : 0.0;
            a += d[i];
            b += d[i] * d[i];
        }
        double ia = a, ib = b;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
            if (lane >= o) {
                ia += ta;
                ib += tb;
            }
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = ia;
        __syncthreads();
        double ra = ia - a;
        for (int w = 0; w < wave; ++w) ra += wsum[w];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = beg + i;
            if (i < per && j <= n) buf[j] = ra;
            ra += d[i];
        }
        if (tid == nthr - 1 && beg + per == n) buf[n] = ra;
        __syncthreads();
        double m1[kPhQ];
#pragma unroll
        for (int i = 0; i < kPhQ; ++i) m1[i] = (okmask >> i) & 1u ? div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) : 0.0;
        // ---- generation 3: exclusive prefix sums of d^2 -> spreads, outputs
        __syncthreads();
        if (lane == 63) wsum[wave] = ib;
        __syncthreads();
        double rb = ib - b;
        for (int w = 0; w < wave; ++w) rb += wsum[w];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = beg + i;
            if (i < per && j <= n) buf[j] = rb;
            rb += d[i] * d[i];
        }
        if (tid == nthr - 1 && beg + per == n) buf[n] = rb;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kPhQ; ++i) {
            const int idx = tid + i * nthr;
            double pred, err;
            if ((okmask >> i) & 1u) {
                const double var = div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) - m1[i] * m1[i];
                pred = ybar + m1[i];                 // gard.py:329-333
                err = sqrt(var > 0.0 ? var : 0.0);   // gard.py:345
            } else if ((nanmask >> i) & 1u) {
                pred = err = nan;
            } else {
                continue;
            }
            orow[idx] = pred;
            erow[idx] = err;
            if (!skip_prob) prow[idx] = pred != pred ? nan : 1.0;  // gard.py:346
        }
    }
#undef SD_LW
#undef SD_TID
}

// ------------------------------------------------------------------------------------------------
// general F predict: brute force, training rows staged through LDS, per-thread top-k in scratch
// ------------------------------------------------------------------------------------------------
constexpr int kBfThreads = 256;
constexpr int kBfChunk = 1024;

__global__ void __launch_bounds__(kBfThreads) analog_bf_predict_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                      int64_t Tq, int64_t T, int F, int64_t C,
                                                                      const double* __restrict__ Xc,
                                                                      const double* __restrict__ yc,
                                                                      const int32_t* __restrict__ fit_status,
                                                                      int32_t* status, double* scratch_d,
                                                                      int32_t* scratch_i, PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xt = reinterpret_cast<double*>(smem_raw);  // [F][kBfChunk]
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int k = pa.k;
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    int64_t step, end;
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const double* Xcell = Xc + c * F * T;
        for (int64_t tq0 = 0; tq0 < Tq; tq0 += nthr) {
            const int64_t tq = tq0 + tid;
            const bool has_q = tq < Tq;
            double q[kMaxF];
            bool ok = active && has_q;
            for (int f = 0; f < F; ++f) {
                q[f] = has_q ? Xq[(tq * F + f) * ld + c] : 0.0;
                if (active && has_q && !sd_finite(q[f])) {
                    atomicOr(&status[c], SDI_NONFINITE);
                    ok = false;
                }
            }
            // unsorted top-k with tracked worst element
            for (int i = 0; i < k; ++i) {
                sd[(int64_t)i * nthr + tid] = inf;
                si[(int64_t)i * nthr + tid] = 0x7fffffff;
            }
            double worst_d = inf;
            int worst_i = 0x7fffffff, worst_slot = 0;
            for (int64_t j0 = 0; j0 < T; j0 += kBfChunk) {
                const int nj = (int)((T - j0) < kBfChunk ? (T - j0) : kBfChunk);
                __syncthreads();
                if (active)
                    for (int i = tid; i < nj * F; i += nthr) {
                        const int f = i / nj, j = i - f * nj;
                        xt[f * kBfChunk + j] = Xcell[(int64_t)f * T + j0 + j];
                    }
                __syncthreads();
                if (!ok) continue;
                for (int j = 0; j < nj; ++j) {
                    double d = 0.0;
                    for (int f = 0; f < F; ++f) {
                        const double df = q[f] - xt[f * kBfChunk + j];
                        d += df * df;
                    }
                    // ascending j: an equal distance with a larger index never displaces
                    if (d < worst_d) {
                        sd[(int64_t)worst_slot * nthr + tid] = d;
                        si[(int64_t)worst_slot * nthr + tid] = (int32_t)(j0 + j);
                        worst_d = -1.0;
                        worst_i = -1;
                        for (int i = 0; i < k; ++i) {
                            const double di = sd[(int64_t)i * nthr + tid];
                            const int ii = si[(int64_t)i * nthr + tid];
                            if (di > worst_d || (di == worst_d && ii > worst_i)) {
                                worst_d = di;
                                worst_i = ii;
                                worst_slot = i;
                            }
                        }
                    }
                }
            }
            if (ok) {
                // selection sort into ascending (rdist, index)
                for (int i = 0; i < k - 1; ++i) {
                    int best = i;
                    double bd = sd[(int64_t)i * nthr + tid];
                    int bi = si[(int64_t)i * nthr + tid];
                    for (int j = i + 1; j < k; ++j) {
                        const double dj = sd[(int64_t)j * nthr + tid];
                        const int ij = si[(int64_t)j * nthr + tid];
                        if (dj < bd || (dj == bd && ij < bi)) {
                            best = j;
                            bd = dj;
                            bi = ij;
                        }
                    }
                    if (best != i) {
                        sd[(int64_t)best * nthr + tid] = sd[(int64_t)i * nthr + tid];
                        si[(int64_t)best * nthr + tid] = si[(int64_t)i * nthr + tid];
                        sd[(int64_t)i * nthr + tid] = bd;
                        si[(int64_t)i * nthr + tid] = bi;
                    }
                }
            }
            if (has_q) finish_query(mode, pa, F, T, c, tq, q, Xcell, yc + c * T, sd, si, nthr, ok);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// general F predict, second form: one wave per (cell, 64 queries).  Every lane owns one query and keeps its
// k best (rdist, index) pairs sorted in LDS ([k][64]: lane-contiguous, conflict-free).  The training set is
// scanned in chunks of 64 points whose coordinates are wave-uniform (scalar loads, no LDS staging, no
// barriers); a chunk first yields a 64-bit mask of points closer than the lane's current k-th distance,
// then only the flagged points are inserted.  After the first few chunks the mask is almost always empty,
// so the steady state is 3F+3 vector instructions per (query, training point).
// ------------------------------------------------------------------------------------------------
// (d, idx) pairs of one lane as a binary max-heap in LDS ([k][64]): the root is the worst of the k best.
template <typename IT>
__device__ __forceinline__ bool pair_gt(double da, IT ia, double db, IT ib) { return da > db || (da == db && ia > ib); }

// place (d, idx) at the root and sift it down within the first `n` entries
template <typename IT>
__device__ __forceinline__ void heap_replace_root(double* sd, IT* si, int lane, int n, double d, IT idx) {
    int pos = 0;
    for (;;) {
        const int l = 2 * pos + 1;
        if (l >= n) break;
        const int r = l + 1;
        double dc = sd[l * 64 + lane];
        IT ic = si[l * 64 + lane];
        int ch = l;
        if (r < n) {
            const double dr = sd[r * 64 + lane];
            const IT ir = si[r * 64 + lane];
            if (pair_gt<IT>(dr, ir, dc, ic)) {
                dc = dr;
                ic = ir;
                ch = r;
            }
        }
        if (!pair_gt<IT>(dc, ic, d, idx)) break;
        sd[pos * 64 + lane] = dc;
        si[pos * 64 + lane] = ic;
        pos = ch;
    }
    sd[pos * 64 + lane] = d;
    si[pos * 64 + lane] = idx;
}

// One 64-point chunk [j0, j0 + nj) of a cell's training set (P: [F][T]) against the 64 queries of the wave.
//  * the coordinates (and, for the sorted copy, the original indices PI) of the chunk are also fetched one point per
//    lane at the start and parked in LDS (stage / stage_i) once the mask is built: the insertion loop, where every
//    lane looks at a different point, then reads LDS instead of paying a global-memory round trip per candidate
//    (that latency, not the 3F+3 instructions per pair, used to dominate these kernels);
//  * full chunks build the mask from wave-uniform coordinates (scalar loads) in groups of G points, the next group
//    requested before the current one is used; the last request fetches the first group of the chunk the caller will
//    scan next (next_j0), so that `cur` is ready on entry.
// SORTED: points arrive in feature-0 order, equal distances are decided by the index comparison with the heap root;
// otherwise they arrive in index order and an equal distance never displaces.
// points per group of wave-uniform coordinates: two groups (current + requested) of F x G doubles must fit the ~100
// scalar registers, or they spill into vector-register lanes inside the scan loop
template <int F>
constexpr int kScanG = F <= 2 ? 8 : 4;
template <int F, typename IT, bool SORTED>
__device__ __forceinline__ void scan_chunk(const double* __restrict__ P, int64_t T, const int32_t* __restrict__ PI, int j0, int nj,
                                           int next_j0, double (&cur)[F][kScanG<F>], const double (&q)[F], double& tau, double* sd,
                                           IT* si, int k, int lane, double* stage /* [F][64] */, int32_t* stage_i /* [64] */,
                                           int ablate = 0, unsigned long long* dbg = nullptr) {
    constexpr int G = kScanG<F>;
    double mine[F];
    int32_t mine_i = 0;
#pragma unroll
    for (int f = 0; f < F; ++f) mine[f] = lane < nj ? P[(int64_t)f * T + j0 + lane] : 0.0;
    if (SORTED) mine_i = lane < nj ? PI[j0 + lane] : 0;
    unsigned long long mask = 0ull;
    if (nj == 64) {
        double nxt[F][G];
#pragma unroll
        for (int jg = 0; jg < 64; jg += G) {
            const int jn = jg + G < 64 ? j0 + jg + G : next_j0;
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int g = 0; g < G; ++g) nxt[f][g] = P[(int64_t)f * T + jn + g];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                double d = 0.0;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const double df = q[f] - cur[f][g];
                    d += df * df;
                }
                const bool hit = SORTED ? d <= tau : d < tau;
                mask |= hit ? (1ull << (jg + g)) : 0ull;
            }
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int g = 0; g < G; ++g) cur[f][g] = nxt[f][g];
        }
    } else {
        for (int j = 0; j < nj; ++j) {
            double d = 0.0;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const double df = q[f] - P[(int64_t)f * T + j0 + j];
                d += df * df;
            }
            const bool hit = SORTED ? d <= tau : d < tau;
            mask |= hit ? (1ull << j) : 0ull;
        }
    }
    if (ablate & 1) mask = 0ull;  // (timing experiments: no insertions)
    if (__builtin_amdgcn_ballot_w64(mask != 0ull) == 0ull) return;  // (the usual case far from the queries)
    __syncthreads();  // one wave per workgroup: orders the LDS traffic of the previous chunk's insertions
#pragma unroll
    for (int f = 0; f < F; ++f) stage[f * 64 + lane] = mine[f];
    if (SORTED) stage_i[lane] = mine_i;
    __syncthreads();
    if (dbg) {  // (experiments: insertion rounds = largest number of flagged points of a lane)
        int pc = __builtin_popcountll(mask);
        for (int o = 32; o >= 1; o >>= 1) pc = max(pc, __shfl_xor(pc, o, 64));
        if (lane == 0) atomicAdd(&dbg[1], (unsigned long long)pc);
    }
    while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        double d = 0.0;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const double df = q[f] - stage[f * 64 + j];
            d += df * df;
        }
        if (SORTED) {
            if (d <= tau) {  // tau may have tightened since the mask was built
                const IT idx = (IT)stage_i[j];
                if (d < tau || idx < si[lane]) {  // equal distance: the smaller training index is the better pair
                    heap_replace_root<IT>(sd, si, lane, k, d, idx);
                    tau = sd[lane];
                }
            }
        } else if (d < tau) {  // ascending index: an equal distance with a larger index never displaces
            heap_replace_root<IT>(sd, si, lane, k, d, (IT)(j0 + j));
            tau = sd[lane];
        }
    }
}

template <int F, typename IT>
__global__ void __launch_bounds__(64) analog_bf2_predict_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                int64_t Tq, int64_t T, int64_t C, int nbatch,
                                                                const double* __restrict__ Xc, const double* __restrict__ yc,
                                                                const int32_t* __restrict__ fit_status, int32_t* status,
                                                                PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int k = pa.k, lane = threadIdx.x;
    double* sd = reinterpret_cast<double*>(smem_raw);  // [k][64]
    IT* si = reinterpret_cast<IT*>(sd + (size_t)k * 64);  // [k][64]; 16-bit indices when T <= 65535 (more waves per CU)
    const int64_t c = blockIdx.x / nbatch;
    const int64_t tq = (int64_t)(blockIdx.x % nbatch) * 64 + lane;
    const bool active = fit_status[c] == 0, has_q = tq < Tq;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double q[F];
    bool ok = active && has_q;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        q[f] = has_q ? Xq[(tq * F + f) * ld + c] : 0.0;
        if (active && has_q && !sd_finite(q[f])) {
            atomicOr(&status[c], SDI_NONFINITE);
            ok = false;
        }
    }
    for (int i = 0; i < k; ++i) {
        sd[i * 64 + lane] = inf;
        si[i * 64 + lane] = (IT)~(IT)0 >> 1;  // larger than any training index
    }
    const double* __restrict__ Xcell = Xc + c * F * T;  // [F][T]
    double tau = ok ? inf : -1.0;  // k-th best distance so far; a lane without a query never flags a point
    double* stage = reinterpret_cast<double*>(si + (size_t)k * 64);  // [F][64] chunk coordinates for the insertion loop
    double cur[F][kScanG<F>];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int g = 0; g < kScanG<F>; ++g) cur[f][g] = T >= 64 ? Xcell[(int64_t)f * T + g] : 0.0;
    for (int64_t j0 = 0; j0 < T; j0 += 64) {
        const int nj = (int)(T - j0 < 64 ? T - j0 : 64);
        const int next_j0 = j0 + 128 <= T ? (int)j0 + 64 : 0;  // one uniform form: the last full chunk re-reads the first group
        scan_chunk<F, IT, false>(Xcell, T, nullptr, (int)j0, nj, next_j0, cur, q, tau, sd, si, k, lane, stage, nullptr);
    }
    // heap -> ascending (rdist, index): move the root behind the shrinking heap, k - 1 times
    for (int n = k - 1; n > 0; --n) {
        const double dl = sd[n * 64 + lane];
        const IT il = si[n * 64 + lane];
        sd[n * 64 + lane] = sd[lane];
        si[n * 64 + lane] = si[lane];
        heap_replace_root<IT>(sd, si, lane, n, dl, il);
    }
    if (has_q) finish_query(mode, pa, F, T, c, tq, q, Xcell, yc + c * T, sd, si, 64, ok);
}

// ------------------------------------------------------------------------------------------------
// general F predict, third form: the bf2 scan restricted to a slab of feature 0.
// fit keeps a copy of the training points sorted by feature 0 (ps, original indices in pi); predict sorts the
// queries of a cell by feature 0 as well, so the 64 queries of a wave are neighbours on that axis.  The wave scans
// the sorted training set outwards from its queries in 64-point chunks, alternating right and left, and a side is
// finished when its next point is farther from every query of the wave *along feature 0 alone* than the largest
// current k-th distance:  rdist >= fl((x0 - q0)^2) >= fl((x0 - qmax0)^2)  (floating-point subtraction, squaring and
// the accumulation of non-negative terms are monotone), so a skipped point is strictly worse than every kept one and
// the (rdist, index) selection stays exact.  Points now arrive out of index order: equal distances are decided by
// the explicit index comparison against the heap root.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}

// ps[c][f][j] = X[c][f][pi[c][j]]: the training points of a cell in feature-0 order
__global__ void __launch_bounds__(256) analog_gather_sorted_kernel(const double* __restrict__ Xc, const int32_t* __restrict__ pi,
                                                                   int64_t T, int F, int64_t C, double* __restrict__ ps) {
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x)
        for (int64_t j = threadIdx.x; j < T; j += blockDim.x) {
            const int idx = pi[c * T + j];
            for (int f = 0; f < F; ++f) ps[(c * F + f) * T + j] = Xc[(c * F + f) * T + idx];
        }
}

// scan state of a wave: [R, T) and [0, L) are still to be scanned, sides alternate
struct SlabCursor {
    int R, L, side;
    bool rdone, ldone;
};
struct SlabChunk {
    int j0, nj;  // nj = 0: both sides are finished
};
// next chunk (all values wave-uniform).  A side is finished when its next point is farther along feature 0 from every
// query of the wave (qlo..qhi) than the largest k-th distance taumax.
__device__ __forceinline__ SlabChunk slab_pick(const double* __restrict__ P0, int T, double qlo, double qhi, double taumax,
                                               SlabCursor& cs) {
    if (!cs.rdone) {
        const double g = P0[cs.R] - qhi;
        cs.rdone = g > 0.0 && g * g > taumax;
    }
    if (!cs.ldone) {
        const double g = qlo - P0[cs.L - 1];
        cs.ldone = g > 0.0 && g * g > taumax;
    }
    SlabChunk ch{0, 0};
    if (cs.rdone && cs.ldone) return ch;
    const bool right = cs.rdone ? false : (cs.ldone ? true : cs.side == 0);
    cs.side ^= 1;
    if (right) {
        ch.j0 = cs.R;
        ch.nj = T - cs.R < 64 ? T - cs.R : 64;
        cs.R += ch.nj;
        cs.rdone = cs.R >= T;
    } else {
        ch.nj = cs.L < 64 ? cs.L : 64;
        ch.j0 = cs.L - ch.nj;
        cs.L = ch.j0;
        cs.ldone = cs.L <= 0;
    }
    ch.j0 = __builtin_amdgcn_readfirstlane(ch.j0);
    ch.nj = __builtin_amdgcn_readfirstlane(ch.nj);
    return ch;
}

template <int F>
__global__ void __launch_bounds__(64) analog_slab_predict_kernel(int mode, const double* __restrict__ qc /* [cc][F][Tq] */,
                                                                 const int32_t* __restrict__ qi /* [cc][Tq] */, int64_t c_base,
                                                                 int64_t Tq, int64_t T, int nbatch,
                                                                 const double* __restrict__ Xc, const double* __restrict__ yc,
                                                                 const double* __restrict__ ps, const int32_t* __restrict__ pi,
                                                                 const int32_t* __restrict__ fit_status, int32_t* status,
                                                                 PredictArgs pa, int ablate, unsigned long long* dbg) {
    typedef uint16_t IT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int k = pa.k, lane = threadIdx.x;
    double* sd = reinterpret_cast<double*>(smem_raw);   // [k][64]
    IT* si = reinterpret_cast<IT*>(sd + (size_t)k * 64);  // [k][64]
    const int64_t cl = blockIdx.x / nbatch, c = c_base + cl;
    const int64_t slot = (int64_t)(blockIdx.x % nbatch) * 64 + lane;
    const bool active = fit_status[c] == 0, has_q = slot < Tq;
    const int64_t tq = has_q ? qi[cl * Tq + slot] : 0;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double q[F];
    bool ok = active && has_q;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        q[f] = has_q ? qc[(cl * F + f) * Tq + tq] : 0.0;
        if (active && has_q && !sd_finite(q[f])) {
            atomicOr(&status[c], SDI_NONFINITE);
            ok = false;
        }
    }
    for (int i = 0; i < k; ++i) {
        sd[i * 64 + lane] = inf;
        si[i * 64 + lane] = (IT)0xffffu;  // (never compared: a real distance is finite)
    }
    const double* __restrict__ P = ps + c * F * T;  // [F][T], ascending in feature 0
    const int32_t* __restrict__ PI = pi + c * T;
    double tau = ok ? inf : -1.0;  // k-th best distance so far; a lane without a query never flags a point
    if (__any(ok)) {
        const double qlo = uniform_f64(wave_min_f64(ok ? q[0] : inf));
        const double qhi = uniform_f64(wave_max_f64(ok ? q[0] : -inf));
        // start between the wave's queries: first sorted point >= the middle of their range, rounded down to 8 points
        const double qmid = qlo + (qhi - qlo) * 0.5;
        int lo = 0, hi = (int)T;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (P[mid] < qmid) lo = mid + 1; else hi = mid;
        }
        const int p8 = __builtin_amdgcn_readfirstlane(lo) & ~7;
        const int R = p8, L = p8;  // [R, T) and [0, L) are still to be scanned
        const bool rdone = R >= (int)T, ldone = L <= 0;
        double* stage = reinterpret_cast<double*>(si + (size_t)k * 64);   // [F][64]
        int32_t* stage_i = reinterpret_cast<int32_t*>(stage + F * 64);     // [64]
        double cur[F][kScanG<F>];
        SlabCursor cs{R, L, 0, rdone, ldone};
        int cur_j = -1;
        SlabChunk ch = slab_pick(P, (int)T, qlo, qhi, uniform_f64(wave_max_f64(tau)), cs);
        while (ch.nj > 0) {
            // the chunk after this one is chosen with the k-th distances known now (one chunk stale: it can only scan
            // more than necessary), so that its first group is requested while this chunk is still being scanned
            const SlabChunk nx = slab_pick(P, (int)T, qlo, qhi, uniform_f64(wave_max_f64(tau)), cs);
            const int j0 = ch.j0, nj = ch.nj, j0n = nx.j0, njn = nx.nj;
            if (nj == 64 && cur_j != j0) {
#pragma unroll
                for (int f = 0; f < F; ++f)
#pragma unroll
                    for (int g = 0; g < kScanG<F>; ++g) cur[f][g] = P[(int64_t)f * T + j0 + g];
            }
            const int next_j0 = njn == 64 ? j0n : j0;
            scan_chunk<F, IT, true>(P, T, PI, j0, nj, next_j0, cur, q, tau, sd, si, k, lane, stage, stage_i, ablate, dbg);
            if (dbg && lane == 0) atomicAdd(&dbg[0], 1ull);
            if (nj == 64) cur_j = next_j0;
            ch = nx;
        }
    }
    if (ablate & 2) return;  // (timing experiments: no epilogue)
    // heap -> ascending (rdist, index): move the root behind the shrinking heap, k - 1 times
    for (int n = k - 1; n > 0; --n) {
        const double dl = sd[n * 64 + lane];
        const IT il = si[n * 64 + lane];
        sd[n * 64 + lane] = sd[lane];
        si[n * 64 + lane] = si[lane];
        heap_replace_root<IT>(sd, si, lane, n, dl, il);
    }
    if (has_q) finish_query(mode, pa, F, T, c, tq, q, Xc + c * F * T, yc + c * T, sd, si, 64, ok);
}

// heap [k][64] of (double, IT) + the chunk staging area [F][64] doubles + [64] indices (16-byte aligned pieces)
size_t bf2_lds_bytes(int k, int F, size_t it_bytes) {
    const size_t heap = ((size_t)k * 64 * (sizeof(double) + it_bytes) + 15) / 16 * 16;
    return heap + (size_t)F * 64 * sizeof(double) + 64 * sizeof(int32_t);
}

template <int F, typename IT>
int launch_bf2i(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int32_t* status_p,
                const PredictArgs& pa) {
    const size_t lds = bf2_lds_bytes(pa.k, F, sizeof(IT));
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_bf2_predict_kernel<F, IT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nbatch = (Tq + 63) / 64, nblocks = st->C * nbatch;
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "sd_analog_predict: too many (cell, query batch) pairs for one launch");
    SD_LAUNCH(ctx, "analog_bf2_predict_kernel", (analog_bf2_predict_kernel<F, IT>), dim3((unsigned)nblocks), dim3(64), lds, mode,
              Xq, ld, Tq, st->T, st->C, (int)nbatch, (const double*)st->X, (const double*)st->y, (const int32_t*)st->status,
              status_p, pa);
    return SD_OK;
}

template <int F>
int launch_bf2(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int32_t* status_p,
               const PredictArgs& pa) {
    if (st->T <= 65535) return launch_bf2i<F, uint16_t>(ctx, mode, st, Xq, ld, Tq, status_p, pa);
    return launch_bf2i<F, int32_t>(ctx, mode, st, Xq, ld, Tq, status_p, pa);
}

// Query order for the slab search.  A wave stops scanning when the axis distance exceeds the *largest* k-th distance of
// its 64 queries, so one query in a sparse region of the other features widens the slab for all of them.  The queries of
// a cell are therefore first classed by s2 = sum_{f>=1} ((q_f - mean_f) / std_f)^2 (moments of the query series itself)
// into nb classes holding 1/2, 1/4, 1/8, ... of the queries (thresholds = order statistics of s2, from a sort), then
// sorted by feature 0 inside a class: key = 4 * class + q0 / (1 + |q0|).  The order only groups the work; any order is
// exact.  (F=3, T=Tq=14 600, k=30, normal data: 64 % of the training points scanned per wave without classes, 29 % with 8.)
__global__ void __launch_bounds__(256) analog_slab_s2_kernel(const double* __restrict__ qc /* [C][F][Tq] */, int64_t Tq, int F,
                                                             int64_t C, double* __restrict__ s2 /* [C][Tq] */) {
    __shared__ double red[2][4];
    __shared__ double mom[2][kMaxF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* q = qc + c * F * Tq;
        for (int f = 1; f < F; ++f) {
            double m = 0.0;
            for (int pass = 0; pass < 2; ++pass) {  // mean, then the sum of squared deviations, over the finite entries
                double a = 0.0, cnt = 0.0;
                for (int64_t j = tid; j < Tq; j += blockDim.x) {
                    const double v = q[f * Tq + j];
                    if (sd_finite(v)) {
                        a += pass == 0 ? v : (v - m) * (v - m);
                        cnt += 1.0;
                    }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    a += __shfl_xor(a, o, 64);
                    cnt += __shfl_xor(cnt, o, 64);
                }
                __syncthreads();
                if (lane == 0) {
                    red[0][wave] = a;
                    red[1][wave] = cnt;
                }
                __syncthreads();
                const double ta = red[0][0] + red[0][1] + red[0][2] + red[0][3];
                const double tc = red[1][0] + red[1][1] + red[1][2] + red[1][3];
                if (pass == 0) {
                    m = tc > 0.0 ? ta / tc : 0.0;
                } else if (tid == 0) {
                    mom[0][f] = m;
                    mom[1][f] = ta > 0.0 ? tc / ta : 0.0;  // 1 / variance
                }
            }
        }
        __syncthreads();
        for (int64_t j = tid; j < Tq; j += blockDim.x) {
            double s = 0.0;
            bool fin = sd_finite(q[j]);
            for (int f = 1; f < F; ++f) {
                const double v = q[f * Tq + j];
                fin = fin && sd_finite(v);
                s += (v - mom[0][f]) * (v - mom[0][f]) * mom[1][f];
            }
            s2[c * Tq + j] = fin && s < 1e300 ? s : 1e300;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) analog_slab_key_kernel(const double* __restrict__ qc, const double* __restrict__ s2_sorted,
                                                              int64_t Tq, int F, int64_t C, int nb,
                                                              double* __restrict__ key /* in: s2, out: sort key */) {
    __shared__ double th[8];
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < nb - 1) {
            const int64_t pos = Tq - (Tq >> (threadIdx.x + 1));  // 1/2, 3/4, 7/8, ... of the queries lie below
            th[threadIdx.x] = s2_sorted[c * Tq + (pos < Tq ? pos : Tq - 1)];
        }
        __syncthreads();
        for (int64_t j = threadIdx.x; j < Tq; j += blockDim.x) {
            const double s = key[c * Tq + j];
            int b = 0;
            for (int i = 0; i < nb - 1; ++i) b += s >= th[i] ? 1 : 0;
            const double q0 = qc[c * F * Tq + j];
            const double t = sd_finite(q0) ? q0 / (1.0 + __builtin_fabs(q0)) : 0.0;  // monotone map into (-1, 1)
            key[c * Tq + j] = 4.0 * (double)b + t;
        }
    }
}

template <int F>
int launch_slab(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* qc, const int32_t* qi, int64_t cb, int64_t cc,
                int64_t Tq, int32_t* status_p, const PredictArgs& pa) {
    const size_t lds = bf2_lds_bytes(pa.k, F, sizeof(uint16_t));
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_slab_predict_kernel<F>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nbatch = (Tq + 63) / 64, nblocks = cc * nbatch;
    const char* eab = sd_dev_env("SD_ANALOG_ABLATE");  // timing experiments only (results are wrong): 1 no insertions, 2 no epilogue
    const int ablate = eab ? atoi(eab) : 0;
    sd_scratch dbg;  // 4: count scanned chunks and insertion rounds, printed per launch
    if (ablate & 4) {
        SD_HIP(dbg.alloc(ctx, 16));
        SD_HIP(hipMemsetAsync(dbg.p, 0, 16, ctx->stream));
    }
    SD_LAUNCH(ctx, "analog_slab_predict_kernel", (analog_slab_predict_kernel<F>), dim3((unsigned)nblocks), dim3(64), lds, mode, qc,
              qi, cb, Tq, st->T, (int)nbatch, (const double*)st->X, (const double*)st->y, (const double*)st->ps,
              (const int32_t*)st->xi, (const int32_t*)st->status, status_p, pa, ablate, dbg.as<unsigned long long>());
    if (ablate & 4) {
        unsigned long long h[2];
        SD_HIP(hipMemcpyAsync(h, dbg.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[slab] waves %lld: chunks/wave %.1f insertion rounds/wave %.1f\n", (long long)nblocks,
                (double)h[0] / (double)nblocks, (double)h[1] / (double)nblocks);
    }
    return SD_OK;
}

// F > 1 with the feature-0 sorted copy: queries go cell-major, are sorted by feature 0 per cell, and every wave scans
// only the slab of training points its 64 neighbouring queries can reach (analog_slab_predict_kernel)
int predict_slab(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int32_t* status_p,
                 const PredictArgs& pa) {
    const int F = st->F;
    const int64_t C = st->C, nbatch = (Tq + 63) / 64;
    int64_t chunk = 4096;
    while (chunk > 1 && chunk * nbatch >= ((int64_t)1 << 31)) chunk >>= 1;
    const int64_t cc_max = C < chunk ? C : chunk;
    const int Kq = sort2_width(Tq, ctx->lds_max);
    // classes of the query order (analog_slab_s2_kernel); a short series would only get waves that straddle classes
    const char* ecl = sd_dev_env("SD_ANALOG_SLAB_CLASSES");
    int nclass = ecl ? atoi(ecl) : (int)std::min<int64_t>(8, Tq / 512);
    nclass = nclass < 1 ? 1 : (nclass > 8 ? 8 : nclass);
    sd_scratch qc, qs, qi, key;
    if (nclass > 1) SD_HIP(key.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
    SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tq * F * cc_max));
    SD_HIP(qs.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
    SD_HIP(qi.alloc(ctx, sizeof(int32_t) * (size_t)Tq * cc_max));
    for (int64_t cb = 0; cb < C; cb += chunk) {
        const int64_t cc = C - cb < chunk ? C - cb : chunk;
        dim3 tgrid((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32));
        for (int f = 0; f < F; ++f)
            SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, tgrid, dim3(256), 0, Xq + cb, ld, Tq, F, f, cc,
                      qc.as<double>(), status_p + cb, 0);
        const int nbk = (int)std::min<int64_t>(cc, (int64_t)ctx->cu_count * 8);
        Sort2Args a{qc.as<double>(), (int64_t)F * Tq, 1, nullptr, Tq, cc, qs.as<double>(), qi.as<int32_t>(),
                    nullptr, nullptr, nullptr};
        if (nclass > 1) {
            SD_LAUNCH(ctx, "analog_slab_s2_kernel", analog_slab_s2_kernel, dim3(nbk), dim3(256), 0, (const double*)qc.p, Tq, F, cc,
                      key.as<double>());
            a.X = key.as<double>();
            a.x_stride = Tq;
            SD_TRY(launch_sort2_width(ctx, Kq, a));  // qs = sorted s2 (class thresholds)
            SD_LAUNCH(ctx, "analog_slab_key_kernel", analog_slab_key_kernel, dim3(nbk), dim3(256), 0, (const double*)qc.p,
                      (const double*)qs.p, Tq, F, cc, nclass, key.as<double>());
        }
        SD_TRY(launch_sort2_width(ctx, Kq, a));  // qi = query order
        const double* q = qc.as<double>();
        const int32_t* qix = qi.as<int32_t>();
        switch (F) {
            case 2: SD_TRY(launch_slab<2>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 3: SD_TRY(launch_slab<3>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 4: SD_TRY(launch_slab<4>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 5: SD_TRY(launch_slab<5>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 6: SD_TRY(launch_slab<6>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 7: SD_TRY(launch_slab<7>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            default: SD_TRY(launch_slab<8>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
        }
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));  // the staging buffers go back to the block cache at scope exit
    return SD_OK;
}

__global__ void __launch_bounds__(256) analog_status_public_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                                   int64_t C, int32_t* __restrict__ outp) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const int32_t bits = a[c] | (b ? b[c] : 0);
        outp[c] = (bits & SDI_MASKED) ? SD_CELL_MASKED : (bits & SDI_NONFINITE) ? SD_CELL_NONFINITE : (bits & SDI_ONE_CLASS) ? SD_CELL_ONE_CLASS : SD_CELL_OK;
    }
}

// exclusive prefix sums of the centred analog values (analog_prefix_kernel), built when a kernel that reads them from
// memory first runs on a state (calls on a context are serialised)
int ensure_prefix_sums(sd_ctx* ctx, const sd_analog_state* st) {
    if (st->pq != nullptr) return SD_OK;
    sd_analog_state* ms = const_cast<sd_analog_state*>(st);
    SD_HIP(sd_pool_malloc(ctx, (void**)&ms->pq, sizeof(double) * 2 * (size_t)(st->T + 1) * st->C));
    const int nbp = (int)std::min<int64_t>(st->C, (int64_t)ctx->cu_count * 2);
    SD_LAUNCH(ctx, "analog_prefix_kernel", analog_prefix_kernel, dim3(nbp), dim3(1024), 0, (const double*)st->yx, st->T, st->C, ms->pq,
              ms->ybar, 1);
    return SD_OK;
}

int predict_common(int mode, sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int k,
                   int kind, int has_thresh, double thresh, const int32_t* sample_dev, int64_t ld_s, double* out,
                   int64_t ld_out, int64_t* inds, double* dist, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xq && out, "sd_analog_predict: NULL argument");
    SD_CHECK_ARG(Tq > 0 && ld >= st->C && ld_out >= st->C, "sd_analog_predict: bad sizes");
    SD_CHECK_ARG(k >= 1 && k <= st->T, "sd_analog_predict: k=%d must be in [1, T=%lld]", k, (long long)st->T);
    SD_CHECK_ARG(mode == 1 || (kind >= SD_ANALOG_BEST && kind <= SD_ANALOG_MEAN), "sd_analog_predict: unknown kind %d", kind);
    SD_CHECK_ARG(!(mode == 0 && kind == SD_ANALOG_SAMPLE) || sample_dev, "sd_analog_predict: sample_analogs needs sample_inds");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C, T = st->T;
    const int F = st->F;
    // PureAnalog.predict with a single analog is 'best_analog' whatever the configured kind (gard.py:291-296: n_analogs == 1);
    // the entry point sees k only, so a one-sample training set (k_ = 1 with n_analogs > 1) is treated the same way
    if (mode == 0 && k == 1) kind = SD_ANALOG_BEST;
    PredictArgs pa;
    pa.k = k;
    pa.kind = kind;
    pa.has_thresh = has_thresh;
    pa.thresh = thresh;
    pa.sample = sample_dev;
    pa.ld_s = ld_s;
    pa.out = out;
    pa.ld_out = ld_out;
    pa.inds = inds;
    pa.dist = dist;
    pa.oc_Tq = 0;
    sd_scratch status_p, sc_d, sc_i, status_pub;
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    pa.one_class = status_p.as<int32_t>();
    const bool f1 = st->xs != nullptr;
    const int nthr = f1 ? 1024 : kBfThreads;
    int nb = ctx->cu_count * (f1 ? 1 : 4);
    nb = (nb / 8) * 8;
    if (nb < 8) nb = 8;
    if ((int64_t)nb > ((C + 7) / 8) * 8) nb = (int)(((C + 7) / 8) * 8);
    SD_HIP(sc_d.alloc(ctx, sizeof(double) * (size_t)nb * k * nthr));
    SD_HIP(sc_i.alloc(ctx, sizeof(int32_t) * (size_t)nb * k * nthr));
    // (a thresholded regression needs the analogs themselves: logistic fit and subset OLS, gard.py:201-219)
    const bool window = f1 && (mode == 1 || kind != SD_ANALOG_SAMPLE) && !inds && !dist && st->yx != nullptr &&
                        !(mode == 1 && has_thresh) && sd_dev_env("SD_ANALOG_WALK") == nullptr;
    if (window) {
        // fewest value ranges such that xs and yx of a range (+ k entries of margin each side) fit the LDS
        int npass = 1;
        size_t lds = 0;
        for (;; ++npass) {
            const size_t cap = (size_t)((T + npass - 1) / npass) + 2 * (size_t)k + 1;
            lds = sizeof(double) * (2 * cap + 1);
            if (lds <= ctx->lds_max || npass >= 64) break;
        }
        SD_CHECK_ARG(lds <= ctx->lds_max, "sd_analog_predict: k=%d too large for the windowed path", k);
        // queries and outputs go through cell-major copies: the column accesses of a cell would be 8-byte
        // requests 8*ld bytes apart (one 64-byte sector each); the tiled transposes stream at HBM speed.
        // Cells are processed in chunks so that the staging buffers stay small (and cache-resident).
        const int64_t chunk = 16384;
        const int64_t cc_max = C < chunk ? C : chunk;
        sd_scratch qc, oc;
        SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
        SD_HIP(oc.alloc(ctx, sizeof(double) * (size_t)Tq * 3 * cc_max));
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_window_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // single pass with only xs in LDS (statistics from the prefix sums, or the window of yx read from memory)
        const size_t lds_mean = sizeof(double) * (size_t)(T + 1);
        const bool mean_only = (mode == 1 ? k >= 3 : (kind == SD_ANALOG_MEAN || kind == SD_ANALOG_WEIGHT || k == 1)) && st->ybar != nullptr &&
                               lds_mean <= ctx->lds_max && sd_dev_env("SD_ANALOG_NOPREFIX") == nullptr;
        const bool phases = mean_only && mode == 0 && ((kind == SD_ANALOG_MEAN && !has_thresh) || k == 1) && T <= 1024 * 20 &&
                            sd_dev_env("SD_ANALOG_NOPHASES") == nullptr;
        const int skip_prob = phases && !has_thresh ? 1 : 0;  // the probability column is 1 wherever the prediction is not NaN (gard.py:346)
        const size_t lds_mean3 = lds_mean;
        if (mean_only) {
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean3_kernel<8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean3));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean3_kernel<16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean3));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean3_kernel<20>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean3));
        }
        if (mean_only && !phases) SD_TRY(ensure_prefix_sums(ctx, st));
        if (mean_only && mode == 1 && st->rx == nullptr) {
            // first regression on this state: the cross-term prefix sums (calls on a context are serialised)
            sd_analog_state* ms = const_cast<sd_analog_state*>(st);
            SD_HIP(sd_pool_malloc(ctx, (void**)&ms->rx, sizeof(double) * (size_t)(T + 1) * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&ms->xbar, sizeof(double) * C));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_rx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_mean));
            SD_LAUNCH(ctx, "analog_rx_kernel", analog_rx_kernel, dim3((unsigned)std::min<int64_t>(C, (int64_t)ctx->cu_count * 2)),
                      dim3(1024), lds_mean, (const double*)st->xs, (const double*)st->yx, (const double*)st->ybar, T, C, ms->rx,
                      ms->xbar);
        }
        for (int64_t cb = 0; cb < C; cb += chunk) {
            const int64_t cc = C - cb < chunk ? C - cb : chunk;
            dim3 tgrid((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32));
            SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, tgrid, dim3(256), 0, Xq + cb, ld, Tq, 1, 0, cc,
                      qc.as<double>(), status_p.as<int32_t>() + cb, 0);
            PredictArgs pw = pa;
            pw.out = oc.as<double>();
            pw.oc_Tq = Tq;
            int nbc = nb;
            if ((int64_t)nbc > ((cc + 7) / 8) * 8) nbc = (int)(((cc + 7) / 8) * 8);
            // workgroups per cell in the single-pass kernel (see its qsplit): only when every XCD still gets whole groups
            const char* eqs = sd_dev_env("SD_ANALOG_QSPLIT");
            int qs = eqs ? atoi(eqs) : (mode == 1 ? 2 : 1);  // measured (ms per 16 384 cells), 1/2/4/8: regression 10.8/8.5/8.7/11.0, mean 5.5/5.7/6.4/8.3
            if (qs < 1 || nbc % (8 * qs) != 0 || cc < (int64_t)nbc || Tq < 4096) qs = 1;
            if (phases) {
                const int per = (int)((T + nthr - 1) / nthr);
                long long* trace_dev = nullptr;
                sd_scratch trace_buf;
                if (sd_dev_env("SD_M3_TRACE") != nullptr) {  // development library: phase clocks of the first cells of block 0
                    SD_HIP(trace_buf.alloc(ctx, sizeof(long long) * 128));
                    SD_HIP(hipMemsetAsync(trace_buf.p, 0, sizeof(long long) * 128, ctx->stream));
                    trace_dev = trace_buf.as<long long>();
                }
#define SD_MEAN3(PER)                                                                                                                  \
    SD_LAUNCH(ctx, "analog_f1_mean3_kernel", analog_f1_mean3_kernel<PER>, dim3(nbc), dim3(nthr), lds_mean3, (const double*)qc.p, Tq, T, \
              cc, (const double*)st->xs + cb * T, (const int32_t*)st->xi + cb * T, (const double*)st->ybar + cb,                      \
              (const double*)st->yx + cb * T, (const double*)st->X + cb * T, (const double*)st->y + cb * T,                           \
              (const int32_t*)st->status + cb, status_p.as<int32_t>() + cb, sc_d.as<double>(), sc_i.as<int32_t>(), pw, skip_prob, \
              trace_dev)
                if (per <= 8) SD_MEAN3(8);
                else if (per <= 16) SD_MEAN3(16);
                else SD_MEAN3(20);
#undef SD_MEAN3
                if (trace_dev != nullptr) {
                    long long h[128];
                    SD_HIP(hipMemcpyAsync(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
                    SD_HIP(hipStreamSynchronize(ctx->stream));
                    for (int r = 0; r < 8; ++r) {
                        fprintf(stderr, "mean3 trace cell %d:", r);
                        for (int j = 1; j <= 9; ++j) fprintf(stderr, " %lld", h[r * 16 + j] - h[r * 16 + j - 1]);
                        fprintf(stderr, "\n");
                    }
                }
            } else if (mean_only) {
                SD_LAUNCH(ctx, "analog_f1_mean_kernel", analog_f1_mean_kernel, dim3(nbc), dim3(nthr), lds_mean, mode,
                          (const double*)qc.p, Tq, T, cc, (const double*)st->xs + cb * T, (const int32_t*)st->xi + cb * T,
                          (const double*)st->pq + 2 * cb * (T + 1), (const double*)st->ybar + cb,
                          (const double*)st->rx + cb * (T + 1), (const double*)st->xbar + cb, (const double*)st->yx + cb * T,
                          (const double*)st->X + cb * T,
                          (const double*)st->y + cb * T, (const int32_t*)st->status + cb, status_p.as<int32_t>() + cb,
                          sc_d.as<double>(), sc_i.as<int32_t>(), pw, qs);
            } else {
                SD_LAUNCH(ctx, "analog_f1_window_kernel", analog_f1_window_kernel, dim3(nbc), dim3(nthr), lds, mode,
                          (const double*)qc.p, Tq, Tq, T, cc, npass, (const double*)st->xs + cb * T, (const int32_t*)st->xi + cb * T,
                          (const double*)st->yx + cb * T, (const double*)st->X + cb * T, (const double*)st->y + cb * T,
                          (const int32_t*)st->status + cb, status_p.as<int32_t>() + cb, sc_d.as<double>(), sc_i.as<int32_t>(), pw);
            }
            SD_LAUNCH(ctx, "analog_untranspose_kernel", analog_untranspose_kernel,
                      dim3((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32), skip_prob ? 2 : 3), dim3(256), 0,
                      (const double*)oc.p, Tq, cc, out + cb, ld_out, skip_prob);
        }
        SD_HIP(hipStreamSynchronize(ctx->stream));  // qc / oc go back to the block cache at scope exit
    } else if (f1) {
        const size_t lds = sizeof(double) * T;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_predict_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SD_LAUNCH(ctx, "analog_f1_predict_kernel", analog_f1_predict_kernel, dim3(nb), dim3(nthr), lds, mode, Xq, ld, Tq,
                  T, C, (const double*)st->xs, (const int32_t*)st->xi, (const double*)st->X, (const double*)st->y,
                  (const int32_t*)st->status, status_p.as<int32_t>(), sc_d.as<double>(), sc_i.as<int32_t>(), pa);
    } else if (F > 1 && st->ps != nullptr && bf2_lds_bytes(k, F, 2) <= ctx->lds_max && sort2_width(Tq, ctx->lds_max) != 0 &&
               sd_dev_env("SD_ANALOG_NOSLAB") == nullptr) {
        SD_TRY(predict_slab(ctx, mode, st, Xq, ld, Tq, status_p.as<int32_t>(), pa));
    } else if (bf2_lds_bytes(k, F, 4) <= ctx->lds_max && sd_dev_env("SD_ANALOG_BF1") == nullptr) {
        int32_t* sp = status_p.as<int32_t>();
        switch (F) {
            case 1: SD_TRY(launch_bf2<1>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 2: SD_TRY(launch_bf2<2>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 3: SD_TRY(launch_bf2<3>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 4: SD_TRY(launch_bf2<4>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 5: SD_TRY(launch_bf2<5>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 6: SD_TRY(launch_bf2<6>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 7: SD_TRY(launch_bf2<7>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            default: SD_TRY(launch_bf2<8>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
        }
    } else {
        const size_t lds = sizeof(double) * F * kBfChunk;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_bf_predict_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SD_LAUNCH(ctx, "analog_bf_predict_kernel", analog_bf_predict_kernel, dim3(nb), dim3(nthr), lds, mode, Xq, ld, Tq,
                  T, F, C, (const double*)st->X, (const double*)st->y, (const int32_t*)st->status,
                  status_p.as<int32_t>(), sc_d.as<double>(), sc_i.as<int32_t>(), pa);
    }
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "analog_status_public_kernel", analog_status_public_kernel, dim3((unsigned)((C + 255) / 256)),
                  dim3(256), 0, (const int32_t*)st->status, (const int32_t*)status_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int predict_host(int mode, sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int kind,
                 int has_thresh, double thresh, const int32_t* sample, double* out, int64_t* inds, double* dist,
                 int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xq && out, "sd_analog_predict: NULL argument");
    SD_CHECK_ARG(Tq > 0 && k >= 1, "sd_analog_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C;
    sd_scratch dq, dout, dinds, ddist, dsamp;
    const size_t qb = sizeof(double) * (size_t)Tq * st->F * C, ob = sizeof(double) * (size_t)Tq * 3 * C;
    SD_HIP(dq.alloc(ctx, qb));
    SD_HIP(dout.alloc(ctx, ob));
    SD_TRY(sd_copy_h2d(ctx, dq.p, Xq, qb));
    if (inds) SD_HIP(dinds.alloc(ctx, sizeof(int64_t) * (size_t)Tq * k * C));
    if (dist) SD_HIP(ddist.alloc(ctx, sizeof(double) * (size_t)Tq * k * C));
    if (sample) {
        SD_HIP(dsamp.alloc(ctx, sizeof(int32_t) * (size_t)Tq * C));
        SD_TRY(sd_copy_h2d(ctx, dsamp.p, sample, sizeof(int32_t) * (size_t)Tq * C));
    }
    SD_TRY(predict_common(mode, ctx, st, dq.as<double>(), C, Tq, k, kind, has_thresh, thresh, dsamp.as<int32_t>(), C,
                          dout.as<double>(), C, dinds.as<int64_t>(), ddist.as<double>(), cell_status));
    SD_TRY(sd_copy_d2h(ctx, out, dout.p, ob));
    if (inds) SD_HIP(hipMemcpyAsync(inds, dinds.p, sizeof(int64_t) * (size_t)Tq * k * C, hipMemcpyDeviceToHost, ctx->stream));
    if (dist) SD_HIP(hipMemcpyAsync(dist, ddist.p, sizeof(double) * (size_t)Tq * k * C, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

// columns `list[0 .. nw)` of a [R, ld] field <-> a packed [R, nw] field (cells the fused kernel handed back)
__global__ void __launch_bounds__(256) analog_gather_cells_kernel(const double* __restrict__ src, int64_t ld, int64_t R,
                                                                  const int32_t* __restrict__ list, int64_t nw,
                                                                  double* __restrict__ dst) {
    const int64_t total = R * nw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nw, j = i - r * nw;
        dst[i] = src[r * ld + list[j]];
    }
}
__global__ void __launch_bounds__(256) analog_scatter_cells_kernel(const double* __restrict__ src, int64_t R,
                                                                   const int32_t* __restrict__ list, int64_t nw,
                                                                   double* __restrict__ dst, int64_t ld) {
    const int64_t total = R * nw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nw, j = i - r * nw;
        dst[r * ld + list[j]] = src[i];
    }
}

template <int K>
int launch_fused_k(sd_ctx* ctx, int nbc, size_t lds, const double* runs, int np, const int32_t* odd, const double* Xc, const double* yc,
                   const double* qc, int64_t Tq, int64_t T, int64_t cc, const int32_t* st_fit, int32_t* st_p, int32_t* worklist,
                   int32_t* work_count, int64_t cell0, const PredictArgs& pw, int skip_prob) {
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_fused_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "analog_f1_fused_kernel", analog_f1_fused_kernel<K>, dim3(nbc), dim3(1024), lds, runs, np, odd, Xc, yc, qc, Tq, T, cc, st_fit,
              st_p, worklist, work_count, cell0, pw, skip_prob);
    return SD_OK;
}

// LDS of analog_f1_fused_kernel: keys (np + 1 doubles), co-ranks (1025 ints, padded), tags (T x 16 bit); + its static arrays
size_t fused_lds_bytes(int np, int64_t T) { return sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1026 + ((sizeof(uint16_t) * (size_t)T + 15) & ~(size_t)15); }

// the split path on device fields: fit -> predict -> drop the state
int fit_predict_split(sd_ctx* ctx, const double* X, const double* y, int64_t ld, int64_t T, int F, int64_t C, const double* Xq,
                      int64_t ld_q, int64_t Tq, int k, int kind, int has_thresh, double thresh, double* out, int64_t ld_out,
                      int32_t* cell_status) {
    SD_CHECK_ARG(ld_q == ld, "sd_analog_fit_predict: this configuration needs ld_q == ld");
    sd_analog_state* st = nullptr;
    SD_TRY(sd_analog_fit_dev(ctx, X, y, ld, T, F, C, &st));
    const int rc = predict_common(0, ctx, st, Xq, ld_q, Tq, k, kind, has_thresh, thresh, nullptr, 0, out, ld_out, nullptr, nullptr, cell_status);
    sd_analog_state_destroy(st);
    return rc;
}

int fit_predict_dev(sd_ctx* ctx, const double* X, const double* y, int64_t ld, int64_t T, int F, int64_t C, const double* Xq,
                    int64_t ld_q, int64_t Tq, int k, int kind, int has_thresh, double thresh, double* out, int64_t ld_out,
                    int32_t* cell_status) {
    SD_CHECK_ARG(ctx && X && y && Xq && out, "sd_analog_fit_predict: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && Tq > 0 && ld >= C && ld_q >= C && ld_out >= C, "sd_analog_fit_predict: bad sizes");
    SD_CHECK_ARG(F >= 1 && F <= kMaxF, "sd_analog_fit_predict: F=%d outside [1,%d]", F, kMaxF);
    SD_CHECK_ARG(k >= 1 && k <= T, "sd_analog_fit_predict: k=%d must be in [1, T=%lld]", k, (long long)T);
    SD_CHECK_ARG(kind >= SD_ANALOG_BEST && kind <= SD_ANALOG_MEAN && kind != SD_ANALOG_SAMPLE, "sd_analog_fit_predict: kind %d (sample_analogs needs the split calls)", kind);
    SD_HIP(hipSetDevice(ctx->device));
    if (k == 1) kind = SD_ANALOG_BEST;  // (as in predict_common: gard.py:291-296)
    const int K = F == 1 ? sort2_width(T, ctx->lds_max) : 0;
    const int64_t chunk_t = 64 * (int64_t)(K > 0 ? K : 1);
    const int np = (int)(((T + chunk_t - 1) / chunk_t) * chunk_t);
    const bool fused = K != 0 && tile_sort_applies(K, T, C, ctx->lds_max) && ((kind == SD_ANALOG_MEAN && !has_thresh) || k == 1) &&
                       Tq <= (int64_t)kPhQ * 1024 && fused_lds_bytes(np, T) + 512 <= ctx->lds_max && sd_dev_env("SD_ANALOG_NOFUSE") == nullptr;
    if (!fused) return fit_predict_split(ctx, X, y, ld, T, F, C, Xq, ld_q, Tq, k, kind, has_thresh, thresh, out, ld_out, cell_status);

    PredictArgs pa;
    pa.k = k; pa.kind = kind; pa.has_thresh = has_thresh; pa.thresh = thresh;
    pa.sample = nullptr; pa.ld_s = 0; pa.out = out; pa.ld_out = ld_out; pa.inds = nullptr; pa.dist = nullptr; pa.oc_Tq = 0;
    const int64_t chunk = 16384;
    const int64_t cc_max = C < chunk ? C : chunk;
    sd_scratch Xc, yc, runs, st_fit, st_p, odd, list, qc, oc, status_pub;
    SD_HIP(Xc.alloc(ctx, sizeof(double) * (size_t)T * C));
    SD_HIP(yc.alloc(ctx, sizeof(double) * (size_t)T * C));
    SD_HIP(runs.alloc(ctx, sizeof(double) * (size_t)np * C));
    SD_HIP(st_fit.alloc(ctx, sizeof(int32_t) * (size_t)C));
    SD_HIP(st_p.alloc(ctx, sizeof(int32_t) * (size_t)C));
    SD_HIP(odd.alloc(ctx, sizeof(int32_t) * (size_t)C));
    SD_HIP(list.alloc(ctx, sizeof(int32_t) * (size_t)(C + 1)));
    SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
    SD_HIP(oc.alloc(ctx, sizeof(double) * (size_t)Tq * 3 * cc_max));
    pa.one_class = st_p.as<int32_t>();
    int32_t* work_count = list.as<int32_t>();
    int32_t* worklist = work_count + 1;
    SD_HIP(hipMemsetAsync(st_fit.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
    SD_HIP(hipMemsetAsync(st_p.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
    SD_HIP(hipMemsetAsync(odd.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
    SD_HIP(hipMemsetAsync(work_count, 0, sizeof(int32_t), ctx->stream));
    SD_TRY(launch_tile_sort(ctx, K, X, y, ld, T, C, Xc.as<double>(), yc.as<double>(), runs.as<double>(), np, st_fit.as<int32_t>(), odd.as<int32_t>()));
    const int skip_prob = !has_thresh ? 1 : 0;
    const size_t lds = fused_lds_bytes(np, T);
    int nb = (ctx->cu_count / 8) * 8;
    if (nb < 8) nb = 8;
    if ((int64_t)nb > ((C + 7) / 8) * 8) nb = (int)(((C + 7) / 8) * 8);
    for (int64_t cb = 0; cb < C; cb += chunk) {
        const int64_t cc = C - cb < chunk ? C - cb : chunk;
        dim3 tgrid((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32));
        SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, tgrid, dim3(256), 0, Xq + cb, ld_q, Tq, 1, 0, cc, qc.as<double>(),
                  st_p.as<int32_t>() + cb, 0);
        PredictArgs pw = pa;
        pw.out = oc.as<double>();
        pw.oc_Tq = Tq;
        int nbc = nb;
        if ((int64_t)nbc > ((cc + 7) / 8) * 8) nbc = (int)(((cc + 7) / 8) * 8);
        const double* r = runs.as<double>() + cb * (int64_t)np;
        const double* xc = Xc.as<double>() + cb * T;
        const double* yy = yc.as<double>() + cb * T;
        const int32_t* sf = st_fit.as<int32_t>() + cb;
        int32_t* sp = st_p.as<int32_t>() + cb;
        const int32_t* od = odd.as<int32_t>() + cb;
        int rc = SD_OK;
        switch (K) {
            case 13: rc = launch_fused_k<13>(ctx, nbc, lds, r, np, od, xc, yy, qc.as<double>(), Tq, T, cc, sf, sp, worklist, work_count, cb, pw, skip_prob); break;
            case 15: rc = launch_fused_k<15>(ctx, nbc, lds, r, np, od, xc, yy, qc.as<double>(), Tq, T, cc, sf, sp, worklist, work_count, cb, pw, skip_prob); break;
            default: rc = launch_fused_k<17>(ctx, nbc, lds, r, np, od, xc, yy, qc.as<double>(), Tq, T, cc, sf, sp, worklist, work_count, cb, pw, skip_prob); break;
        }
        SD_TRY(rc);
        SD_LAUNCH(ctx, "analog_untranspose_kernel", analog_untranspose_kernel,
                  dim3((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32), skip_prob ? 2 : 3), dim3(256), 0, (const double*)oc.p, Tq, cc,
                  out + cb, ld_out, skip_prob);
    }
    int32_t nw = 0;
    SD_HIP(hipMemcpyAsync(&nw, work_count, sizeof(nw), hipMemcpyDeviceToHost, ctx->stream));
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "analog_status_public_kernel", analog_status_public_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                  (const int32_t*)st_fit.p, (const int32_t*)st_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
#ifdef SD_DEV
    if (sd_dev_env("SD_ANALOG_COUNT")) fprintf(stderr, "analog fit_predict: %d of %lld cells handed back to the split path\n", nw, (long long)C);
#endif
    if (nw == 0) return SD_OK;
    // cells handed back (ties among the training values or on a window boundary): the split path answers them.  Few: on packed
    // copies of their columns; many: the whole grid in place (the same numbers either way).
    if ((int64_t)nw * 2 > C) return fit_predict_split(ctx, X, y, ld, T, F, C, Xq, ld_q, Tq, k, kind, has_thresh, thresh, out, ld_out, nullptr);
    sd_scratch Xw, yw, Qw, Ow;
    SD_HIP(Xw.alloc(ctx, sizeof(double) * (size_t)T * nw));
    SD_HIP(yw.alloc(ctx, sizeof(double) * (size_t)T * nw));
    SD_HIP(Qw.alloc(ctx, sizeof(double) * (size_t)Tq * nw));
    SD_HIP(Ow.alloc(ctx, sizeof(double) * (size_t)Tq * 3 * nw));
    auto blocks = [&](int64_t total) { return dim3((unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)ctx->cu_count * 16)); };
    SD_LAUNCH(ctx, "analog_gather_cells_kernel", analog_gather_cells_kernel, blocks(T * nw), dim3(256), 0, X, ld, T, (const int32_t*)worklist, (int64_t)nw, Xw.as<double>());
    SD_LAUNCH(ctx, "analog_gather_cells_kernel", analog_gather_cells_kernel, blocks(T * nw), dim3(256), 0, y, ld, T, (const int32_t*)worklist, (int64_t)nw, yw.as<double>());
    SD_LAUNCH(ctx, "analog_gather_cells_kernel", analog_gather_cells_kernel, blocks(Tq * nw), dim3(256), 0, Xq, ld_q, Tq, (const int32_t*)worklist, (int64_t)nw, Qw.as<double>());
    SD_TRY(fit_predict_split(ctx, Xw.as<double>(), yw.as<double>(), nw, T, 1, nw, Qw.as<double>(), nw, Tq, k, kind, has_thresh, thresh, Ow.as<double>(), nw, nullptr));
    SD_LAUNCH(ctx, "analog_scatter_cells_kernel", analog_scatter_cells_kernel, blocks(3 * Tq * nw), dim3(256), 0, (const double*)Ow.p, 3 * Tq, (const int32_t*)worklist,
              (int64_t)nw, out, ld_out);
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

}  // namespace

extern "C" {

int sd_analog_state_destroy(sd_analog_state* st) {
    if (!st) return SD_OK;
    if (st->ctx) {
        (void)hipSetDevice(st->ctx->device);
        (void)hipStreamSynchronize(st->ctx->stream);
    }
    sd_pool_release(st->ctx, st->X);
    sd_pool_release(st->ctx, st->y);
    sd_pool_release(st->ctx, st->status);
    sd_pool_release(st->ctx, st->xs);
    sd_pool_release(st->ctx, st->xi);
    sd_pool_release(st->ctx, st->yx);
    sd_pool_release(st->ctx, st->pq);
    sd_pool_release(st->ctx, st->ybar);
    sd_pool_release(st->ctx, st->rx);
    sd_pool_release(st->ctx, st->xbar);
    sd_pool_release(st->ctx, st->ps);
    delete st;
    return SD_OK;
}

int sd_analog_state_info(const sd_analog_state* st, int64_t* T, int* F, int64_t* C) {
    SD_CHECK_ARG(st, "state is NULL");
    if (T) *T = st->T;
    if (F) *F = st->F;
    if (C) *C = st->C;
    return SD_OK;
}

int sd_analog_fit_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                      sd_analog_state** out) {
    SD_CHECK_ARG(ctx && X_dev && y_dev && out, "sd_analog_fit: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && ld >= C, "sd_analog_fit: bad sizes");
    SD_CHECK_ARG(F >= 1 && F <= kMaxF, "sd_analog_fit: F=%d outside [1,%d]", F, kMaxF);
    *out = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    sd_analog_state* st = new sd_analog_state();
    st->ctx = ctx;
    st->T = T;
    st->F = F;
    st->C = C;
    auto body = [&]() -> int {
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->X, sizeof(double) * (size_t)T * F * C));
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->y, sizeof(double) * (size_t)T * C));
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->status, sizeof(int32_t) * C));
        SD_HIP(hipMemsetAsync(st->status, 0, sizeof(int32_t) * C, ctx->stream));
        const size_t lds = (size_t)T * (sizeof(double) + sizeof(uint16_t));
        const bool f1_sorted = F == 1 && T <= 65535 && (lds <= ctx->lds_max || sort2_width(T, ctx->lds_max) != 0) &&
                               sizeof(double) * (size_t)(T + 1) <= ctx->lds_max;
        const int K2t = (f1_sorted && !sd_dev_env("SD_ANALOG_SORT1")) ? sort2_width(T, ctx->lds_max) : 0;
        bool tiled = K2t != 0 && tile_sort_applies(K2t, T, C, ctx->lds_max);
        sd_scratch runs_buf, odd_buf;
        int np_runs = 0;
        if (tiled) {
            const int64_t chunk = 64 * K2t;
            np_runs = (int)(((T + chunk - 1) / chunk) * chunk);
            if (runs_buf.alloc(ctx, sizeof(double) * (size_t)np_runs * (size_t)C) != hipSuccess) {  // no room for the runs: the two-transpose path
                (void)hipGetLastError();
                tiled = false;
            }
        }
        if (tiled) {
            // F == 1: one tile-shaped kernel makes the cell-major copies and the sorted runs of 64 * K keys (csrc: analog_tile_sort_kernel)
            SD_HIP(odd_buf.alloc(ctx, sizeof(int32_t) * (size_t)C));
            SD_HIP(hipMemsetAsync(odd_buf.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
            SD_TRY(launch_tile_sort(ctx, K2t, X_dev, y_dev, ld, T, C, st->X, st->y, runs_buf.as<double>(), np_runs, st->status,
                                    odd_buf.as<int32_t>()));
        } else {
            dim3 grid((unsigned)((C + 31) / 32), (unsigned)((T + 31) / 32));
            for (int f = 0; f < F; ++f)
                SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, grid, dim3(256), 0, X_dev, ld, T, F, f, C,
                          st->X, st->status, 1);
            SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, grid, dim3(256), 0, y_dev, ld, T, 1, 0, C,
                      st->y, st->status, 0);
        }
        if (F == 1 && T <= 65535 && (lds <= ctx->lds_max || sort2_width(T, ctx->lds_max) != 0) &&
            sizeof(double) * (size_t)(T + 1) <= ctx->lds_max) {
            // sorted view for the 1-D fast path: values, original indices, and y in the same order
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->xs, sizeof(double) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->xi, sizeof(int32_t) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->yx, sizeof(double) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->ybar, sizeof(double) * C));
            const int K2 = sd_dev_env("SD_ANALOG_SORT1") ? 0 : sort2_width(T, ctx->lds_max);
            if (K2 == 0)
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_sort_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (K2 != 0) {
                // (no prefix sums yet: the BASELINE path -- analog_f1_mean3_kernel -- builds its own on chip; the kernels
                // that read them from memory get them from ensure_prefix_sums on their first call)
                Sort2Args a{st->X, T, 0, st->y, T, C, st->xs, st->xi, st->yx, nullptr, st->ybar};
                if (tiled && K2 == K2t) {
                    a.runs = runs_buf.as<double>();
                    a.np_runs = np_runs;
                    a.odd_flags = odd_buf.as<int32_t>();
                }
                SD_TRY(launch_sort2_width(ctx, K2, a));
            } else {
                SD_HIP(sd_pool_malloc(ctx, (void**)&st->pq, sizeof(double) * 2 * (size_t)(T + 1) * C));
                int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * 4);
                SD_LAUNCH(ctx, "analog_sort_kernel", analog_sort_kernel, dim3(nb), dim3(1024), lds, (const double*)st->X,
                          (const double*)st->y, T, C, st->xs, st->xi, st->yx);
                const int nbp = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * 2);
                SD_LAUNCH(ctx, "analog_prefix_kernel", analog_prefix_kernel, dim3(nbp), dim3(1024), 0, (const double*)st->yx, T, C,
                          st->pq, st->ybar, 0);
            }
            SD_HIP(hipStreamSynchronize(ctx->stream));
        }
        const int Ks = F > 1 && sd_dev_env("SD_ANALOG_NOSLAB") == nullptr ? sort2_width(T, ctx->lds_max) : 0;
        if (Ks != 0) {
            // F > 1: training points in feature-0 order for the slab search (analog_slab_predict_kernel)
            sd_scratch keys;
            SD_HIP(keys.alloc(ctx, sizeof(double) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->xi, sizeof(int32_t) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->ps, sizeof(double) * (size_t)T * F * C));
            const Sort2Args a{st->X, (int64_t)F * T, 1, nullptr, T, C, keys.as<double>(), st->xi,
                              nullptr, nullptr, nullptr};
            SD_TRY(launch_sort2_width(ctx, Ks, a));
            SD_LAUNCH(ctx, "analog_gather_sorted_kernel", analog_gather_sorted_kernel,
                      dim3((unsigned)std::min<int64_t>(C, (int64_t)ctx->cu_count * 64)), dim3(256), 0, (const double*)st->X,
                      (const int32_t*)st->xi, T, F, C, st->ps);
            SD_HIP(hipStreamSynchronize(ctx->stream));
        }
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    };
    int rc = body();
    if (rc != SD_OK) {
        sd_analog_state_destroy(st);
        return rc;
    }
    *out = st;
    return SD_OK;
}

int sd_analog_fit(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, sd_analog_state** out) {
    SD_CHECK_ARG(ctx && X && y && out, "sd_analog_fit: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && F >= 1, "sd_analog_fit: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy;
    SD_HIP(dX.alloc(ctx, sizeof(double) * (size_t)T * F * C));
    SD_HIP(dy.alloc(ctx, sizeof(double) * (size_t)T * C));
    SD_TRY(sd_copy_h2d(ctx, dX.p, X, sizeof(double) * (size_t)T * F * C));
    SD_TRY(sd_copy_h2d(ctx, dy.p, y, sizeof(double) * (size_t)T * C));
    return sd_analog_fit_dev(ctx, dX.as<double>(), dy.as<double>(), C, T, F, C, out);
}

int sd_analog_predict_dev(sd_ctx* ctx, const sd_analog_state* st, const double* Xq_dev, int64_t ld, int64_t Tq, int k,
                          int kind, int has_thresh, double thresh, const int32_t* sample_inds_dev, double* out_dev,
                          int64_t ld_out, int64_t* inds_dev, double* dist_dev, int32_t* cell_status) {
    return predict_common(0, ctx, st, Xq_dev, ld, Tq, k, kind, has_thresh, thresh, sample_inds_dev, ld, out_dev, ld_out,
                          inds_dev, dist_dev, cell_status);
}

int sd_analog_predict(sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int kind,
                      int has_thresh, double thresh, const int32_t* sample_inds, double* out, int64_t* inds,
                      double* dist, int32_t* cell_status) {
    return predict_host(0, ctx, st, Xq, Tq, k, kind, has_thresh, thresh, sample_inds, out, inds, dist, cell_status);
}

int sd_analog_fit_predict_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                              const double* Xq_dev, int64_t ld_q, int64_t Tq, int k, int kind, int has_thresh, double thresh,
                              double* out_dev, int64_t ld_out, int32_t* cell_status) {
    return fit_predict_dev(ctx, X_dev, y_dev, ld, T, F, C, Xq_dev, ld_q, Tq, k, kind, has_thresh, thresh, out_dev, ld_out, cell_status);
}

int sd_analog_fit_predict(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, const double* Xq, int64_t Tq, int k,
                          int kind, int has_thresh, double thresh, double* out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && X && y && Xq && out, "sd_analog_fit_predict: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && Tq > 0 && F >= 1, "sd_analog_fit_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy, dq, dout;
    const size_t xb = sizeof(double) * (size_t)T * F * C, yb = sizeof(double) * (size_t)T * C, qb = sizeof(double) * (size_t)Tq * F * C,
                 ob = sizeof(double) * (size_t)Tq * 3 * C;
    SD_HIP(dX.alloc(ctx, xb));
    SD_HIP(dy.alloc(ctx, yb));
    SD_HIP(dq.alloc(ctx, qb));
    SD_HIP(dout.alloc(ctx, ob));
    SD_TRY(sd_copy_h2d(ctx, dX.p, X, xb));
    SD_TRY(sd_copy_h2d(ctx, dy.p, y, yb));
    SD_TRY(sd_copy_h2d(ctx, dq.p, Xq, qb));
    SD_TRY(fit_predict_dev(ctx, dX.as<double>(), dy.as<double>(), C, T, F, C, dq.as<double>(), C, Tq, k, kind, has_thresh, thresh, dout.as<double>(), C,
                           cell_status));
    return sd_copy_d2h(ctx, out, dout.p, ob);
}

int sd_analogreg_predict_dev(sd_ctx* ctx, const sd_analog_state* st, const double* Xq_dev, int64_t ld, int64_t Tq,
                             int k, int has_thresh, double thresh, double* out_dev, int64_t ld_out, int32_t* cell_status) {
    return predict_common(1, ctx, st, Xq_dev, ld, Tq, k, SD_ANALOG_MEAN, has_thresh ? 1 : 0, thresh, nullptr, ld, out_dev, ld_out, nullptr,
                          nullptr, cell_status);
}

int sd_analogreg_predict(sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int has_thresh, double thresh,
                         double* out, int32_t* cell_status) {
    return predict_host(1, ctx, st, Xq, Tq, k, SD_ANALOG_MEAN, has_thresh ? 1 : 0, thresh, nullptr, out, nullptr, nullptr, cell_status);
}

}  // extern "C"
