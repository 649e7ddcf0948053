// sd_analog_fit.h -- part of the translation unit csrc/sd_analog.hip (included there, inside its unnamed namespace; not a
// stand-alone header).  fit: staging transposes, the tile-shaped first stage, per-cell sorts (sorted view xs / xi / yx), prefix sums.

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
    // The series goes through LDS (coalesced loads and stores; the serial part of the scan reads an odd number of
    // consecutive elements per thread: conflict-free), once for each of the two sums -- the same partial-sum order as the
    // version that had every thread walk its 8 * per bytes of global memory (8-byte requests 120 bytes apart: 42 ms per
    // 100 000 cells x 14 600 where this one is bound by its 35 GB of traffic).
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* e = reinterpret_cast<double*>(smem_raw);  // n + 1 doubles
    __shared__ double wsum[2][16];
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + nthr - 1) / nthr;  // consecutive elements per thread
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* yx = yx_all + c * T;
        double* pq = pq_all + 2 * c * (T + 1);
        const int beg = tid * per < n ? tid * per : n, end = beg + per < n ? beg + per : n;
        __syncthreads();
        for (int i = tid; i < n; i += nthr) e[i] = yx[i];
        __syncthreads();
        // mean of y
        double s = 0.0;
        for (int i = beg; i < end; ++i) s += e[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) wsum[0][wave] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += wsum[0][w];
        const double ybar = keep_ybar ? ybar_all[c] : tot / (double)n;
        if (tid == 0 && !keep_ybar) ybar_all[c] = ybar;
        // per-thread totals of d and d^2, exclusive scan across the workgroup
        double a = 0.0, b = 0.0;
        for (int i = beg; i < end; ++i) {
            const double d = e[i] - ybar;
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
        const double ra0 = oa + (ia - a), rb0 = ob + (ib - b);  // exclusive prefixes at this thread's first element
        // pass 1: sums of d in place of the values, out as pq[i].x; the centred values are kept in registers for pass 2
        constexpr int kMaxPer = 20;  // (T <= 1024 * 20: what the callers send here; longer series re-read the values)
        double dk[kMaxPer];
        double ra = ra0;
#pragma unroll
        for (int t = 0; t < kMaxPer; ++t) {
            const int i = beg + t;
            const bool in = t < per && i < end;
            const double d = in ? e[i] - ybar : 0.0;
            dk[t] = d;
            if (in) e[i] = ra;
            ra += d;
        }
        if (per > kMaxPer)
            for (int i = beg + kMaxPer; i < end; ++i) {
                const double d = e[i] - ybar;
                e[i] = ra;
                ra += d;
            }
        if (end == n && beg < n) e[n] = ra;
        if (n == 0 && tid == 0) e[0] = 0.0;
        __syncthreads();
        for (int i = tid; i <= n; i += nthr) pq[2 * (int64_t)i] = e[i];
        __syncthreads();
        // pass 2: sums of d^2
        double rb = rb0;
        if (per <= kMaxPer) {
#pragma unroll
            for (int t = 0; t < kMaxPer; ++t) {
                const int i = beg + t;
                if (t < per && i < end) e[i] = rb;
                rb += dk[t] * dk[t];
            }
        } else {
            for (int i = tid; i < n; i += nthr) e[i] = yx[i];
            __syncthreads();
            for (int i = beg; i < end; ++i) {
                const double d = e[i] - ybar;
                e[i] = rb;
                rb += d * d;
            }
        }
        if (end == n && beg < n) e[n] = rb;
        __syncthreads();
        for (int i = tid; i <= n; i += nthr) pq[2 * (int64_t)i + 1] = e[i];
    }
}

// the same without LDS staging (series too long for the LDS: any T)
__global__ void __launch_bounds__(1024) analog_prefix_direct_kernel(const double* __restrict__ yx_all, int64_t T, int64_t C,
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
