// Quantile-mapping regressors of the reference (skdownscale/pointwise_models/quantile.py), batched over the cell
// axis: QuantileMappingReressor (160-395) and EquidistantCdfMatcher (556-636), every extrapolate mode.
//
// fit    : per cell np.sort(X), np.sort(y) (quantile.py:217-218 via 352-356) -> xs[C][T], ys[C][T]
// predict: QMR  x -> p = interp(x, xs, pp) -> interp(p, pp, ys)                       (quantile.py:247-249, 268-269)
//          ECM  rank r of x among the new series -> p = pp_m[r] -> interp(p, pp, ys) + (x - interp(p, pp, xs))
//               (or * x / interp(p, pp, xs))                                          (quantile.py:612-623)
//          '1to1': samples beyond the fitted X range keep their offset to it          (quantile.py:277-310)
// pp = Cunnane plotting positions (quantile.py:23-43).  The reference works on extended CDFs of n + 2 points
// (quantile.py:312-387).  For extrapolate None / '1to1' the two extra points duplicate the ends, which np.interp's
// clamping makes equivalent to the plain arrays.  For 'min' / 'max' / 'both' the extra points are synthetic: position
// -+1e20 and the value of the least-squares line through the n_endpoints outermost (position, value) pairs there
// (~ -+1e21).  Samples inside the fitted range never touch them (same brackets, same results as None); samples beyond
// it are interpolated across them with the reference's own formula, slope * (x - xp[j]) + fp[j], an ill-conditioned
// expression (the 1e20 cancels: ~1e4 of absolute rounding noise in the position, ~1e5 in the result -- in the
// reference's outputs as well).
// np.interp arithmetic is spelled out: last xp <= x, exact hit -> fp[j], slope * (x - xp[j]) + fp[j] otherwise.
#include <algorithm>
#include <cstdlib>

#include "sd_internal.h"
#include "sd_sortnet.h"
#include "sd_wave.h"

struct sd_qm_state {
    sd_ctx* ctx = nullptr;
    int64_t T = 0, C = 0;
    double* xs = nullptr;       // device [C][T] sorted X
    double* ys = nullptr;       // device [C][T] sorted y
    int32_t* status = nullptr;  // device [C] internal bitmask
};

namespace {

__device__ __forceinline__ bool qm_finite(double v) { return (__double_as_longlong(v) & 0x7ff0000000000000ll) != 0x7ff0000000000000ll; }

// [T, ld] -> [C][T] through a 32x33 LDS tile, with mask / finite bookkeeping (core.py:35-37, base.py:18-20)
__global__ void __launch_bounds__(256) qm_transpose_kernel(const double* __restrict__ src, int64_t ld, int64_t T, int64_t C,
                                                           double* __restrict__ dst, int32_t* status, int set_mask) {
    __shared__ double tile[32][33];
    const int64_t t0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int r = ty; r < 32; r += 8) {
        const int64_t t = t0 + r, c = c0 + tx;
        double v = 0.0;
        if (t < T && c < C) {
            v = src[t * ld + c];
            if (set_mask && t == 0 && v != v) atomicOr(&status[c], SDI_MASKED);
            if (!qm_finite(v)) atomicOr(&status[c], SDI_NONFINITE);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t c = c0 + r, t = t0 + tx;
        if (t < T && c < C) dst[c * T + t] = tile[tx][r];
    }
}

// [C][T] -> [T, ld]; cells with a non-zero status (fit or predict) are written as NaN
__global__ void __launch_bounds__(256) qm_untranspose_kernel(const double* __restrict__ oc, int64_t T, int64_t C,
                                                             double* __restrict__ out, int64_t ld,
                                                             const int32_t* __restrict__ s0, const int32_t* __restrict__ s1) {
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int r = ty; r < 32; r += 8) {
        const int64_t c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < C && t < T) ? oc[c * T + t] : 0.0;
    }
    __syncthreads();
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int r = ty; r < 32; r += 8) {
        const int64_t t = t0 + r, c = c0 + tx;
        if (t < T && c < C) out[t * ld + c] = (s0[c] | s1[c]) ? nan : tile[tx][r];
    }
}

// per-cell np.sort: one 1024-thread workgroup per cell, K consecutive samples per thread (sd_sortnet.h), in place
template <int K>
__global__ void __launch_bounds__(1024) qm_sort_kernel(double* __restrict__ data /* [C][T] */, int64_t T, int64_t C) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x;
    const int np = (n + K - 1) / K * K;
    double* buf = reinterpret_cast<double*>(smem_raw);  // np + 1 doubles
    int* xch = reinterpret_cast<int*>(buf + np + 1);     // nthr + 1 ints
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        double* x = data + c * T;
        __syncthreads();
        for (int i = tid; i <= np; i += nthr) buf[i] = i < n ? x[i] : inf;
        __syncthreads();
        double v[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            v[i] = buf[j < np ? j : np];
        }
        __syncthreads();
        sdsort::block_merge_sort<K>(v, buf, np, xch, tid, nthr);
        for (int i = tid; i < n; i += nthr) x[i] = buf[i];
    }
}


// ---- fit, tile-shaped first stage (round 6) -----------------------------------------------------------------------------
// qm_tile_runs_kernel<K>: one 512-thread workgroup = 8 adjacent cells x one chunk of 64 * K consecutive time steps of ONE
// time-major field, read as 64-byte row fragments (the geometry of the BCSD kernels and of analog_tile_sort_kernel, sd_wave.h).
// It replaces the staging transpose (mask / finite bookkeeping included) and the first six rounds of qm_sort_kernel: while
// the tile is on chip every wave sorts its cell's chunk (sdw::sort_segment) and the sorted runs go out cell-major, 512
// consecutive bytes per wave store.  qm_merge_runs_kernel<K> -- one 1 024-thread workgroup per cell -- then only merges the
// at most 16 runs (sdsort::block_merge_rounds from round 6).  np.sort needs no index: plain float64 keys, pads = +inf.
// Non-finite samples sort as 0 (their cell is flagged and answers NaN; NaNs must not enter the min / max networks).
template <int K>
__global__ void __launch_bounds__(sdw::kThreads, 4) qm_tile_runs_kernel(const double* __restrict__ X, int64_t ld, int64_t T, int64_t C,
                                                                        int nchunks, double* __restrict__ runs, int64_t runs_stride,
                                                                        int32_t* status, int set_mask) {
    using namespace sdw;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int CHUNK = kWave * K;
    constexpr int NR = (CHUNK + kRowsPerPass - 1) / kRowsPerPass;
    constexpr int RS = CHUNK + 2 + ((4 - (CHUNK + 2) % 4) + 2) % 4;  // row stride: >= CHUNK + 1 slots, RS % 4 == 2
    double* const tile = reinterpret_cast<double*>(smem_raw) + kHeadDoubles;
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
    const bool vec = (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && cpair + 1 < C;
    double x0[NR], x1[NR];
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
    {
        double* d0 = tile + (2 * cp) * RS;
        double* d1 = d0 + RS;
        int bits0 = 0, bits1 = 0;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rr + k * kRowsPerPass;
            if (r < nq) {
                if (set_mask && r0 + r == 0) {  // core.py:35-37: the first sample of X is NaN
                    bits0 |= x0[k] != x0[k] ? SDI_MASKED : 0;
                    bits1 |= x1[k] != x1[k] ? SDI_MASKED : 0;
                }
                const bool f0 = qm_finite(x0[k]), f1 = qm_finite(x1[k]);
                bits0 |= f0 ? 0 : SDI_NONFINITE;
                bits1 |= f1 ? 0 : SDI_NONFINITE;
                d0[r] = f0 ? x0[k] : 0.0;
                d1[r] = f1 ? x1[k] : 0.0;
            }
        }
        if (bits0 && cpair < C) atomicOr(&status[cpair], bits0);
        if (bits1 && cpair + 1 < C) atomicOr(&status[cpair + 1], bits1);
    }
    __syncthreads();
    const int64_t c = c0 + wave;
    double* const row = tile + wave * RS;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double v[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int jl = K * lane + i;  // (lane stride K is odd: conflict-free)
        v[i] = jl < nq ? row[jl] : inf;
    }
    wave_fence();
    sort_segment<K>(v, row, CHUNK, lane);  // every slot of the chunk is an element: pads sort behind the data
    if (c < C) {
        double* dst = runs + c * runs_stride + r0;
#pragma unroll
        for (int i = 0; i < K; ++i) dst[lane + i * kWave] = row[lane + i * kWave];
    }
}

template <int K>
__global__ void __launch_bounds__(1024) qm_merge_runs_kernel(const double* __restrict__ runs, int np, int64_t T, int64_t C,
                                                             double* __restrict__ xs /* [C][T] */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* buf = reinterpret_cast<double*>(smem_raw);  // np + 1 doubles
    int* xch = reinterpret_cast<int*>(buf + np + 1);     // nthr + 1 ints
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* rc = runs + c * (int64_t)np;
        __syncthreads();
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
        __syncthreads();
        sdsort::block_merge_rounds<K>(buf, np, xch, tid, nthr, 6);
        double* dst = xs + c * T;
        for (int i = tid; i < n; i += nthr) dst[i] = buf[i];
    }
}

// rank of every sample of a cell's series in (value, index) order (np.argsort, stable): rank[C][T] as int32
template <int K>
__global__ void __launch_bounds__(1024) qm_rank_kernel(const double* __restrict__ data /* [C][T] */, int64_t T, int64_t C,
                                                       int32_t* __restrict__ rank) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x;
    const int np = (n + K - 1) / K * K;
    double* buf = reinterpret_cast<double*>(smem_raw);
    int* xch = reinterpret_cast<int*>(buf + np + 1);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* x = data + c * T;
        __syncthreads();
        for (int i = tid; i <= np; i += nthr) buf[i] = i < n ? x[i] : inf;
        __syncthreads();
        double v[K], orig[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            v[i] = buf[j < np ? j : np];
            orig[i] = v[i];
        }
        __syncthreads();
        sdsort::block_merge_sort<K>(v, buf, np, xch, tid, nthr);
        bool tie = false;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            tie |= j + 1 < n && buf[j] == buf[j + 1];
        }
        const bool ties = __syncthreads_or(tie) != 0;
        int lb[K];  // number of sorted values < x
        {
            int pos[K];
#pragma unroll
            for (int i = 0; i < K; ++i) pos[i] = -1;
#pragma unroll 1
            for (int len = n; len > 1;) {
                int half = len >> 1;
                if ((half & 15) == 0) --half;  // keep the probe strides off the LDS bank period
                len -= half;
#pragma unroll
                for (int i = 0; i < K; ++i) pos[i] += buf[pos[i] + half] < orig[i] ? half : 0;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) lb[i] = pos[i] + 1 + (buf[pos[i] + 1] < orig[i] ? 1 : 0);
        }
        if (!ties) {  // distinct values: lb is the rank
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * tid + i;
                if (j < n) rank[c * T + j] = lb[i];
            }
            continue;
        }
        // keys lb * 65536 + index: distinct integers < 2^32, ordered like (value, index); position = rank
        double key2[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            key2[i] = j < n ? (double)lb[i] * 65536.0 + (double)j : inf;
        }
        __syncthreads();
        sdsort::block_merge_sort<K>(key2, buf, np, xch, tid, nthr);
        int idx[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            idx[i] = j < n ? (int)((unsigned)buf[j] & 0xffffu) : -1;
        }
        __syncthreads();
        int* ibuf = reinterpret_cast<int*>(buf);
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (idx[i] >= 0) ibuf[idx[i]] = K * tid + i;
        __syncthreads();
        for (int i = tid; i < n; i += nthr) rank[c * T + i] = ibuf[i];
    }
}

constexpr double kAlpha = 0.4, kBeta = 0.4;
__device__ __forceinline__ double pp_denom(int n) { return ((double)n + 1.0 - kAlpha) - kBeta; }
__device__ __forceinline__ double pp_at(int i, double denom) { return ((double)(i + 1) - kAlpha) / denom; }
// The same quotient without the division (12 instructions, five of them quarter rate; qm_map_kernel needs ~8 per sample and was bound
// by them): q0 = a * r, q = q0 + (a - q0 * d) * r with r = RN(1 / d) and both products fused -- Markstein's correction step, which
// returns the correctly rounded quotient whenever q0 is within an ulp of it.  Not taken on trust: qm_ppcheck_kernel compares it with
// the division for every position of the grid (i = 0 .. n-1, the only arguments the kernels ever pass) before a launch uses it.
template <bool FAST>
__device__ __forceinline__ double ppq(int i, double denom, double rden) {
    const double a = (double)(i + 1) - kAlpha;
    if (!FAST) return a / denom;
    const double q0 = a * rden;
    return __builtin_fma(__builtin_fma(-q0, denom, a), rden, q0);
}
__global__ void __launch_bounds__(256) qm_ppcheck_kernel(int n, double denom, double rden, int32_t* __restrict__ mismatch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && __double_as_longlong(ppq<true>(i, denom, rden)) != __double_as_longlong(ppq<false>(i, denom, rden))) atomicOr(mismatch, 1);
}

constexpr double kSyntheticMin = -1e20, kSyntheticMax = 1e20;  // quantile.py:17-18
// value of the least-squares line through (pp[first + i], f[first + i]), i < e, at position x0: sklearn's
// LinearRegression().fit(...).predict(x0) = x0 * slope + (mean_y - mean_x * slope) (quantile.py:366-385)
__device__ inline double ols_value_at(const double* __restrict__ f, int first, int e, double denom, double x0) {
    double xm = 0.0, ym = 0.0;
    for (int i = 0; i < e; ++i) {
        xm += pp_at(first + i, denom);
        ym += f[first + i];
    }
    xm /= (double)e;
    ym /= (double)e;
    double sxx = 0.0, sxy = 0.0;
    for (int i = 0; i < e; ++i) {
        const double dx = pp_at(first + i, denom) - xm;
        sxx += dx * dx;
        sxy += dx * (f[first + i] - ym);
    }
    const double slope = sxx > 0.0 ? sxy / sxx : 0.0;
    return (ym - slope * xm) + slope * x0;
}

// np.interp(p, pp_grid(n), f[0..n)): the abscissae are the Cunnane positions, so the bracket index is analytic.
// ext_lo / ext_hi: the grid is preceded / followed by a synthetic node (kSyntheticMin, f_lo) / (kSyntheticMax, f_hi).
__device__ __forceinline__ double interp_on_grid(double p, int n, double denom, const double* __restrict__ f, bool ext_lo = false,
                                                 double f_lo = 0.0, bool ext_hi = false, double f_hi = 0.0) {
    if (p <= pp_at(0, denom)) {  // left clamp / exact hit of the first node
        if (!ext_lo || p == pp_at(0, denom)) return f[0];
        if (p <= kSyntheticMin) return f_lo;
        const double slope = (f[0] - f_lo) / (pp_at(0, denom) - kSyntheticMin);
        return slope * (p - kSyntheticMin) + f_lo;
    }
    if (p >= pp_at(n - 1, denom)) {
        if (!ext_hi || p == pp_at(n - 1, denom)) return f[n - 1];
        if (p >= kSyntheticMax) return f_hi;
        const double slope = (f_hi - f[n - 1]) / (kSyntheticMax - pp_at(n - 1, denom));
        return slope * (p - pp_at(n - 1, denom)) + f[n - 1];
    }
    int j = (int)floor(p * denom + kAlpha) - 1;
    j = j < 0 ? 0 : (j > n - 2 ? n - 2 : j);
    while (j > 0 && pp_at(j, denom) > p) --j;      // guard the analytic index by one step either way
    while (j < n - 2 && pp_at(j + 1, denom) <= p) ++j;
    const double x0 = pp_at(j, denom);
    if (x0 == p) return f[j];
    const double slope = (f[j + 1] - f[j]) / (pp_at(j + 1, denom) - x0);
    return slope * (p - x0) + f[j];
}

// interp_on_grid in two steps, so that the table reads of several samples can be in flight together: the bracket ...
constexpr int kMapQ = 4;
template <bool FAST>
__device__ __forceinline__ int grid_bracket(double p, int n, double denom, double rden) {
    const double pc = p > 0.0 ? (p < 2.0 ? p : 2.0) : 0.0;  // (tail positions reach +-1e20 / +-inf: first / last interval)
    int j = (int)floor(pc * denom + kAlpha) - 1;
    j = j < 0 ? 0 : (j > n - 2 ? n - 2 : j);
    while (j > 0 && ppq<FAST>(j, denom, rden) > p) --j;      // guard the analytic index by one step either way
    while (j < n - 2 && ppq<FAST>(j + 1, denom, rden) <= p) ++j;
    return j;
}
// ... and the value, given fa = f[j], fb = f[j + 1] of j = grid_bracket(p): for p at or beyond the ends of the grid the bracket is
// the first / last interval, whose outer value is the one interp_on_grid's clamped branches read
template <bool FAST>
__device__ __forceinline__ double interp_on_grid_with(double p, int n, double denom, double rden, int j, double fa, double fb, bool ext_lo,
                                                      double f_lo, bool ext_hi, double f_hi) {
    if (p <= ppq<FAST>(0, denom, rden)) {  // left clamp / exact hit of the first node (j == 0: fa = f[0])
        if (!ext_lo || p == ppq<FAST>(0, denom, rden)) return fa;
        if (p <= kSyntheticMin) return f_lo;
        const double slope = (fa - f_lo) / (ppq<FAST>(0, denom, rden) - kSyntheticMin);
        return slope * (p - kSyntheticMin) + f_lo;
    }
    if (p >= ppq<FAST>(n - 1, denom, rden)) {  // (j == n - 2: fb = f[n - 1])
        if (!ext_hi || p == ppq<FAST>(n - 1, denom, rden)) return fb;
        if (p >= kSyntheticMax) return f_hi;
        const double slope = (f_hi - fb) / (kSyntheticMax - ppq<FAST>(n - 1, denom, rden));
        return slope * (p - ppq<FAST>(n - 1, denom, rden)) + fb;
    }
    const double x0 = ppq<FAST>(j, denom, rden);
    if (x0 == p) return fa;
    const double slope = (fb - fa) / (ppq<FAST>(j + 1, denom, rden) - x0);
    return slope * (p - x0) + fa;
}

// model: 0 QuantileMappingReressor, 1 EquidistantCdfMatcher 'difference', 2 'ratio'.  One workgroup per cell, the cell's two
// sorted tables staged through ONE LDS array, one after the other (round 6; the round-3 kernel kept xs in LDS and read the
// brackets of ys -- EDCDF: of xs and ys -- from memory, two to four scattered 8-byte requests per sample, one sample in
// flight per thread):
//   generation 1, LDS = xs:  QMR  the value -> position search and p = interp(x, xs, pp);  EDCDF  x_train = interp(p, pp, xs)
//   generation 2, LDS = ys:  y = interp(p, pp, ys) and the result
// A thread carries one number per sample from the first generation to the second (p or x_train), kMapPer samples per thread
// and pass (QMR 16: one pass for series of up to 16 384 samples; EDCDF 8: it holds ranks and samples as well), kMapQ samples in flight at a time: the reads of a step are issued
// before its compares (sched_barrier: see window_starts_n in sd_analog_f1.h).
// a cell's table -> LDS, eight requests per thread in flight (a plain copy loop pays one memory round trip per element: the
// sixteen waves of the workgroup all wait at the same time, nothing hides it)
__device__ __forceinline__ void fill_table(double* __restrict__ dst, const double* __restrict__ src, int n, int tid, int nthr) {
#pragma unroll 1
    for (int base = 0; base < n; base += 8 * nthr) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthr + tid;
            v[u] = i < n ? src[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthr + tid;
            if (i < n) dst[i] = v[u];
        }
    }
}
#ifdef SD_DEV
__device__ long long sd_qm_trace[8 * 8];  // development library: phase clocks of the first 8 cells of workgroup 0 (SD_QM_TRACE)
#define SD_QSTAMP(slot)                                                                                                        \
    do {                                                                                                                       \
        if (blockIdx.x == 0 && threadIdx.x == 0 && traced < 8) sd_qm_trace[traced * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SD_QSTAMP(slot) do { } while (0)
#endif
// a wave-uniform double, pinned to scalar registers (the allocator otherwise keeps the cell's constants in vector registers)
__device__ __forceinline__ double qm_uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// values of the tail lines at the synthetic end points (quantile.py:366-385) of every cell: tails[c] = (vx_lo, vy_lo, vx_hi, vy_hi)
__global__ void __launch_bounds__(256) qm_tails_kernel(int mode, int n_end, const double* __restrict__ xs_all, const double* __restrict__ ys_all,
                                                       int64_t T, int64_t C, double* __restrict__ tails) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int n = (int)T;
    const double dn = pp_denom(n);
    const int e = n_end < n ? n_end : n;
    const double* xs = xs_all + c * T;
    const double* ys = ys_all + c * T;
    const bool ext_lo = (mode & SD_EXTRAP_MIN) != 0, ext_hi = (mode & SD_EXTRAP_MAX) != 0;
    tails[c * 4 + 0] = ext_lo ? ols_value_at(xs, 0, e, dn, kSyntheticMin) : 0.0;
    tails[c * 4 + 1] = ext_lo ? ols_value_at(ys, 0, e, dn, kSyntheticMin) : 0.0;
    tails[c * 4 + 2] = ext_hi ? ols_value_at(xs, n - e, e, dn, kSyntheticMax) : 0.0;
    tails[c * 4 + 3] = ext_hi ? ols_value_at(ys, n - e, e, dn, kSyntheticMax) : 0.0;
}

template <bool QMR, int kMapPer, bool FAST>
__global__ void __launch_bounds__(1024) qm_map_kernel(int model, int mode, int n_end, const double* __restrict__ qc /* [C][Tp] */,
                                                      const int32_t* __restrict__ rank /* [C][Tp] or null */,
                                                      const double* __restrict__ xs_all, const double* __restrict__ ys_all,
                                                      int64_t T, int64_t Tp64, int64_t C, double* __restrict__ oc /* [C][Tp] */,
                                                      double rdn_in, double rdm_in, const double* __restrict__ tails /* [C][4] or null */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xl = reinterpret_cast<double*>(smem_raw);  // n table values: xs, then ys
    const int n = (int)T, Tp = (int)Tp64, tid = threadIdx.x, nthr = blockDim.x;
    const double dn = qm_uniform(pp_denom(n)), dm = qm_uniform(pp_denom(Tp));
    const double rdn = qm_uniform(rdn_in), rdm = qm_uniform(rdm_in);  // RN(1 / dn), RN(1 / dm) from the host
    const double pp_first = qm_uniform(ppq<FAST>(0, dn, rdn)), pp_last = qm_uniform(ppq<FAST>(n - 1, dn, rdn));
    static_assert(kMapPer % kMapQ == 0, "whole batches");
#ifdef SD_DEV
    int traced = 0;
#endif
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        SD_QSTAMP(0);
        const double* __restrict__ xs = xs_all + c * T;
        const double* __restrict__ ys = ys_all + c * T;
        const double* __restrict__ qrow = qc + c * Tp64;
        const int32_t* __restrict__ rrow = QMR ? nullptr : rank + c * Tp64;
        double* __restrict__ orow = oc + c * Tp64;
        const double x_min = qm_uniform(xs[0]), x_max = qm_uniform(xs[n - 1]), y_min = qm_uniform(ys[0]), y_max = qm_uniform(ys[n - 1]);
        // synthetic end points of the extended CDFs (quantile.py:366-385); every thread computes the same four numbers
        const bool ext_lo = (mode & SD_EXTRAP_MIN) != 0, ext_hi = (mode & SD_EXTRAP_MAX) != 0, one_to_one = (mode & SD_EXTRAP_1TO1) != 0;
        (void)n_end;  // (qm_tails_kernel has worked the tail lines out: 80 dependent memory round trips per cell if every thread does)
        const double vx_lo = qm_uniform(ext_lo ? tails[c * 4 + 0] : 0.0), vy_lo = qm_uniform(ext_lo ? tails[c * 4 + 1] : 0.0);
        const double vx_hi = qm_uniform(ext_hi ? tails[c * 4 + 2] : 0.0), vy_hi = qm_uniform(ext_hi ? tails[c * 4 + 3] : 0.0);
        for (int pass0 = 0; pass0 < Tp; pass0 += nthr * kMapPer) {
            // the thread's samples (EDCDF: their ranks in the new series) are requested before the table: one round trip for both
            double xq[kMapPer];
            int rk[kMapPer];
#pragma unroll
            for (int i = 0; i < kMapPer; ++i) {
                const int tq = pass0 + i * nthr + tid;
                if (QMR) xq[i] = tq < Tp ? qrow[tq] : x_min;
                else rk[i] = tq < Tp ? rrow[tq] : 0;
            }
            SD_QSTAMP(1);
            __syncthreads();  // (the previous pass / cell has left the LDS array)
            fill_table(xl, xs, n, tid, nthr);
            __syncthreads();
            SD_QSTAMP(2);
            double carry[kMapPer];  // QMR: p; EDCDF: x_train
#pragma unroll
            for (int b0 = 0; b0 < kMapPer; b0 += kMapQ) {
                if (QMR) {
                    double x[kMapQ];
                    int pos[kMapQ];  // last index known to hold a value <= x
#pragma unroll
                    for (int i = 0; i < kMapQ; ++i) {
                        x[i] = xq[b0 + i];
                        pos[i] = -1;
                    }
#pragma unroll 1
                    for (int len = n; len > 1;) {
                        int half = len >> 1;
                        if ((half & 15) == 0) --half;  // keep the probe strides off the LDS bank period
                        len -= half;
                        double a[kMapQ];
#pragma unroll
                        for (int i = 0; i < kMapQ; ++i) a[i] = xl[pos[i] + half];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < kMapQ; ++i) pos[i] += a[i] <= x[i] ? half : 0;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    double xa[kMapQ], xb[kMapQ];
                    int jx[kMapQ];
#pragma unroll
                    for (int i = 0; i < kMapQ; ++i) {
                        const double nx = xl[pos[i] + 1 < n ? pos[i] + 1 : n - 1];
                        int j = pos[i] + (nx <= x[i] ? 1 : 0);
                        j = j < 0 ? 0 : (j > n - 2 ? n - 2 : j);  // (in [0, n-2] already for a sample inside the fitted range)
                        jx[i] = j;
                    }
#pragma unroll
                    for (int i = 0; i < kMapQ; ++i) {
                        xa[i] = xl[jx[i]];
                        xb[i] = xl[jx[i] + 1];
                    }
#pragma unroll
                    for (int i = 0; i < kMapQ; ++i) {
                        // p = np.interp(x, xs, pp): j = last index with xs[j] <= x
                        const double xv = x[i];
                        double pv;
                        if (xv < x_min) {
                            if (ext_lo) {  // bracket (synthetic node, first value); below the synthetic node: left = -inf (quantile.py:245)
                                // (tied end points give a level tail line: vx_lo == x_min, every x < x_min is "below the node")
                                if (xv < vx_lo || !(x_min > vx_lo)) {
                                    pv = -__builtin_inf();
                                } else {
                                    const double slope = (pp_first - kSyntheticMin) / (x_min - vx_lo);
                                    pv = slope * (xv - vx_lo) + kSyntheticMin;
                                }
                            } else {
                                pv = pp_first;
                            }
                        } else if (xv >= x_max) {
                            if (ext_hi && xv > x_max) {
                                if (xv > vx_hi || !(vx_hi > x_max)) {
                                    pv = __builtin_inf();
                                } else {
                                    const double slope = (kSyntheticMax - pp_last) / (vx_hi - x_max);
                                    pv = slope * (xv - x_max) + pp_last;
                                }
                            } else {
                                pv = pp_last;
                            }
                        } else {
                            const int j = jx[i];
                            if (xa[i] == xv) {
                                pv = ppq<FAST>(j, dn, rdn);
                            } else {
                                const double slope = (ppq<FAST>(j + 1, dn, rdn) - ppq<FAST>(j, dn, rdn)) / (xb[i] - xa[i]);
                                pv = slope * (xv - xa[i]) + ppq<FAST>(j, dn, rdn);
                            }
                        }
                        carry[b0 + i] = pv;
                    }
                } else {
                    double p[kMapQ], ta[kMapQ], tb[kMapQ];
                    int jb[kMapQ];
#pragma unroll
                    for (int i = 0; i < kMapQ; ++i) {
                        p[i] = ppq<FAST>(rk[b0 + i], dm, rdm);  // plotting position of x within the new series
                        jb[i] = grid_bracket<FAST>(p[i], n, dn, rdn);
                    }
#pragma unroll
                    for (int i = 0; i < kMapQ; ++i) {
                        ta[i] = xl[jb[i]];
                        tb[i] = xl[jb[i] + 1];
                    }
#pragma unroll
                    for (int i = 0; i < kMapQ; ++i)
                        carry[b0 + i] = interp_on_grid_with<FAST>(p[i], n, dn, rdn, jb[i], ta[i], tb[i], ext_lo, vx_lo, ext_hi, vx_hi);  // quantile.py:613
                }
            }
            if (!QMR) {  // (the samples themselves enter in the second generation only)
#pragma unroll
                for (int i = 0; i < kMapPer; ++i) {
                    const int tq = pass0 + i * nthr + tid;
                    xq[i] = tq < Tp ? qrow[tq] : x_min;
                }
            }
            SD_QSTAMP(3);
            __syncthreads();
            SD_QSTAMP(4);
            fill_table(xl, ys, n, tid, nthr);
            __syncthreads();
            SD_QSTAMP(5);
#pragma unroll
            for (int b0 = 0; b0 < kMapPer; b0 += kMapQ) {
                double p[kMapQ], ya[kMapQ], yb[kMapQ];
                int jb[kMapQ];
#pragma unroll
                for (int i = 0; i < kMapQ; ++i) {
                    p[i] = QMR ? carry[b0 + i] : ppq<FAST>(rk[b0 + i], dm, rdm);
                    jb[i] = grid_bracket<FAST>(p[i], n, dn, rdn);
                }
#pragma unroll
                for (int i = 0; i < kMapQ; ++i) {
                    ya[i] = xl[jb[i]];
                    yb[i] = xl[jb[i] + 1];
                }
#pragma unroll
                for (int i = 0; i < kMapQ; ++i) {
                    const int tq = pass0 + (b0 + i) * nthr + tid;
                    if (tq >= Tp) continue;
                    const double xv = xq[b0 + i];
                    const double y_map = interp_on_grid_with<FAST>(p[i], n, dn, rdn, jb[i], ya[i], yb[i], ext_lo, vy_lo, ext_hi, vy_hi);  // quantile.py:268-269
                    double res = y_map;
                    if (!QMR) {
                        const double x_train = carry[b0 + i];
                        res = model == 1 ? y_map + (xv - x_train) : y_map * (xv / x_train);  // quantile.py:616-623
                    }
                    if (one_to_one) {  // quantile.py:277-310 (fit X and y have the same length)
                        if (xv > x_max) res = y_max + (xv - x_max);
                        if (xv < x_min) res = y_min + (xv - x_min);
                    }
                    orow[tq] = res;
                }
            }
            SD_QSTAMP(6);
        }
#ifdef SD_DEV
        ++traced;
#endif
    }
}

// CunnaneTransformer.transform / inverse_transform (quantile.py:465-545) for every sample of a cell.  One workgroup
// per cell; the forward direction keeps the sorted fit values in LDS for the value -> index search, the inverse
// direction finds its bracket analytically on the Cunnane grid.  Tails (inverse): centred least squares through the
// first / last e = min(n_endpoints, n) (position, value) pairs, as sklearn's LinearRegression solves it.
__global__ void __launch_bounds__(1024) qm_cunnane_kernel(int direction, int ext_lo, int ext_hi, int n_end,
                                                          const double* __restrict__ qc /* [C][Tp] */,
                                                          const double* __restrict__ xs_all, int64_t T, int64_t Tp, int64_t C,
                                                          double* __restrict__ oc /* [C][Tp] */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xl = reinterpret_cast<double*>(smem_raw);  // forward: n sorted fit values
    __shared__ double line[4];                          // inverse: (slope, intercept) of the lower and upper tail
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x;
    const double dn = pp_denom(n);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* xs = xs_all + c * T;
        __syncthreads();
        if (direction == 0) {
            for (int i = tid; i < n; i += nthr) xl[i] = xs[i];
        } else if (tid < 2 && (tid == 0 ? ext_lo : ext_hi)) {
            const int e = n_end < n ? n_end : n, first = tid == 0 ? 0 : n - e;
            double pm = 0.0, vm = 0.0;
            for (int i = 0; i < e; ++i) {
                pm += pp_at(first + i, dn);
                vm += xs[first + i];
            }
            pm /= (double)e;
            vm /= (double)e;
            double spp = 0.0, spv = 0.0;
            for (int i = 0; i < e; ++i) {
                const double dp = pp_at(first + i, dn) - pm;
                spp += dp * dp;
                spv += dp * (xs[first + i] - vm);
            }
            const double slope = spp > 0.0 ? spv / spp : 0.0;
            line[2 * tid] = slope;
            line[2 * tid + 1] = vm - slope * pm;
        }
        __syncthreads();
        const double x_min = xs[0], x_max = xs[n - 1];
        for (int64_t tq = tid; tq < Tp; tq += nthr) {
            const double x = qc[c * Tp + tq];
            double res;
            if (direction == 0) {
                if (x != x) {
                    res = x;
                } else if (x < x_min) {
                    res = ext_lo ? -inf : pp_at(0, dn);
                } else if (x >= x_max) {
                    res = (x > x_max && ext_hi) ? inf : pp_at(n - 1, dn);
                } else {
                    int pos = -1;  // last index known to hold a value <= x
                    for (int len = n; len > 1;) {
                        int half = len >> 1;
                        if ((half & 15) == 0) --half;
                        len -= half;
                        pos += xl[pos + half] <= x ? half : 0;
                    }
                    const int j = pos + (xl[pos + 1] <= x ? 1 : 0);  // in [0, n-2] here
                    const double x0 = xl[j];
                    if (x0 == x) {
                        res = pp_at(j, dn);
                    } else {
                        const double slope = (pp_at(j + 1, dn) - pp_at(j, dn)) / (xl[j + 1] - x0);
                        res = slope * (x - x0) + pp_at(j, dn);
                    }
                }
            } else {
                if (x != x) res = x;
                else if (x < pp_at(0, dn) && ext_lo) res = line[0] * x + line[1];
                else if (x > pp_at(n - 1, dn) && ext_hi) res = line[2] * x + line[3];
                else res = interp_on_grid(x, n, dn, xs);
            }
            oc[c * Tp + tq] = res;
        }
    }
}

__global__ void __launch_bounds__(256) qm_status_public_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                               int64_t C, int32_t* __restrict__ outp) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const int32_t bits = a[c] | (b ? b[c] : 0);
        outp[c] = (bits & SDI_MASKED) ? SD_CELL_MASKED : (bits & SDI_NONFINITE) ? SD_CELL_NONFINITE : SD_CELL_OK;
    }
}

int sort_width(int64_t T, size_t lds_max) {
    const int widths[] = {1, 3, 5, 9, 13, 15, 17, 19};
    for (int K : widths) {
        const int64_t np = (T + K - 1) / K * K;
        if (T <= (int64_t)1024 * K && sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025 <= lds_max) return K;
    }
    return 0;
}

template <int K>
int launch_sort(sd_ctx* ctx, double* data, int64_t T, int64_t C) {
    const int np = (int)((T + K - 1) / K * K);
    const size_t lds = sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qm_sort_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * 4);
    SD_LAUNCH(ctx, "qm_sort_kernel", qm_sort_kernel<K>, dim3(nb), dim3(1024), lds, data, T, C);
    return SD_OK;
}

template <int K>
int launch_rank(sd_ctx* ctx, const double* data, int64_t T, int64_t C, int32_t* rank) {
    const int np = (int)((T + K - 1) / K * K);
    const size_t lds = sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qm_rank_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * 4);
    SD_LAUNCH(ctx, "qm_rank_kernel", qm_rank_kernel<K>, dim3(nb), dim3(1024), lds, data, T, C, rank);
    return SD_OK;
}


// widths of the tile-shaped fit stage: at most 16 runs of 64 * K samples, merged by a 1 024-thread workgroup
int tile_runs_width(int64_t T, size_t lds_max) {
    if (sd_dev_env("SD_QM_NOTILE") != nullptr) return 0;
    const int widths[] = {13, 15, 17};
    for (int K : widths) {
        const int64_t chunk = 64 * K, nchunks = (T + chunk - 1) / chunk;
        if (nchunks <= 16 && sizeof(double) * (size_t)(nchunks * chunk + 1) + sizeof(int) * 1025 <= lds_max) return K;
    }
    return 0;
}

// np.sort of every cell's series of one time-major field -> xs [C][T] (plus mask / finite bookkeeping)
template <int K>
int launch_tile_sort(sd_ctx* ctx, const double* X_dev, int64_t ld, int64_t T, int64_t C, double* xs, int32_t* status, int set_mask,
                     double* runs) {
    constexpr int CHUNK = 64 * K;
    constexpr int RS = CHUNK + 2 + ((4 - (CHUNK + 2) % 4) + 2) % 4;
    const int nchunks = (int)((T + CHUNK - 1) / CHUNK);
    const int np = nchunks * CHUNK;
    const size_t lds_t = sizeof(double) * ((size_t)sdw::kW * RS + sdw::kHeadDoubles);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qm_tile_runs_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    const int64_t ntiles = (C + sdw::kW - 1) / sdw::kW, tx = (ntiles + 7) / 8;
    const int64_t nblocks = 8 * tx * nchunks;
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "sd_qm_fit: grid too large");
    SD_LAUNCH(ctx, "qm_tile_runs_kernel", qm_tile_runs_kernel<K>, dim3((unsigned)nblocks), dim3(sdw::kThreads), lds_t, X_dev, ld, T, C, nchunks,
              runs, (int64_t)np, status, set_mask);
    const size_t lds_m = sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qm_merge_runs_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    const int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * 4);
    SD_LAUNCH(ctx, "qm_merge_runs_kernel", qm_merge_runs_kernel<K>, dim3(nb), dim3(1024), lds_m, (const double*)runs, np, T, C, xs);
    return SD_OK;
}
int launch_tile_sort_width(sd_ctx* ctx, int K, const double* X_dev, int64_t ld, int64_t T, int64_t C, double* xs, int32_t* status, int set_mask,
                           double* runs) {
    switch (K) {
        case 13: return launch_tile_sort<13>(ctx, X_dev, ld, T, C, xs, status, set_mask, runs);
        case 15: return launch_tile_sort<15>(ctx, X_dev, ld, T, C, xs, status, set_mask, runs);
        case 17: return launch_tile_sort<17>(ctx, X_dev, ld, T, C, xs, status, set_mask, runs);
    }
    return sd_set_error(SD_ERR_INVALID, "qm tile sort: width %d not instantiated", K);
}

#define QM_DISPATCH_K(K, call_prefix, ...)                              \
    switch (K) {                                                        \
        case 1: SD_TRY(call_prefix<1>(__VA_ARGS__)); break;             \
        case 3: SD_TRY(call_prefix<3>(__VA_ARGS__)); break;             \
        case 5: SD_TRY(call_prefix<5>(__VA_ARGS__)); break;             \
        case 9: SD_TRY(call_prefix<9>(__VA_ARGS__)); break;             \
        case 13: SD_TRY(call_prefix<13>(__VA_ARGS__)); break;           \
        case 15: SD_TRY(call_prefix<15>(__VA_ARGS__)); break;           \
        case 17: SD_TRY(call_prefix<17>(__VA_ARGS__)); break;           \
        default: SD_TRY(call_prefix<19>(__VA_ARGS__)); break;           \
    }

}  // namespace

extern "C" {

int sd_qm_state_destroy(sd_qm_state* st) {
    if (!st) return SD_OK;
    if (st->ctx) {
        (void)hipSetDevice(st->ctx->device);
        (void)hipStreamSynchronize(st->ctx->stream);
    }
    sd_pool_release(st->ctx, st->xs);
    sd_pool_release(st->ctx, st->ys);
    sd_pool_release(st->ctx, st->status);
    delete st;
    return SD_OK;
}

int sd_qm_state_info(const sd_qm_state* st, int64_t* T, int64_t* C) {
    SD_CHECK_ARG(st, "state is NULL");
    if (T) *T = st->T;
    if (C) *C = st->C;
    return SD_OK;
}

int sd_qm_state_export(const sd_qm_state* st, double* x_sorted, double* y_sorted, int32_t* cell_status) {
    SD_CHECK_ARG(st, "state is NULL");
    sd_ctx* ctx = st->ctx;
    SD_HIP(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(double) * (size_t)st->T * st->C;
    if (x_sorted) SD_HIP(hipMemcpyAsync(x_sorted, st->xs, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SD_CHECK_ARG(!y_sorted || st->ys, "sd_qm_state_export: the state was fitted without y");
    if (y_sorted) SD_HIP(hipMemcpyAsync(y_sorted, st->ys, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (cell_status) {
        std::vector<int32_t> bits(st->C);
        SD_HIP(hipMemcpyAsync(bits.data(), st->status, sizeof(int32_t) * st->C, hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        for (int64_t c = 0; c < st->C; ++c) cell_status[c] = sd_public_status(bits[c]);
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_qm_fit_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int64_t C, sd_qm_state** out) {
    SD_CHECK_ARG(ctx && X_dev && out, "sd_qm_fit: NULL argument");  // y may be NULL (CunnaneTransformer: only the X CDF)
    SD_CHECK_ARG(T >= 2 && C > 0 && ld >= C, "sd_qm_fit: bad sizes");
    *out = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    const int K = sort_width(T, ctx->lds_max);
    if (K == 0) return sd_set_error(SD_ERR_UNSUPPORTED, "sd_qm_fit: series of %lld samples exceed the workgroup sort (19456)", (long long)T);
    sd_qm_state* st = new sd_qm_state();
    st->ctx = ctx;
    st->T = T;
    st->C = C;
    auto body = [&]() -> int {
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->xs, sizeof(double) * (size_t)T * C));
        if (y_dev) SD_HIP(sd_pool_malloc(ctx, (void**)&st->ys, sizeof(double) * (size_t)T * C));
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->status, sizeof(int32_t) * C));
        SD_HIP(hipMemsetAsync(st->status, 0, sizeof(int32_t) * C, ctx->stream));
        const int Kt = tile_runs_width(T, ctx->lds_max);
        if (Kt != 0) {  // tile-shaped first stage: sorted runs straight from the time-major fields, then the merge rounds
            sd_scratch runs;
            const int64_t np = (T + 64 * Kt - 1) / (64 * Kt) * (64 * Kt);
            SD_HIP(runs.alloc(ctx, sizeof(double) * (size_t)np * C));
            SD_TRY(launch_tile_sort_width(ctx, Kt, X_dev, ld, T, C, st->xs, st->status, 1, runs.as<double>()));
            if (y_dev) SD_TRY(launch_tile_sort_width(ctx, Kt, y_dev, ld, T, C, st->ys, st->status, 0, runs.as<double>()));
            SD_HIP(hipStreamSynchronize(ctx->stream));  // (the runs go back to the block cache at scope exit)
            return SD_OK;
        }
        dim3 grid((unsigned)((C + 31) / 32), (unsigned)((T + 31) / 32));
        SD_LAUNCH(ctx, "qm_transpose_kernel", qm_transpose_kernel, grid, dim3(256), 0, X_dev, ld, T, C, st->xs, st->status, 1);
        QM_DISPATCH_K(K, launch_sort, ctx, st->xs, T, C);
        if (y_dev) {
            SD_LAUNCH(ctx, "qm_transpose_kernel", qm_transpose_kernel, grid, dim3(256), 0, y_dev, ld, T, C, st->ys, st->status, 0);
            QM_DISPATCH_K(K, launch_sort, ctx, st->ys, T, C);
        }
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    };
    const int rc = body();
    if (rc != SD_OK) {
        sd_qm_state_destroy(st);
        return rc;
    }
    *out = st;
    return SD_OK;
}

int sd_qm_fit(sd_ctx* ctx, const double* X, const double* y, int64_t T, int64_t C, sd_qm_state** out) {
    SD_CHECK_ARG(ctx && X && out, "sd_qm_fit: NULL argument");
    SD_CHECK_ARG(T >= 2 && C > 0, "sd_qm_fit: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy;
    const size_t bytes = sizeof(double) * (size_t)T * C;
    SD_HIP(dX.alloc(ctx, bytes));
    SD_TRY(sd_copy_h2d(ctx, dX.p, X, bytes));
    if (y) {
        SD_HIP(dy.alloc(ctx, bytes));
        SD_TRY(sd_copy_h2d(ctx, dy.p, y, bytes));
    }
    return sd_qm_fit_dev(ctx, dX.as<double>(), y ? dy.as<double>() : nullptr, C, T, C, out);
}

int sd_qm_predict_dev(sd_ctx* ctx, const sd_qm_state* st, int model, int extrapolate, int n_endpoints, const double* Xp_dev, int64_t ld,
                      int64_t Tp, double* out_dev, int64_t ld_out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xp_dev && out_dev, "sd_qm_predict: NULL argument");
    SD_CHECK_ARG(extrapolate == SD_EXTRAP_1TO1 || (extrapolate >= SD_EXTRAP_NONE && extrapolate <= SD_EXTRAP_BOTH),
                 "sd_qm_predict: unknown extrapolate code %d", extrapolate);
    SD_CHECK_ARG(n_endpoints >= 2, "Invalid number of n_endpoints, must be >= 2");
    SD_CHECK_ARG(model >= SD_QM_REGRESSOR && model <= SD_QM_EDCDF_RATIO, "sd_qm_predict: unknown model %d", model);
    SD_CHECK_ARG(st->ys, "sd_qm_predict: the state was fitted without y");
    SD_CHECK_ARG(Tp > 0 && ld >= st->C && ld_out >= st->C, "sd_qm_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C, T = st->T;
    const int K = model == SD_QM_REGRESSOR ? 1 : sort_width(Tp, ctx->lds_max);
    if (K == 0) return sd_set_error(SD_ERR_UNSUPPORTED, "sd_qm_predict: series of %lld samples exceed the workgroup sort (19456)", (long long)Tp);
    const size_t lds_map = sizeof(double) * (size_t)T;  // one table of the fit at a time (qm_map_kernel)
    SD_CHECK_ARG(lds_map <= ctx->lds_max, "sd_qm_predict: fitted series too long for the LDS-resident search");
    sd_scratch qc, oc, rk, status_p, status_pub;
    SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tp * C));
    SD_HIP(oc.alloc(ctx, sizeof(double) * (size_t)Tp * C));
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((Tp + 31) / 32));
    SD_LAUNCH(ctx, "qm_transpose_kernel", qm_transpose_kernel, grid, dim3(256), 0, Xp_dev, ld, Tp, C, qc.as<double>(),
              status_p.as<int32_t>(), 0);
    if (model != SD_QM_REGRESSOR) {
        SD_HIP(rk.alloc(ctx, sizeof(int32_t) * (size_t)Tp * C));
        QM_DISPATCH_K(K, launch_rank, ctx, qc.as<double>(), Tp, C, rk.as<int32_t>());
    }
    // plotting positions without divisions (ppq): only if the correction step reproduces the division on both grids
    const double dn_h = ((double)T + 1.0 - kAlpha) - kBeta, dm_h = ((double)Tp + 1.0 - kAlpha) - kBeta;  // pp_denom
    const double rdn_h = 1.0 / dn_h, rdm_h = 1.0 / dm_h;
    sd_scratch ppflag, tails;
    SD_HIP(ppflag.alloc(ctx, sizeof(int32_t)));
    SD_HIP(hipMemsetAsync(ppflag.p, 0, sizeof(int32_t), ctx->stream));
    SD_LAUNCH(ctx, "qm_ppcheck_kernel", qm_ppcheck_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (int)T, dn_h, rdn_h, ppflag.as<int32_t>());
    SD_LAUNCH(ctx, "qm_ppcheck_kernel", qm_ppcheck_kernel, dim3((unsigned)((Tp + 255) / 256)), dim3(256), 0, (int)Tp, dm_h, rdm_h, ppflag.as<int32_t>());
    int32_t pp_mismatch = 1;
    SD_HIP(hipMemcpyAsync(&pp_mismatch, ppflag.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((extrapolate & (SD_EXTRAP_MIN | SD_EXTRAP_MAX)) != 0 && extrapolate != SD_EXTRAP_1TO1) {
        SD_HIP(tails.alloc(ctx, sizeof(double) * 4 * (size_t)C));
        SD_LAUNCH(ctx, "qm_tails_kernel", qm_tails_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, extrapolate, n_endpoints,
                  (const double*)st->xs, (const double*)st->ys, T, C, tails.as<double>());
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    const bool fastpp = pp_mismatch == 0 && sd_dev_env("SD_QM_DIVIDE") == nullptr;
    const int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * (lds_map > ctx->lds_max / 2 ? 1 : 2));
    SD_CHECK_ARG(Tp < ((int64_t)1 << 31), "sd_qm_predict: series too long");
#define SD_QM_MAP(QMR, PER, FAST)                                                                                                          \
    do {                                                                                                                                   \
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qm_map_kernel<QMR, PER, FAST>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds_map));                                                                                         \
        SD_LAUNCH(ctx, "qm_map_kernel", (qm_map_kernel<QMR, PER, FAST>), dim3(nb), dim3(1024), lds_map, model, extrapolate, n_endpoints,   \
                  (const double*)qc.p, (const int32_t*)rk.p, (const double*)st->xs, (const double*)st->ys, T, Tp, C, oc.as<double>(),      \
                  rdn_h, rdm_h, (const double*)tails.p);                                                                                   \
    } while (0)
    if (model == SD_QM_REGRESSOR) {
        if (fastpp) SD_QM_MAP(true, 16, true);
        else SD_QM_MAP(true, 16, false);
    } else {
        if (fastpp) SD_QM_MAP(false, 8, true);
        else SD_QM_MAP(false, 8, false);
    }
#undef SD_QM_MAP
#ifdef SD_DEV
    if (sd_dev_env("SD_QM_TRACE") != nullptr) {
        long long h[64];
        SD_HIP(hipStreamSynchronize(ctx->stream));
        SD_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(sd_qm_trace), sizeof(h)));
        for (int r = 0; r < 8; ++r)
            fprintf(stderr, "qm_map trace cell %d: consts+loads %lld fill xs %lld gen1 %lld barrier %lld fill ys %lld gen2 %lld\n", r, h[r * 8 + 1] - h[r * 8],
                    h[r * 8 + 2] - h[r * 8 + 1], h[r * 8 + 3] - h[r * 8 + 2], h[r * 8 + 4] - h[r * 8 + 3], h[r * 8 + 5] - h[r * 8 + 4], h[r * 8 + 6] - h[r * 8 + 5]);
    }
#endif
    SD_LAUNCH(ctx, "qm_untranspose_kernel", qm_untranspose_kernel, grid, dim3(256), 0, (const double*)oc.p, Tp, C, out_dev, ld_out,
              (const int32_t*)st->status, (const int32_t*)status_p.p);
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "qm_status_public_kernel", qm_status_public_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                  (const int32_t*)st->status, (const int32_t*)status_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_qm_predict(sd_ctx* ctx, const sd_qm_state* st, int model, int extrapolate, int n_endpoints, const double* Xp, int64_t Tp, double* out,
                  int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xp && out, "sd_qm_predict: NULL argument");
    SD_CHECK_ARG(Tp > 0, "sd_qm_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dout;
    const size_t bytes = sizeof(double) * (size_t)Tp * st->C;
    SD_HIP(dX.alloc(ctx, bytes));
    SD_HIP(dout.alloc(ctx, bytes));
    SD_TRY(sd_copy_h2d(ctx, dX.p, Xp, bytes));
    SD_TRY(sd_qm_predict_dev(ctx, st, model, extrapolate, n_endpoints, dX.as<double>(), st->C, Tp, dout.as<double>(), st->C, cell_status));
    SD_TRY(sd_copy_d2h(ctx, out, dout.p, bytes));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_qm_cunnane_dev(sd_ctx* ctx, const sd_qm_state* st, int direction, int extrapolate, int n_endpoints, const double* X_dev,
                      int64_t ld, int64_t Tp, double* out_dev, int64_t ld_out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && X_dev && out_dev, "sd_qm_cunnane: NULL argument");
    SD_CHECK_ARG(direction == SD_CUNNANE_FORWARD || direction == SD_CUNNANE_INVERSE, "sd_qm_cunnane: unknown direction %d", direction);
    SD_CHECK_ARG(extrapolate >= SD_EXTRAP_NONE && extrapolate <= SD_EXTRAP_BOTH, "sd_qm_cunnane: unknown extrapolate code %d", extrapolate);
    SD_CHECK_ARG(n_endpoints >= 1, "sd_qm_cunnane: n_endpoints must be positive");
    SD_CHECK_ARG(Tp > 0 && ld >= st->C && ld_out >= st->C, "sd_qm_cunnane: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C, T = st->T;
    const size_t lds = direction == SD_CUNNANE_FORWARD ? sizeof(double) * (size_t)T : 8;
    SD_CHECK_ARG(lds <= ctx->lds_max, "sd_qm_cunnane: fitted series too long for the LDS-resident search");
    sd_scratch qc, oc, status_p, status_pub;
    SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tp * C));
    SD_HIP(oc.alloc(ctx, sizeof(double) * (size_t)Tp * C));
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((Tp + 31) / 32));
    SD_LAUNCH(ctx, "qm_transpose_kernel", qm_transpose_kernel, grid, dim3(256), 0, X_dev, ld, Tp, C, qc.as<double>(),
              status_p.as<int32_t>(), 0);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qm_cunnane_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * (lds > ctx->lds_max / 2 ? 1 : 2));
    SD_LAUNCH(ctx, "qm_cunnane_kernel", qm_cunnane_kernel, dim3(nb), dim3(1024), lds, direction, extrapolate & SD_EXTRAP_MIN,
              extrapolate & SD_EXTRAP_MAX, n_endpoints, (const double*)qc.p, (const double*)st->xs, T, Tp, C, oc.as<double>());
    SD_LAUNCH(ctx, "qm_untranspose_kernel", qm_untranspose_kernel, grid, dim3(256), 0, (const double*)oc.p, Tp, C, out_dev, ld_out,
              (const int32_t*)st->status, (const int32_t*)status_p.p);
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "qm_status_public_kernel", qm_status_public_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                  (const int32_t*)st->status, (const int32_t*)status_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_qm_cunnane(sd_ctx* ctx, const sd_qm_state* st, int direction, int extrapolate, int n_endpoints, const double* X, int64_t Tp,
                  double* out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && X && out, "sd_qm_cunnane: NULL argument");
    SD_CHECK_ARG(Tp > 0, "sd_qm_cunnane: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dout;
    const size_t bytes = sizeof(double) * (size_t)Tp * st->C;
    SD_HIP(dX.alloc(ctx, bytes));
    SD_HIP(dout.alloc(ctx, bytes));
    SD_TRY(sd_copy_h2d(ctx, dX.p, X, bytes));
    SD_TRY(sd_qm_cunnane_dev(ctx, st, direction, extrapolate, n_endpoints, dX.as<double>(), st->C, Tp, dout.as<double>(), st->C,
                             cell_status));
    SD_TRY(sd_copy_d2h(ctx, out, dout.p, bytes));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

}  // extern "C"
