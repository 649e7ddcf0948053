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

// model: 0 QuantileMappingReressor, 1 EquidistantCdfMatcher 'difference', 2 'ratio'.  One workgroup per cell;
// QMR keeps the cell's sorted fit X in LDS for the value -> position search.
__global__ void __launch_bounds__(1024) qm_map_kernel(int model, int mode, int n_end, const double* __restrict__ qc /* [C][Tp] */,
                                                      const int32_t* __restrict__ rank /* [C][Tp] or null */,
                                                      const double* __restrict__ xs_all, const double* __restrict__ ys_all,
                                                      int64_t T, int64_t Tp, int64_t C, double* __restrict__ oc /* [C][Tp] */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xl = reinterpret_cast<double*>(smem_raw);  // QMR: n sorted fit values
    const int n = (int)T, tid = threadIdx.x, nthr = blockDim.x;
    const double dn = pp_denom(n), dm = pp_denom((int)Tp);
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* xs = xs_all + c * T;
        const double* ys = ys_all + c * T;
        __syncthreads();
        if (model == 0)
            for (int i = tid; i < n; i += nthr) xl[i] = xs[i];
        __syncthreads();
        const double x_min = xs[0], x_max = xs[n - 1], y_min = ys[0], y_max = ys[n - 1];
        // synthetic end points of the extended CDFs (quantile.py:366-385); every thread computes the same four numbers
        const bool ext_lo = (mode & SD_EXTRAP_MIN) != 0, ext_hi = (mode & SD_EXTRAP_MAX) != 0, one_to_one = (mode & SD_EXTRAP_1TO1) != 0;
        const int e = n_end < n ? n_end : n;
        const double vx_lo = ext_lo ? ols_value_at(xs, 0, e, dn, kSyntheticMin) : 0.0, vy_lo = ext_lo ? ols_value_at(ys, 0, e, dn, kSyntheticMin) : 0.0;
        const double vx_hi = ext_hi ? ols_value_at(xs, n - e, e, dn, kSyntheticMax) : 0.0, vy_hi = ext_hi ? ols_value_at(ys, n - e, e, dn, kSyntheticMax) : 0.0;
        for (int64_t tq = tid; tq < Tp; tq += nthr) {
            const double x = qc[c * Tp + tq];
            double res;
            if (model == 0) {
                // p = np.interp(x, xs, pp): j = last index with xs[j] <= x
                double p;
                if (x < x_min) {
                    if (ext_lo) {  // bracket (synthetic node, first value); below the synthetic node: left = -inf (quantile.py:245)
                        // (tied end points give a level tail line: vx_lo == x_min, every x < x_min is "below the node")
                        if (x < vx_lo || !(x_min > vx_lo)) {
                            p = -__builtin_inf();
                        } else {
                            const double slope = (pp_at(0, dn) - kSyntheticMin) / (x_min - vx_lo);
                            p = slope * (x - vx_lo) + kSyntheticMin;
                        }
                    } else {
                        p = pp_at(0, dn);
                    }
                } else if (x >= x_max) {
                    if (ext_hi && x > x_max) {
                        if (x > vx_hi || !(vx_hi > x_max)) {
                            p = __builtin_inf();
                        } else {
                            const double slope = (kSyntheticMax - pp_at(n - 1, dn)) / (vx_hi - x_max);
                            p = slope * (x - x_max) + pp_at(n - 1, dn);
                        }
                    } else {
                        p = pp_at(n - 1, dn);
                    }
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
                        p = pp_at(j, dn);
                    } else {
                        const double slope = (pp_at(j + 1, dn) - pp_at(j, dn)) / (xl[j + 1] - x0);
                        p = slope * (x - x0) + pp_at(j, dn);
                    }
                }
                res = interp_on_grid(p, n, dn, ys, ext_lo, vy_lo, ext_hi, vy_hi);  // quantile.py:268-269
            } else {
                const double p = pp_at(rank[c * Tp + tq], dm);  // plotting position of x within the new series
                const double x_train = interp_on_grid(p, n, dn, xs, ext_lo, vx_lo, ext_hi, vx_hi);  // quantile.py:613
                const double y_map = interp_on_grid(p, n, dn, ys, ext_lo, vy_lo, ext_hi, vy_hi);
                res = model == 1 ? y_map + (x - x_train) : y_map * (x / x_train);  // quantile.py:616-623
            }
            if (one_to_one) {  // quantile.py:277-310 (fit X and y have the same length)
                if (x > x_max) res = y_max + (x - x_max);
                if (x < x_min) res = y_min + (x - x_min);
            }
            oc[c * Tp + tq] = res;
        }
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
    const size_t lds_map = model == SD_QM_REGRESSOR ? sizeof(double) * (size_t)T : 0;
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
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qm_map_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(lds_map ? lds_map : 8)));
    const int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * (lds_map > ctx->lds_max / 2 ? 1 : 2));
    SD_LAUNCH(ctx, "qm_map_kernel", qm_map_kernel, dim3(nb), dim3(1024), lds_map ? lds_map : 8, model, extrapolate, n_endpoints,
              (const double*)qc.p, (const int32_t*)rk.p, (const double*)st->xs, (const double*)st->ys, T, Tp, C, oc.as<double>());
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
