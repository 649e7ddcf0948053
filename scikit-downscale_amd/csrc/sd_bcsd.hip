// BCSD empirical quantile mapping, batched over the cell axis.
//
// One workgroup = W adjacent cells x one time group (calendar month); one 64-lane wave per cell.
// A *segment* is the chronologically ordered samples of one (cell, group): n ~ 1240 doubles for a
// 40-year daily series.  Reference semantics (file:line under skdownscale/pointwise_models):
//   fit      bcsd.py:197-228 / 115-147 -> quantile.py:81-107 -> 438-463 (np.sort) + 23-43 (Cunnane pp)
//   predict  bcsd.py:230-269 / 149-185 -> quantile.py:109-147 -> 505-521,488 (self ECDF, np.interp
//            exact-hit rule = last xp <= x) -> 523-545 (inverse through fitted CDF + 10-point OLS tails)
//
// Data layout: fields are [T, ld] with cells contiguous, so a workgroup reads W*8-byte row
// fragments (coalesced over the cell axis) and transposes them through LDS into cell-major
// segments; sorted state is stored cell-major [C][T] so each wave streams contiguous segments.
#include <cmath>
#include <cstdlib>

#include <cstring>

#include <chrono>
#include <thread>

#include "sd_bcsd_rs.h"
#include "sd_internal.h"
#include "sd_sortnet.h"

namespace {

constexpr int kWave = 64;
constexpr double kAlpha = 0.4;  // quantile.py:423
constexpr double kBeta = 0.4;   // quantile.py:424

__device__ __forceinline__ bool sd_finite(double v) { return (__double_as_longlong(v) & 0x7ff0000000000000ll) != 0x7ff0000000000000ll; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Cunnane plotting position of 0-based rank i among n (quantile.py:43, same operation order).
__device__ __forceinline__ double pp_denom(int n) { return ((double)n + 1.0 - kAlpha) - kBeta; }
__device__ __forceinline__ double pp_at(int i, double denom) { return ((double)(i + 1) - kAlpha) / denom; }

// ---- wave-cooperative bitonic sort of a[0..n) in LDS, ascending --------------------------------
// Standard-form network (every comparator puts the min at the lower index) for the next power of
// two, with comparators that touch an index >= n dropped: the dropped slots behave as +inf
// padding that never moves, so the truncated network still sorts any n.
// All waves of the block call this with the same n (block-level barriers).
__device__ void block_bitonic_sort_rows(double* row, int n, int lane) {
    int N = 1;
    while (N < n) N <<= 1;
    const int half = N >> 1;
    for (int size = 2; size <= N; size <<= 1) {
        const int hs = size >> 1;
        for (int i = lane; i < half; i += kWave) {
            const int blk = i / hs, off = i - blk * hs;
            const int lo = blk * size + off, hi = blk * size + size - 1 - off;
            if (hi < n) {
                const double a = row[lo], b = row[hi];
                if (b < a) { row[lo] = b; row[hi] = a; }
            }
        }
        __syncthreads();
        for (int stride = size >> 2; stride >= 1; stride >>= 1) {
            for (int i = lane; i < half; i += kWave) {
                const int blk = i / stride, off = i - blk * stride;
                const int lo = blk * 2 * stride + off, hi = lo + stride;
                if (hi < n) {
                    const double a = row[lo], b = row[hi];
                    if (b < a) { row[lo] = b; row[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Load the rows of one group for W adjacent cells into a cell-major LDS tile (transpose), flagging
// non-finite samples.  tile[cell * stride + r]; out-of-range cells are filled with 0.
template <int W>
__device__ void load_group_tile(const double* __restrict__ src, int64_t ld, const int32_t* __restrict__ order, int n,
                                int64_t c0, int64_t C, double* tile, int stride, int32_t* status_bits) {
    const int total = n * W;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int r = i / W, cl = i - r * W;
        const int64_t c = c0 + cl;
        double v = 0.0;
        if (c < C) {
            v = src[(int64_t)order[r] * ld + c];
            if (!sd_finite(v)) atomicOr(&status_bits[c], SDI_NONFINITE);
        }
        tile[cl * stride + r] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// mask: core.py:35-37  active iff first sample of the first feature is not NaN
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bcsd_mask_kernel(const double* __restrict__ X0, int64_t C, int32_t* status) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const double v = X0[c];
        status[c] = (v != v) ? SDI_MASKED : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// fit: per (cell tile, group): x_climo, y_climo, sorted y segment
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(64 * W) bcsd_fit_kernel(int kind, const double* __restrict__ X,
                                                          const double* __restrict__ y, int64_t ld,
                                                          const int32_t* __restrict__ order,
                                                          const int32_t* __restrict__ goff, int G, int64_t T, int64_t C,
                                                          int stride, int return_anoms, double* __restrict__ ys,
                                                          double* __restrict__ x_climo, double* __restrict__ y_climo,
                                                          int32_t* status) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);
    const int g = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * W;
    const int beg = goff[g], n = goff[g + 1] - beg;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int64_t c = c0 + wave;
    double* row = tile + wave * stride;
    if (n == 0) return;

    // x climatology (bcsd.py:222) -- X only needs its group mean; PR validates X only.
    if (X != nullptr) {
        load_group_tile<W>(X, ld, order + beg, n, c0, C, tile, stride, status);
        __syncthreads();
        if (kind == SD_BCSD_TAS) {
            double s = 0.0;
            for (int i = lane; i < n; i += kWave) s += row[i];
            s = wave_sum(s);
            if (lane == 0 && c < C) x_climo[c * G + g] = s / (double)n;
        }
        __syncthreads();
    }
    load_group_tile<W>(y, ld, order + beg, n, c0, C, tile, stride, status);
    __syncthreads();
    {
        double s = 0.0;
        for (int i = lane; i < n; i += kWave) s += row[i];
        s = wave_sum(s);
        const double m = s / (double)n;
        if (lane == 0 && c < C) {
            y_climo[c * G + g] = m;  // bcsd.py:223 / 138
            if (kind == SD_BCSD_PR && return_anoms && m <= 0.0) atomicOr(&status[c], SDI_BAD_CLIMO);  // bcsd.py:140-141
        }
    }
    block_bitonic_sort_rows(row, n, lane);  // quantile.py:462 np.sort
    if (c < C) {
        double* dst = ys + c * T + beg;
        for (int i = lane; i < n; i += kWave) dst[i] = row[i];
    }
}

// ------------------------------------------------------------------------------------------------
// predict
// ------------------------------------------------------------------------------------------------
struct TailFit {
    double slope_lo, icpt_lo, slope_hi, icpt_hi;
    int tails;  // SD_QT_TAIL_LOWER | SD_QT_TAIL_UPPER: which sides continue along their line (the other takes np.interp's end value)
};

// 10-endpoint OLS lines for the CDF tails (quantile.py:532-543; sklearn LinearRegression = centred LS)
__device__ void ols_line(const double* __restrict__ ysg, int first, int e, double denom, double* slope, double* icpt) {
    double xm = 0.0, ym = 0.0;
    for (int i = 0; i < e; ++i) {
        xm += pp_at(first + i, denom);
        ym += ysg[first + i];
    }
    xm /= (double)e;
    ym /= (double)e;
    double sxx = 0.0, sxy = 0.0;
    for (int i = 0; i < e; ++i) {
        const double dx = pp_at(first + i, denom) - xm;
        sxx += dx * dx;
        sxy += dx * (ysg[first + i] - ym);
    }
    const double s = sxx > 0.0 ? sxy / sxx : 0.0;
    *slope = s;
    *icpt = ym - s * xm;
}

// value of the fitted inverse CDF at probability p (np.interp semantics + OLS tails)
__device__ __forceinline__ double inverse_cdf(double p, const double* __restrict__ ysg, int n, double denom,
                                              const TailFit& tf) {
    const double pp0 = pp_at(0, denom), ppl = pp_at(n - 1, denom);
    if (p < pp0) return (tf.tails & SD_QT_TAIL_LOWER) ? p * tf.slope_lo + tf.icpt_lo : ysg[0];      // quantile.py:527-545
    if (p > ppl) return (tf.tails & SD_QT_TAIL_UPPER) ? p * tf.slope_hi + tf.icpt_hi : ysg[n - 1];
    // pp is an affine grid: analytic guess, then guard against rounding of the guess
    int i = (int)floor(p * denom + kAlpha) - 1;
    i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    while (i + 1 < n && pp_at(i + 1, denom) <= p) ++i;
    while (i > 0 && pp_at(i, denom) > p) --i;
    const double pi = pp_at(i, denom);
    const double yi = ysg[i];
    if (i == n - 1 || pi == p) return yi;
    const double slope = (ysg[i + 1] - yi) / (pp_at(i + 1, denom) - pi);
    return slope * (p - pi) + yi;
}

// rolling(9, center=True, min_periods=1).mean() at position j of a segment (bcsd.py:247-250)
__device__ __forceinline__ double rolling9(const double* __restrict__ x, int m, int j) {
    const int lo = j - 4 < 0 ? 0 : j - 4;
    const int hi = j + 5 > m ? m : j + 5;
    double s = 0.0;
    for (int i = lo; i < hi; ++i) s += x[i];
    return s / (double)(hi - lo);
}

template <int W>
__global__ void __launch_bounds__(64 * W) bcsd_predict_kernel(
    int kind, const double* __restrict__ Xp, int64_t ld, const int32_t* __restrict__ order_p,
    const int32_t* __restrict__ goff_p, const int32_t* __restrict__ goff_f, int G, int64_t Tf, int64_t C, int stride,
    int return_anoms, const double* __restrict__ ys, const double* __restrict__ x_climo,
    const double* __restrict__ y_climo, const int32_t* __restrict__ fit_status, int32_t* status,
    double* __restrict__ out, int64_t ld_out, int qt_tails, int qt_endpoints) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tile_x = reinterpret_cast<double*>(smem_raw);
    double* tile_s = tile_x + W * stride;
    double* stage = tile_s + W * stride;  // [W][64]
    const int g = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * W;
    const int begp = goff_p[g], m = goff_p[g + 1] - begp;
    const int begf = goff_f[g], n = goff_f[g + 1] - begf;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int64_t c = c0 + wave;
    const bool cell_ok = c < C;
    if (m == 0) return;
    const int32_t* ord = order_p + begp;

    load_group_tile<W>(Xp, ld, ord, m, c0, C, tile_x, stride, status);
    __syncthreads();

    const double* xr = tile_x + wave * stride;
    double* sr = tile_s + wave * stride;
    const double xc = (kind == SD_BCSD_TAS && cell_ok) ? x_climo[c * G + g] : 0.0;
    const double yc = cell_ok ? y_climo[c * G + g] : 1.0;
    // u = X - (rolling mean - x_climo)   (bcsd.py:247-256); PR maps raw X (bcsd.py:167)
    for (int j = lane; j < m; j += kWave) {
        double u = xr[j];
        if (kind == SD_BCSD_TAS) u = u - (rolling9(xr, m, j) - xc);
        sr[j] = u;
    }
    __syncthreads();
    block_bitonic_sort_rows(sr, m, lane);  // self ECDF: np.sort(u)  (quantile.py:462 via 505-521)

    const double* ysg = ys + (cell_ok ? c : 0) * Tf + begf;
    const double dn = pp_denom(n), dm = pp_denom(m);
    TailFit tf = {0.0, 0.0, 0.0, 0.0, qt_tails};
    const bool active = cell_ok && n > 0 && fit_status[c] == 0;
    if (active && m > n) {  // p can leave [pp_0, pp_{n-1}] only when the predict segment is longer
        const int e = n < qt_endpoints ? n : qt_endpoints;
        ols_line(ysg, 0, e, dn, &tf.slope_lo, &tf.icpt_lo);
        ols_line(ysg, n - e, e, dn, &tf.slope_hi, &tf.icpt_hi);
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ll);

    for (int k0 = 0; k0 < m; k0 += kWave) {
        const int j = k0 + lane;
        double res = nan;
        if (j < m && active) {
            const double x = xr[j];
            double shift = 0.0, u = x;
            if (kind == SD_BCSD_TAS) {
                shift = rolling9(xr, m, j) - xc;
                u = x - shift;
            }
            // rank = (number of sorted values <= u) - 1  == np.interp exact-hit index (max rank among ties)
            int lo = 0, hi = m;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sr[mid] <= u) lo = mid + 1; else hi = mid;
            }
            const int r = lo > 0 ? lo - 1 : 0;
            const double p = pp_at(r, dm);
            const double q = inverse_cdf(p, ysg, n, dn, tf);
            if (kind == SD_BCSD_TAS) {
                res = shift + q;                      // bcsd.py:263
                if (return_anoms) res = res - yc;     // bcsd.py:266-267
            } else {
                res = return_anoms ? q / yc : q;      // bcsd.py:170-185
            }
        }
        stage[wave * kWave + lane] = res;
        __syncthreads();
        {   // transposed, coalesced store of 64 rows x W cells
            const int rr = threadIdx.x / W, cl = threadIdx.x - rr * W;
            const int j2 = k0 + rr;
            if (j2 < m && c0 + cl < C) out[(int64_t)ord[j2] * ld_out + c0 + cl] = stage[cl * kWave + rr];
        }
        __syncthreads();
    }
}

// fold internal status bits into public codes, in place
__global__ void __launch_bounds__(256) status_public_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                            int64_t C, int32_t* __restrict__ outp) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        // a = what fit found, b = what predict found: the reference raises in fit first (base.py:18-20, then bcsd.py:140-141),
        // so a cell with a bad climatology AND a non-finite predict sample reports the climatology
        int32_t code = sd_public_status(a[c]);
        if (code == SD_CELL_OK && b) code = sd_public_status(b[c]);
        outp[c] = code;
    }
}

// fill the columns of cells whose status != 0 with NaN (masked / failed cells; core.py:119).
// One workgroup per 32 cells; an all-OK strip costs one status read.
__global__ void __launch_bounds__(256) nan_fill_kernel(double* __restrict__ out, int64_t ld, int64_t Tp, int64_t C,
                                                       const int32_t* __restrict__ st_a, const int32_t* __restrict__ st_b) {
    const int64_t c = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const bool bad = c < C && (st_a[c] | (st_b ? st_b[c] : 0)) != 0;
    if (!__syncthreads_or(bad)) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (bad)
        for (int64_t t = threadIdx.x >> 5; t < Tp; t += 8) out[t * ld + c] = nan;
}

// ---- predict with a climate-trend grouper that differs from the time grouper (bcsd.py:247-267) ----
// One thread per cell (256 adjacent cells per workgroup: 2 KB row fragments), one trend group per blockIdx.y: the
// 9-sample centred rolling mean (min_periods=1) walks the group's time steps in order; window sums are taken in the
// same fixed order as in the tile kernels.
__global__ void __launch_bounds__(256) bcsd_trend_shift_kernel(const double* __restrict__ X, int64_t ld, const int32_t* __restrict__ ord,
                                                               const int32_t* __restrict__ off, const int32_t* __restrict__ gid_qm,
                                                               int G, const double* __restrict__ x_climo, int64_t C,
                                                               double* __restrict__ shift, double* __restrict__ u) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int beg = off[blockIdx.y], m = off[blockIdx.y + 1] - beg;
    const int32_t* o = ord + beg;
    double w[9];  // w[d] = sample j - 4 + d (0 outside the group)
#pragma unroll
    for (int d = 0; d < 9; ++d) w[d] = (d >= 4 && d - 4 < m) ? X[(int64_t)o[d - 4] * ld + c] : 0.0;
    for (int j = 0; j < m; ++j) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < 9; ++d) s += w[d];
        const int lo = j - 4 > 0 ? j - 4 : 0, hi = j + 5 < m ? j + 5 : m;
        const double mean = s / (double)(hi - lo);
        const int64_t t = o[j];
        const double sh = mean - x_climo[c * G + gid_qm[t]];  // bcsd.py:253
        shift[t * C + c] = sh;
        u[t * C + c] = w[4] - sh;  // bcsd.py:256
#pragma unroll
        for (int d = 0; d < 8; ++d) w[d] = w[d + 1];
        w[8] = j + 5 < m ? X[(int64_t)o[j + 5] * ld + c] : 0.0;
    }
}

__global__ void __launch_bounds__(256) bcsd_trend_restore_kernel(double* __restrict__ out, int64_t ld_out, const double* __restrict__ shift,
                                                                 const int32_t* __restrict__ gid_qm, int G, const double* __restrict__ y_climo,
                                                                 int return_anoms, int64_t Tp, int64_t C) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t t1 = min((int64_t)(blockIdx.y + 1) * 64, Tp);
    for (int64_t t = (int64_t)blockIdx.y * 64; t < t1; ++t) {
        double v = shift[t * C + c] + out[t * ld_out + c];            // bcsd.py:263
        if (return_anoms) v = v - y_climo[c * G + gid_qm[t]];           // bcsd.py:266-267
        out[t * ld_out + c] = v;
    }
}

int pick_tile_width(size_t lds_max, int nmax, int tiles, int* W, int* stride) {
    // LDS need: tiles * W * stride * 8 (+ W*64*8 staging when tiles == 2)
    const int st = nmax | 1;  // odd stride: rows of different cells start on different banks
    for (int w = 8; w >= 1; w >>= 1) {
        size_t need = (size_t)tiles * w * st * sizeof(double) + (tiles == 2 ? (size_t)w * 64 * sizeof(double) : 0);
        if (need <= lds_max) {
            *W = w;
            *stride = st;
            return SD_OK;
        }
    }
    return sd_set_error(SD_ERR_UNSUPPORTED, "BCSD segment of %d samples does not fit the %zu-byte LDS", nmax, lds_max);
}

struct DevPtr {
    void* p = nullptr;
};
struct DevGroupTable {  // borrowed from the context's group-table cache (valid until the next upload_group_table calls evict it)
    DevPtr order, off;
    int nmax = 0;
    std::vector<int64_t> host_off;
    sd_scratch own_order, own_off;  // explicit tables (upload_explicit_table) are owned by the call
};

// Explicit group table: `order` lists time indices group by group (a time step may belong to several groups: the +-15
// day windows of time_grouper='daily_nasa-nex', groupers.py:19-89), off[G+1] delimits the groups.
int upload_explicit_table(sd_ctx* ctx, const int32_t* order, const int64_t* off, int G, int64_t T, DevGroupTable* d) {
    SD_CHECK_ARG(off[0] == 0, "group_offsets[0] must be 0");
    const int64_t N = off[G];
    SD_CHECK_ARG(N > 0 && N < (int64_t)1 << 31, "group table: %lld entries", (long long)N);
    d->host_off.assign(off, off + G + 1);
    d->nmax = 0;
    for (int g = 0; g < G; ++g) {
        SD_CHECK_ARG(off[g + 1] >= off[g], "group_offsets must not decrease");
        d->nmax = std::max(d->nmax, (int)(off[g + 1] - off[g]));
    }
    for (int64_t i = 0; i < N; ++i) SD_CHECK_ARG(order[i] >= 0 && order[i] < T, "group_order[%lld] = %d outside [0,%lld)", (long long)i, order[i], (long long)T);
    std::vector<int32_t> off32(off, off + G + 1);
    SD_HIP(d->own_order.alloc(ctx, sizeof(int32_t) * (size_t)N));
    SD_HIP(d->own_off.alloc(ctx, sizeof(int32_t) * (size_t)(G + 1)));
    SD_HIP(hipMemcpyAsync(d->own_order.p, order, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(hipMemcpyAsync(d->own_off.p, off32.data(), sizeof(int32_t) * (size_t)(G + 1), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    d->order.p = d->own_order.p;
    d->off.p = d->own_off.p;
    return SD_OK;
}

constexpr size_t kGtCacheEntries = 8;

int upload_group_table(sd_ctx* ctx, const int32_t* gid, int64_t T, int G, DevGroupTable* d) {
    SD_CHECK_ARG(T < (int64_t)1 << 31, "T too large");
    sd_gt_cache_entry* hit = nullptr;
    for (auto& e : ctx->gt_cache)
        if (e.G == G && (int64_t)e.gid.size() == T && memcmp(e.gid.data(), gid, sizeof(int32_t) * (size_t)T) == 0) hit = &e;
    if (hit == nullptr) {
        sd_group_table gt;
        SD_TRY(sd_build_group_table(gid, T, G, &gt));
        std::vector<int32_t> off32(gt.off.begin(), gt.off.end());
        if (ctx->gt_cache.size() >= kGtCacheEntries) {  // replace the least recently used entry (no call is in flight: calls synchronise)
            size_t lru = 0;
            for (size_t i = 1; i < ctx->gt_cache.size(); ++i)
                if (ctx->gt_cache[i].last_use < ctx->gt_cache[lru].last_use) lru = i;
            SD_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->gt_cache[lru].order);
            (void)hipFree(ctx->gt_cache[lru].off);
            ctx->gt_cache.erase(ctx->gt_cache.begin() + (long)lru);
        }
        sd_gt_cache_entry e;
        e.G = G;
        e.gid.assign(gid, gid + T);
        e.nmax = gt.nmax;
        e.host_off = gt.off;
        SD_HIP(hipMalloc((void**)&e.order, sizeof(int32_t) * (size_t)T));
        hipError_t rc = hipMalloc((void**)&e.off, sizeof(int32_t) * (size_t)(G + 1));
        if (rc != hipSuccess) {
            (void)hipFree(e.order);
            SD_HIP(rc);
        }
        ctx->gt_cache.push_back(std::move(e));
        hit = &ctx->gt_cache.back();
        SD_HIP(hipMemcpyAsync(hit->order, gt.order.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipMemcpyAsync(hit->off, off32.data(), sizeof(int32_t) * (G + 1), hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
    }
    hit->last_use = ++ctx->gt_clock;
    d->order.p = hit->order;
    d->off.p = hit->off;
    d->nmax = hit->nmax;
    d->host_off = hit->host_off;
    return SD_OK;
}

// BcsdTemperature and BcsdPrecipitation take the fused kernels of sd_bcsd_fx.hip (x side, y side, inverse CDF and shift /
// ratio of a segment in one workgroup pass, sorts on 32-bit keys in registers); the segments they hand back (work list:
// exactly tied samples, runs of equal keys too long for the fix-up) take RANK + APPLY, and so does every segment of
// more than 1 536 samples and QuantileMapper(detrend=True).
bool use_fused_path(int nmax, bool detrend) {
    const char* e = sd_dev_env("SD_BCSD_FUSED");  // "0": RANK + APPLY for every segment (A/B measurements)
    if (e && e[0] == '0') return false;
    if (detrend) return false;  // QuantileMapper(detrend=True): RANK / APPLY carry the trend lines
    return sd_bcsd_fx_supported(nmax);
}

// Hand-off / work-list workspace of one predict call, carved from the context workspace.
struct RsWorkspace {
    uint32_t* ranks = nullptr;
    double* shift = nullptr;
    double* x_climo = nullptr;  // [C][G], only for calls without a state
    double* trend_u = nullptr;  // [C*G][2], only for detrended quantile mapping
    int64_t* worklist = nullptr;
    int* work_count = nullptr;
    int work_cap = 0;
    int64_t* worklist2 = nullptr;  // second list of the compacting precipitation kernel (same capacity)
    int* work_count2 = nullptr;
};
int carve_workspace(sd_ctx* ctx, int nmax, int64_t C, int G, bool fused, bool want_x_climo, bool want_trend, RsWorkspace* w) {
    size_t rank_bytes = 0, shift_bytes = 0;
    sd_bcsd_rs_handoff_bytes(nmax, C, G, &rank_bytes, &shift_bytes);
    shift_bytes = 0;  // (the round-3 fused kernel could park its shift here; the current one keeps it in registers)
    const size_t cg_bytes = ((sizeof(double) * (size_t)G * (size_t)C + 255) / 256) * 256;
    const size_t xc_bytes = (want_x_climo ? cg_bytes : 0) + (want_trend ? 2 * cg_bytes : 0);
    const int64_t items = ((C + 7) / 8) * (int64_t)G;
    SD_CHECK_ARG(items < ((int64_t)1 << 31), "too many (tile, group) items");
    const size_t list_bytes = fused ? ((sizeof(int64_t) * (size_t)items + 255) / 256) * 256 : 0;
    void* ws = nullptr;
    SD_TRY(sd_workspace(ctx, rank_bytes + shift_bytes + xc_bytes + 2 * list_bytes + 256, &ws));
    char* base = static_cast<char*>(ws);
    w->ranks = reinterpret_cast<uint32_t*>(base);
    w->shift = shift_bytes ? reinterpret_cast<double*>(base + rank_bytes) : nullptr;
    w->x_climo = want_x_climo ? reinterpret_cast<double*>(base + rank_bytes + shift_bytes) : nullptr;
    w->trend_u = want_trend ? reinterpret_cast<double*>(base + rank_bytes + shift_bytes + (want_x_climo ? cg_bytes : 0)) : nullptr;
    if (fused) {
        w->worklist = reinterpret_cast<int64_t*>(base + rank_bytes + shift_bytes + xc_bytes);
        w->work_count = reinterpret_cast<int*>(base + rank_bytes + shift_bytes + xc_bytes + list_bytes);
        w->work_cap = (int)items;
        w->worklist2 = reinterpret_cast<int64_t*>(base + rank_bytes + shift_bytes + xc_bytes + list_bytes + 256);
        w->work_count2 = w->work_count + 1;  // (the 256 bytes behind the first list hold both counters)
        SD_HIP(hipMemsetAsync(w->work_count, 0, 2 * sizeof(int), ctx->stream));
    }
    return SD_OK;
}

// the kernels of one predict call: fused kernel + RANK / APPLY over its work list, or RANK + APPLY over everything
int run_predict_kernels(sd_ctx* ctx, sdrs::Params& p, bool fused, int nmax_all, const std::vector<int>& glen) {
    if (fused) {
        if (const char* e = sd_dev_env("SD_FZ_ABLATE")) p.dev_flags = atoi(e);
        SD_TRY(sd_bcsd_fx_launch(ctx, p, nmax_all, glen.data()));
        p.use_worklist = 1;
        p.shift = nullptr;
    }
    SD_TRY(sd_bcsd_rs_launch(ctx, sdrs::MODE_RANK, p, nmax_all, glen.data()));
    SD_TRY(sd_bcsd_rs_launch(ctx, sdrs::MODE_APPLY, p, nmax_all, glen.data()));
    return SD_OK;
}

// longest segment of every group over one or two group tables (host offsets [G+1])
std::vector<int> group_lengths(const std::vector<int64_t>& a, const std::vector<int64_t>* b, int G) {
    std::vector<int> n((size_t)G);
    for (int g = 0; g < G; ++g) {
        int64_t v = a[g + 1] - a[g];
        if (b) v = std::max(v, (*b)[g + 1] - (*b)[g]);
        n[g] = (int)v;
    }
    return n;
}

bool use_rs_path(int nmax, int64_t ld_max) {
    const char* e = sd_dev_env("SD_BCSD_PATH");  // "v1" forces the generic LDS-bitonic kernels (A/B testing)
    if (e && e[0] == 'v' && e[1] == '1') return false;
    if (ld_max >= ((int64_t)1 << 29)) return false;  // the fast kernels address rows with a 32-bit byte pitch
    return sd_bcsd_rs_supported(nmax);
}

double h_pp_denom(int n) { return ((double)n + 1.0 - kAlpha) - kBeta; }
double h_pp_at(int i, double denom) { return ((double)(i + 1) - kAlpha) / denom; }

// Inverse-CDF lookup tables (quantile.py:523-545): for every (group, rank r of the predict segment)
// the fitted-CDF bracket index and interpolation weight; identical for all cells.
//   idx >= 0 : value = ys[idx] + w * (ys[idx+1] - ys[idx])   (w == 0: exact hit / last point)
//   idx = -1 / -2 : lower / upper OLS tail evaluated at p = val ; idx = -3 : group absent in fit
struct QTables {
    sd_scratch idx, val;
};
int build_q_tables(sd_ctx* ctx, const std::vector<int64_t>& off_f, const std::vector<int64_t>& off_p, int G, QTables* q,
                   int tails = SD_QT_TAIL_LOWER | SD_QT_TAIL_UPPER) {
    const int64_t Tp = off_p[G];
    std::vector<int32_t> qi(Tp);
    std::vector<double> qv(Tp);
    for (int g = 0; g < G; ++g) {
        const int n = (int)(off_f[g + 1] - off_f[g]), m = (int)(off_p[g + 1] - off_p[g]);
        const double dn = h_pp_denom(n), dm = h_pp_denom(m);
        for (int r = 0; r < m; ++r) {
            const int64_t t = off_p[g] + r;
            const double p = h_pp_at(r, dm);
            if (n == 0) { qi[t] = -3; qv[t] = 0.0; continue; }
            // beyond the fitted positions: the OLS line of that side, or -- extrapolate 'min' / 'max' / None / '1to1' --
            // np.interp's end value (quantile.py:527-530)
            if (p < h_pp_at(0, dn)) {
                if (tails & SD_QT_TAIL_LOWER) { qi[t] = -1; qv[t] = p; } else { qi[t] = 0; qv[t] = 0.0; }
                continue;
            }
            if (p > h_pp_at(n - 1, dn)) {
                if (tails & SD_QT_TAIL_UPPER) { qi[t] = -2; qv[t] = p; } else { qi[t] = n - 1; qv[t] = 0.0; }
                continue;
            }
            int i = (int)std::floor(p * dn + kAlpha) - 1;
            i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
            while (i + 1 < n && h_pp_at(i + 1, dn) <= p) ++i;
            while (i > 0 && h_pp_at(i, dn) > p) --i;
            const double pi = h_pp_at(i, dn);
            qi[t] = i;
            qv[t] = (i == n - 1 || pi == p) ? 0.0 : (p - pi) / (h_pp_at(i + 1, dn) - pi);
        }
    }
    SD_HIP(q->idx.alloc(ctx, sizeof(int32_t) * Tp));
    SD_HIP(q->val.alloc(ctx, sizeof(double) * Tp));
    SD_HIP(hipMemcpyAsync(q->idx.p, qi.data(), sizeof(int32_t) * Tp, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(hipMemcpyAsync(q->val.p, qv.data(), sizeof(double) * Tp, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

template <int W>
int launch_fit(sd_ctx* ctx, int kind, const double* X, const double* y, int64_t ld, const DevGroupTable& gt, int G,
               int64_t T, int64_t C, int stride, int return_anoms, sd_bcsd_state* st) {
    const size_t lds = (size_t)W * stride * sizeof(double);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fit_kernel<W>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)((C + W - 1) / W), (unsigned)G);
    SD_LAUNCH(ctx, "bcsd_fit_kernel", bcsd_fit_kernel<W>, grid, dim3(64 * W), lds, kind, X, y, ld,
              (const int32_t*)gt.order.p, (const int32_t*)gt.off.p, G, T, C, stride, return_anoms, st->ys, st->x_climo,
              st->y_climo, st->status);
    return SD_OK;
}

template <int W>
int launch_predict(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp, int64_t ld, const DevGroupTable& gt,
                   int stride, int32_t* status_p, double* out, int64_t ld_out) {
    const size_t lds = (size_t)2 * W * stride * sizeof(double) + (size_t)W * 64 * sizeof(double);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_predict_kernel<W>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)((st->C + W - 1) / W), (unsigned)st->G);
    SD_LAUNCH(ctx, "bcsd_predict_kernel", bcsd_predict_kernel<W>, grid, dim3(64 * W), lds, st->kind, Xp, ld,
              (const int32_t*)gt.order.p, (const int32_t*)gt.off.p, (const int32_t*)st->goff_dev, st->G, st->T, st->C,
              stride, st->return_anoms, (const double*)st->ys, (const double*)st->x_climo, (const double*)st->y_climo,
              (const int32_t*)st->status, status_p, out, ld_out, st->qt_tails, st->qt_endpoints);
    return SD_OK;
}

// ------------------------------------------------------------------------------------------------
// long segments (2 113 ... 19 456 samples per group, e.g. a whole 40-year daily series as one group):
// one 1024-thread workgroup per (cell, group), the workgroup merge sort of sd_sortnet.h on a single LDS
// array; the samples of a thread (K consecutive ones) and their shifts stay in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* red /* 16 doubles */, int lane, int wave) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
}

// Least-squares line of a blocked series (thread t holds samples j = K * t + i, valid while j < n) over j = 0 .. n-1:
// trend.py:51 (LinearRegression on np.arange(len(X))), centred sums like sd_wave.h: trend_line.  Every thread of the
// workgroup calls it (two block reductions).
template <int K>
__device__ __forceinline__ void block_trend_line(const double (&v)[K], int n, int tid, double* red, int lane, int wave, double* slope,
                                                 double* icpt) {
    const double tbar = 0.5 * (double)(n - 1);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) s += K * tid + i < n ? v[i] : 0.0;
    const double vbar = block_sum(s, red, lane, wave) / (double)n;
    double sxy = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = K * tid + i;
        sxy += j < n ? ((double)j - tbar) * (v[i] - vbar) : 0.0;
    }
    sxy = block_sum(sxy, red, lane, wave);
    const double dn = (double)n;
    const double sxx = dn * (dn * dn - 1.0) / 12.0;
    const double a = n > 1 ? sxy / sxx : 0.0;
    *slope = a;
    *icpt = vbar - a * tbar;
}

template <int K>
__global__ void __launch_bounds__(1024) bcsd_long_fit_kernel(int kind, const double* __restrict__ X, const double* __restrict__ y,
                                                             int64_t ld, const int32_t* __restrict__ order,
                                                             const int32_t* __restrict__ goff, int G, int64_t T, int64_t C,
                                                             int return_anoms, double* __restrict__ ys,
                                                             double* __restrict__ x_climo, double* __restrict__ y_climo,
                                                             int32_t* status, int detrend, double* __restrict__ y_trend) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int64_t c = blockIdx.x;
    const int g = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int beg = goff[g], n = goff[g + 1] - beg;
    if (n == 0) return;
    const int np = (n + K - 1) / K * K;
    double* buf = reinterpret_cast<double*>(smem_raw);  // np + 1 doubles
    int* xch = reinterpret_cast<int*>(buf + np + 1);     // nthr + 1 ints
    double* red = reinterpret_cast<double*>(xch);        // 16 doubles, used between the sorts' exchanges
    const int32_t* ord = order + beg;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    bool bad = false;
    if (X != nullptr) {  // x climatology (bcsd.py:222); PR only validates X
        double s = 0.0;
        for (int i = tid; i < n; i += nthr) {
            const double v = X[(int64_t)ord[i] * ld + c];
            bad |= !sd_finite(v);
            s += v;
        }
        s = block_sum(s, red, lane, wave);
        if (kind == SD_BCSD_TAS && tid == 0) x_climo[c * G + g] = s / (double)n;
    }
    double s = 0.0;
    for (int i = tid; i <= np; i += nthr) {
        double v = inf;
        if (i < n) {
            v = y[(int64_t)ord[i] * ld + c];
            bad |= !sd_finite(v);
            s += v;
        }
        buf[i] = v;
    }
    if (bad) atomicOr(&status[c], SDI_NONFINITE);
    s = block_sum(s, red, lane, wave);
    if (tid == 0) {
        const double m = s / (double)n;
        y_climo[c * G + g] = m;  // bcsd.py:223 / 138
        if (kind == SD_BCSD_PR && return_anoms && m <= 0.0) atomicOr(&status[c], SDI_BAD_CLIMO);  // bcsd.py:140-141
    }
    double v[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = K * tid + i;
        v[i] = buf[j < np ? j : np];
    }
    __syncthreads();
    if (detrend) {  // quantile.py:95-98: the CDF is fitted on y minus its least-squares line (trend.py:65,83)
        double a, b;
        block_trend_line<K>(v, n, tid, red, lane, wave, &a, &b);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            if (j < n) v[i] = v[i] - ((double)j * a + b);
        }
        if (tid == 0) {
            y_trend[2 * (c * G + g)] = a;
            y_trend[2 * (c * G + g) + 1] = b;
        }
        __syncthreads();  // red aliases the sort's exchange area
    }
    sdsort::block_merge_sort<K>(v, buf, np, xch, tid, nthr);  // quantile.py:462 np.sort
    double* dst = ys + c * T + beg;
    for (int i = tid; i < n; i += nthr) dst[i] = buf[i];
}

template <int K>
__global__ void __launch_bounds__(1024) bcsd_long_predict_kernel(
    int kind, const double* __restrict__ Xp, int64_t ld, const int32_t* __restrict__ order_p,
    const int32_t* __restrict__ goff_p, const int32_t* __restrict__ goff_f, int G, int64_t Tf, int64_t C, int return_anoms,
    const double* __restrict__ ys, const double* __restrict__ x_climo, const double* __restrict__ y_climo,
    const int32_t* __restrict__ fit_status, int32_t* status, double* __restrict__ out, int64_t ld_out, int detrend,
    const double* __restrict__ y_trend, int qt_tails, int qt_endpoints) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int64_t c = blockIdx.x;
    const int g = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int begp = goff_p[g], m = goff_p[g + 1] - begp;
    const int begf = goff_f[g], n = goff_f[g + 1] - begf;
    if (m == 0) return;
    const int np = (m + K - 1) / K * K;
    double* buf = reinterpret_cast<double*>(smem_raw);  // np + 1 doubles
    int* xch = reinterpret_cast<int*>(buf + np + 1);     // nthr + 1 ints
    __shared__ TailFit tf_s;
    const int32_t* ord = order_p + begp;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    bool bad = false;
    for (int i = tid; i <= np; i += nthr) {
        double v = 0.0;
        if (i < m) {
            v = Xp[(int64_t)ord[i] * ld + c];
            bad |= !sd_finite(v);
        }
        buf[i] = v;
    }
    if (bad) atomicOr(&status[c], SDI_NONFINITE);
    const bool active = n > 0 && fit_status[c] == 0;
    const double* ysg = ys + c * Tf + begf;
    const double dn = pp_denom(n), dm = pp_denom(m);
    if (tid == 0) {
        TailFit tf = {0.0, 0.0, 0.0, 0.0, qt_tails};
        if (active && m > n) {  // p can leave [pp_0, pp_{n-1}] only when the predict segment is longer
            const int e = n < qt_endpoints ? n : qt_endpoints;
            ols_line(ysg, 0, e, dn, &tf.slope_lo, &tf.icpt_lo);
            ols_line(ysg, n - e, e, dn, &tf.slope_hi, &tf.icpt_hi);
        }
        tf_s = tf;
    }
    __syncthreads();
    const double xc = kind == SD_BCSD_TAS ? x_climo[c * G + g] : 0.0;
    const double yc = y_climo[c * G + g];
    // u = X - (rolling mean - x_climo) (bcsd.py:247-256); PR maps raw X (bcsd.py:167)
    double u[K], shift[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = K * tid + i;
        double x = 0.0, sh = 0.0;
        if (j < m) {
            x = buf[j];
            if (kind == SD_BCSD_TAS) sh = rolling9(buf, m, j) - xc;
        }
        shift[i] = sh;
        u[i] = j < m ? x - sh : inf;
    }
    __syncthreads();
    double ta = 0.0, tb = 0.0;  // detrend: the predict segment's own line (quantile.py:128-132)
    if (detrend) {
        block_trend_line<K>(u, m, tid, reinterpret_cast<double*>(xch), lane, wave, &ta, &tb);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int j = K * tid + i;
            if (j < m) u[i] = u[i] - ((double)j * ta + tb);  // trend.py:65,83
        }
        __syncthreads();
    }
    {
        double v[K];
#pragma unroll
        for (int i = 0; i < K; ++i) v[i] = u[i];
        sdsort::block_merge_sort<K>(v, buf, np, xch, tid, nthr);  // self ECDF: np.sort(u) (quantile.py:462 via 505-521)
    }
    // rank = (#sorted <= u) - 1: np.interp's exact-hit index = max rank among ties (quantile.py:488)
    int pos[K];
#pragma unroll
    for (int i = 0; i < K; ++i) pos[i] = -1;  // index of the last element known to be <= u
#pragma unroll 1
    for (int len = m; len > 1;) {
        int half = len >> 1;
        if ((half & 15) == 0) --half;  // keep the probe strides off the LDS bank period
        len -= half;
#pragma unroll
        for (int i = 0; i < K; ++i) pos[i] += buf[pos[i] + half] <= u[i] ? half : 0;
    }
    const TailFit tf = tf_s;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = K * tid + i;
        if (j >= m) continue;
        double res = nan;
        if (active) {
            const int cnt = pos[i] + 1 + (buf[pos[i] + 1] <= u[i] ? 1 : 0);
            const int r = cnt > 0 ? cnt - 1 : 0;
            double q = inverse_cdf(pp_at(r, dm), ysg, n, dn, tf);
            if (detrend)  // quantile.py:140-145: the predict line comes back, re-based on the fitted intercept
                q = (q + ((double)j * ta + tb)) - (tb - y_trend[2 * (c * G + g) + 1]);
            if (kind == SD_BCSD_TAS) {
                res = shift[i] + q;                // bcsd.py:263
                if (return_anoms) res = res - yc;  // bcsd.py:266-267
            } else {
                res = return_anoms ? q / yc : q;   // bcsd.py:170-185
            }
        }
        out[(int64_t)ord[j] * ld_out + c] = res;
    }
}

// register widths of the workgroup sort: n <= 1024 * K and the keys fit the LDS
int long_width(int nmax, size_t lds_max) {
    const int widths[] = {3, 5, 9, 13, 15, 17, 19};
    for (int K : widths) {
        const int64_t np = ((int64_t)nmax + K - 1) / K * K;
        if (nmax <= 1024 * K && sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025 + 64 <= lds_max) return K;
    }
    return 0;
}

template <int K>
int launch_long_fit(sd_ctx* ctx, int kind, const double* X, const double* y, int64_t ld, const DevGroupTable& gt, int G, int64_t T,
                    int64_t C, int return_anoms, sd_bcsd_state* st) {
    const int np = (gt.nmax + K - 1) / K * K;
    const size_t lds = sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_long_fit_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    SD_LAUNCH(ctx, "bcsd_long_fit_kernel", bcsd_long_fit_kernel<K>, dim3((unsigned)C, (unsigned)G), dim3(1024), lds, kind, X, y, ld,
              (const int32_t*)gt.order.p, (const int32_t*)gt.off.p, G, T, C, return_anoms, st->ys, st->x_climo, st->y_climo,
              st->status, st->detrend, st->y_trend);
    return SD_OK;
}

template <int K>
int launch_long_predict(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp, int64_t ld, const DevGroupTable& gt,
                        int32_t* status_p, double* out, int64_t ld_out) {
    const int np = (gt.nmax + K - 1) / K * K;
    const size_t lds = sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1025;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_long_predict_kernel<K>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "bcsd_long_predict_kernel", bcsd_long_predict_kernel<K>, dim3((unsigned)st->C, (unsigned)st->G), dim3(1024), lds,
              st->kind, Xp, ld, (const int32_t*)gt.order.p, (const int32_t*)gt.off.p, (const int32_t*)st->goff_dev, st->G, st->T,
              st->C, st->return_anoms, (const double*)st->ys, (const double*)st->x_climo, (const double*)st->y_climo,
              (const int32_t*)st->status, status_p, out, ld_out, st->detrend, (const double*)st->y_trend, st->qt_tails, st->qt_endpoints);
    return SD_OK;
}

#define SD_LONG_DISPATCH(K, fn, ...)                       \
    switch (K) {                                           \
        case 3: SD_TRY(fn<3>(__VA_ARGS__)); break;         \
        case 5: SD_TRY(fn<5>(__VA_ARGS__)); break;         \
        case 9: SD_TRY(fn<9>(__VA_ARGS__)); break;         \
        case 13: SD_TRY(fn<13>(__VA_ARGS__)); break;       \
        case 15: SD_TRY(fn<15>(__VA_ARGS__)); break;       \
        case 17: SD_TRY(fn<17>(__VA_ARGS__)); break;       \
        default: SD_TRY(fn<19>(__VA_ARGS__)); break;       \
    }

bool use_long_path(int nmax, size_t lds_max) {
    const char* e = sd_dev_env("SD_BCSD_PATH");  // "v1" keeps the generic LDS-bitonic kernels
    if (e && e[0] == 'v' && e[1] == '1') return false;
    return nmax > 64 * 33 && long_width(nmax, lds_max) != 0;
}

// options = SD_BCSD_RETURN_ANOMS | SD_BCSD_QM_DETREND bits (the public `return_anoms` argument of the fit entry points)
int alloc_state(sd_ctx* ctx, int kind, int G, int64_t T, int64_t C, int options, sd_bcsd_state** out) {
    *out = nullptr;
    SD_CHECK_ARG(options >= 0 && options <= 3, "options %d: expected a combination of SD_BCSD_RETURN_ANOMS and SD_BCSD_QM_DETREND", options);
    const int return_anoms = options & SD_BCSD_RETURN_ANOMS;
    sd_bcsd_state* st = new sd_bcsd_state();
    st->detrend = (options & SD_BCSD_QM_DETREND) ? 1 : 0;
    st->ctx = ctx;
    st->kind = kind;
    st->G = G;
    st->T = T;
    st->C = C;
    st->return_anoms = return_anoms;
    *out = st;
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->ys, sizeof(double) * T * C));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->x_climo, sizeof(double) * G * C));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->y_climo, sizeof(double) * G * C));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->y_trend, sizeof(double) * 2 * G * C));
    SD_HIP(hipMemsetAsync(st->y_trend, 0, sizeof(double) * 2 * G * C, ctx->stream));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->status, sizeof(int32_t) * C));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->goff_dev, sizeof(int32_t) * (G + 1)));
    SD_HIP(hipMemsetAsync(st->x_climo, 0, sizeof(double) * G * C, ctx->stream));
    SD_HIP(hipMemsetAsync(st->y_climo, 0, sizeof(double) * G * C, ctx->stream));
    return SD_OK;
}

}  // namespace

static int fit_with_table(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld, const DevGroupTable& gt, int G,
                          int64_t T_rows, int64_t C, int return_anoms, sd_bcsd_state** out);
static int predict_with_table(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp_dev, int64_t ld, const DevGroupTable& gt, int64_t Tp,
                              double* out_dev, int64_t ld_out, int32_t* status_p);
static int finish_predict(sd_ctx* ctx, const sd_bcsd_state* st, const int32_t* status_p, int64_t Tp, double* out_dev, int64_t ld_out,
                          int32_t* cell_status);

extern "C" {

int sd_bcsd_state_destroy(sd_bcsd_state* st) {
    if (!st) return SD_OK;
    if (st->ctx) {
        (void)hipSetDevice(st->ctx->device);
        (void)hipStreamSynchronize(st->ctx->stream);
    }
    sd_pool_release(st->ctx, st->ys);
    sd_pool_release(st->ctx, st->x_climo);
    sd_pool_release(st->ctx, st->y_climo);
    sd_pool_release(st->ctx, st->y_trend);
    sd_pool_release(st->ctx, st->status);
    sd_pool_release(st->ctx, st->goff_dev);
    delete st;
    return SD_OK;
}

int sd_bcsd_fit_dev(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld,
                    const int32_t* group_id, int G, int64_t T, int64_t C, int return_anoms, sd_bcsd_state** out) {
    SD_CHECK_ARG(ctx && y_dev && group_id && out, "sd_bcsd_fit: NULL argument");
    SD_CHECK_ARG(kind == SD_BCSD_TAS || kind == SD_BCSD_PR, "sd_bcsd_fit: unknown kind %d", kind);
    SD_CHECK_ARG(kind == SD_BCSD_PR || X_dev, "sd_bcsd_fit: BcsdTemperature needs X");
    SD_CHECK_ARG(T > 0 && C > 0 && G > 0 && ld >= C, "sd_bcsd_fit: bad sizes T=%lld C=%lld G=%d ld=%lld", (long long)T,
                 (long long)C, G, (long long)ld);
    *out = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    DevGroupTable gt;
    SD_TRY(upload_group_table(ctx, group_id, T, G, &gt));
    return fit_with_table(ctx, kind, X_dev, y_dev, ld, gt, G, T, C, return_anoms, out);
}

}  // extern "C"

// the kernels of a predict call on an uploaded predict group table (status_p: device [C], zeroed)
static int predict_with_table(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp_dev, int64_t ld, const DevGroupTable& gt, int64_t Tp,
                              double* out_dev, int64_t ld_out, int32_t* status_p) {
    const int64_t C = st->C;
    int W = 0, stride = 0;
    const int nmax_all = gt.nmax > st->nmax ? gt.nmax : st->nmax;
    const bool rs = use_rs_path(nmax_all, ld > ld_out ? ld : ld_out);
    const bool lng = !rs && use_long_path(nmax_all, ctx->lds_max) && long_width(gt.nmax, ctx->lds_max) != 0;
    if (st->detrend && !rs && !lng)
        return sd_set_error(SD_ERR_UNSUPPORTED, "detrended quantile mapping serves group segments of up to %d samples (longest here: %d)",
                            1024 * 19, nmax_all);
    if (!rs && !lng) SD_TRY(pick_tile_width(ctx->lds_max, gt.nmax, 2, &W, &stride));
    QTables qt;
    if (rs) {
        const bool identity = st->goff == gt.host_off;  // equal fit / predict group lengths: no inverse-CDF tables needed
        if (!identity) SD_TRY(build_q_tables(ctx, st->goff, gt.host_off, st->G, &qt, st->qt_tails));
        sdrs::Params p = {};
        p.n_endpoints = st->qt_endpoints;
        p.kind = st->kind; p.G = st->G; p.return_anoms = st->return_anoms; p.RS = sd_bcsd_rs_row_stride(nmax_all);
        p.C = C; p.Tf = st->T; p.ntiles = (C + 7) / 8;
        p.Xp = Xp_dev; p.ld_p = ld; p.out = out_dev; p.ld_out = ld_out;
        p.off_f = (const int32_t*)st->goff_dev;
        p.ord_p = (const int32_t*)gt.order.p; p.off_p = (const int32_t*)gt.off.p;
        p.qidx = qt.idx.as<int32_t>(); p.qval = qt.val.as<double>();
        p.ys = st->ys; p.x_climo = st->x_climo; p.y_climo = st->y_climo;
        p.status_fit = st->status; p.status_p = status_p;
        p.identity = identity ? 1 : 0;
        p.from_state = 1;
        p.detrend = st->detrend; p.y_trend = st->y_trend;
        const bool fused = use_fused_path(nmax_all, st->detrend != 0);
        RsWorkspace w;
        SD_TRY(carve_workspace(ctx, nmax_all, C, st->G, fused, false, st->detrend != 0, &w));
        p.ranks = w.ranks; p.shift = w.shift; p.trend_u = w.trend_u;
        p.worklist = w.worklist; p.work_count = w.work_count; p.work_cap = w.work_cap;
        p.worklist2 = w.worklist2; p.work_count2 = w.work_count2;
        const std::vector<int> glen = group_lengths(st->goff, &gt.host_off, st->G);
        SD_TRY(run_predict_kernels(ctx, p, fused, nmax_all, glen));
        SD_HIP(hipStreamSynchronize(ctx->stream));  // the inverse-CDF tables go back to the block cache
    } else if (lng) {
        SD_LONG_DISPATCH(long_width(gt.nmax, ctx->lds_max), launch_long_predict, ctx, st, Xp_dev, ld, gt, status_p, out_dev, ld_out);
    } else
    switch (W) {
        case 8: SD_TRY(launch_predict<8>(ctx, st, Xp_dev, ld, gt, stride, status_p, out_dev, ld_out)); break;
        case 4: SD_TRY(launch_predict<4>(ctx, st, Xp_dev, ld, gt, stride, status_p, out_dev, ld_out)); break;
        case 2: SD_TRY(launch_predict<2>(ctx, st, Xp_dev, ld, gt, stride, status_p, out_dev, ld_out)); break;
        default: SD_TRY(launch_predict<1>(ctx, st, Xp_dev, ld, gt, stride, status_p, out_dev, ld_out)); break;
    }
    return SD_OK;
}

// cells that are masked / failed in fit or non-finite in predict -> NaN columns; public status codes to the host
static int finish_predict(sd_ctx* ctx, const sd_bcsd_state* st, const int32_t* status_p, int64_t Tp, double* out_dev, int64_t ld_out,
                          int32_t* cell_status) {
    const int64_t C = st->C;
    sd_scratch status_pub;
    SD_LAUNCH(ctx, "nan_fill_kernel", nan_fill_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), 0, out_dev, ld_out, Tp, C,
              (const int32_t*)st->status, status_p);
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "status_public_kernel", status_public_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                  (const int32_t*)st->status, status_p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

// fit on an uploaded group table; the state's series length is the table's entry count (= T unless groups overlap)
static int fit_with_table(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld, const DevGroupTable& gt, int G,
                          int64_t T_rows, int64_t C, int options, sd_bcsd_state** out) {
    const int64_t T = gt.host_off[G];
    (void)T_rows;
    const int return_anoms = options & SD_BCSD_RETURN_ANOMS;
    int W = 0, stride = 0;
    const bool rs = use_rs_path(gt.nmax, ld);
    const bool lng = !rs && use_long_path(gt.nmax, ctx->lds_max);
    if (!rs && !lng) SD_TRY(pick_tile_width(ctx->lds_max, gt.nmax, 1, &W, &stride));
    sd_bcsd_state* st = nullptr;
    int rc = alloc_state(ctx, kind, G, T, C, options, &st);
    if (rc == SD_OK && st->detrend && !rs && !lng)
        rc = sd_set_error(SD_ERR_UNSUPPORTED, "detrended quantile mapping serves group segments of up to %d samples (longest here: %d)",
                          1024 * 19, gt.nmax);
    if (rc != SD_OK) {
        sd_bcsd_state_destroy(st);
        return rc;
    }
    st->goff = gt.host_off;
    st->nmax = gt.nmax;
    auto body = [&]() -> int {
        SD_HIP(hipMemcpyAsync(st->goff_dev, gt.off.p, sizeof(int32_t) * (G + 1), hipMemcpyDeviceToDevice, ctx->stream));
        const double* first = X_dev ? X_dev : y_dev;
        SD_LAUNCH(ctx, "bcsd_mask_kernel", bcsd_mask_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, first, C,
                  st->status);
        if (rs) {
            sdrs::Params p = {};
            p.kind = kind; p.G = G; p.return_anoms = return_anoms; p.RS = sd_bcsd_rs_row_stride(gt.nmax);
            p.C = C; p.Tf = T; p.ntiles = (C + 7) / 8;
            p.X = X_dev; p.y = y_dev; p.ld = ld;
            p.ord_f = (const int32_t*)gt.order.p; p.off_f = (const int32_t*)gt.off.p;
            p.ys = st->ys; p.x_climo = st->x_climo; p.y_climo = st->y_climo; p.status_fit = st->status;
            p.detrend = st->detrend; p.y_trend = st->y_trend;
            SD_TRY(sd_bcsd_rs_launch(ctx, sdrs::MODE_FIT, p, gt.nmax, group_lengths(gt.host_off, nullptr, G).data()));
        } else if (lng) {
            SD_LONG_DISPATCH(long_width(gt.nmax, ctx->lds_max), launch_long_fit, ctx, kind, X_dev, y_dev, ld, gt, G, T, C, return_anoms, st);
        } else
        switch (W) {
            case 8: SD_TRY(launch_fit<8>(ctx, kind, X_dev, y_dev, ld, gt, G, T, C, stride, return_anoms, st)); break;
            case 4: SD_TRY(launch_fit<4>(ctx, kind, X_dev, y_dev, ld, gt, G, T, C, stride, return_anoms, st)); break;
            case 2: SD_TRY(launch_fit<2>(ctx, kind, X_dev, y_dev, ld, gt, G, T, C, stride, return_anoms, st)); break;
            default: SD_TRY(launch_fit<1>(ctx, kind, X_dev, y_dev, ld, gt, G, T, C, stride, return_anoms, st)); break;
        }
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    };
    rc = body();
    if (rc != SD_OK) {
        sd_bcsd_state_destroy(st);
        return rc;
    }
    *out = st;
    return SD_OK;
}

extern "C" {

int sd_bcsd_predict_dev(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp_dev, int64_t ld,
                        const int32_t* group_id_p, int64_t Tp, double* out_dev, int64_t ld_out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xp_dev && group_id_p && out_dev, "sd_bcsd_predict: NULL argument");
    SD_CHECK_ARG(Tp > 0 && ld >= st->C && ld_out >= st->C, "sd_bcsd_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C;
    DevGroupTable gt;
    SD_TRY(upload_group_table(ctx, group_id_p, Tp, st->G, &gt));
    sd_scratch status_p;
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    SD_TRY(predict_with_table(ctx, st, Xp_dev, ld, gt, Tp, out_dev, ld_out, status_p.as<int32_t>()));
    return finish_predict(ctx, st, status_p.as<int32_t>(), Tp, out_dev, ld_out, cell_status);
}

int sd_bcsd_predict_trend_dev(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp_dev, int64_t ld, const int32_t* group_id_p,
                              const int32_t* trend_group_id, int G_trend, int64_t Tp, double* out_dev, int64_t ld_out,
                              int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xp_dev && group_id_p && trend_group_id && out_dev, "sd_bcsd_predict_trend: NULL argument");
    SD_CHECK_ARG(Tp > 0 && G_trend > 0 && ld >= st->C && ld_out >= st->C, "sd_bcsd_predict_trend: bad sizes");
    if (st->kind != SD_BCSD_TAS)  // BcsdPrecipitation has no climate-trend shift (bcsd.py:149-170)
        return sd_bcsd_predict_dev(ctx, st, Xp_dev, ld, group_id_p, Tp, out_dev, ld_out, cell_status);
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C;
    DevGroupTable gq, gr;
    SD_TRY(upload_group_table(ctx, group_id_p, Tp, st->G, &gq));
    SD_TRY(upload_group_table(ctx, trend_group_id, Tp, G_trend, &gr));
    sd_scratch status_p, u, shift, gidq;
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    SD_HIP(u.alloc(ctx, sizeof(double) * (size_t)Tp * (size_t)C));
    SD_HIP(shift.alloc(ctx, sizeof(double) * (size_t)Tp * (size_t)C));
    SD_HIP(gidq.alloc(ctx, sizeof(int32_t) * (size_t)Tp));
    SD_HIP(hipMemcpyAsync(gidq.p, group_id_p, sizeof(int32_t) * (size_t)Tp, hipMemcpyHostToDevice, ctx->stream));
    // shift = rolling mean over the trend groups - x_climo of the sample's quantile-mapping group (bcsd.py:247-253),
    // u = X - shift (bcsd.py:256)
    SD_LAUNCH(ctx, "bcsd_trend_shift_kernel", bcsd_trend_shift_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)G_trend), dim3(256), 0,
              Xp_dev, ld, (const int32_t*)gr.order.p, (const int32_t*)gr.off.p, (const int32_t*)gidq.p, st->G,
              (const double*)st->x_climo, C, shift.as<double>(), u.as<double>());
    // quantile mapping of u by the time grouper's groups (bcsd.py:260): the precipitation path of the kernels (no shift)
    sd_bcsd_state qm = *st;
    qm.kind = SD_BCSD_PR;
    qm.return_anoms = 0;
    SD_TRY(predict_with_table(ctx, &qm, u.as<double>(), C, gq, Tp, out_dev, ld_out, status_p.as<int32_t>()));
    // restore the shift (bcsd.py:263), remove the target climatology (bcsd.py:266-267)
    SD_LAUNCH(ctx, "bcsd_trend_restore_kernel", bcsd_trend_restore_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)((Tp + 63) / 64)),
              dim3(256), 0, out_dev, ld_out, shift.as<double>(), (const int32_t*)gidq.p, st->G, (const double*)st->y_climo,
              st->return_anoms, Tp, C);
    return finish_predict(ctx, st, status_p.as<int32_t>(), Tp, out_dev, ld_out, cell_status);
}

int sd_bcsd_fit_predict_dev(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld,
                            const int32_t* group_id, int G, int64_t T, int64_t C, int return_anoms,
                            const double* Xp_dev, int64_t ld_p, const int32_t* group_id_p, int64_t Tp,
                            double* out_dev, int64_t ld_out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && y_dev && group_id && Xp_dev && group_id_p && out_dev, "sd_bcsd_fit_predict: NULL argument");
    SD_CHECK_ARG(kind == SD_BCSD_TAS || kind == SD_BCSD_PR, "sd_bcsd_fit_predict: unknown kind %d", kind);
    SD_CHECK_ARG(kind == SD_BCSD_PR || X_dev, "sd_bcsd_fit_predict: BcsdTemperature needs X");
    SD_CHECK_ARG(T > 0 && Tp > 0 && C > 0 && G > 0 && ld >= C && ld_p >= C && ld_out >= C, "sd_bcsd_fit_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    DevGroupTable gf, gp;
    SD_TRY(upload_group_table(ctx, group_id, T, G, &gf));
    SD_TRY(upload_group_table(ctx, group_id_p, Tp, G, &gp));
    const int nmax_all = gf.nmax > gp.nmax ? gf.nmax : gp.nmax;
    SD_CHECK_ARG(return_anoms >= 0 && return_anoms <= 3, "sd_bcsd_fit_predict: options %d", return_anoms);
    const bool detrend = (return_anoms & SD_BCSD_QM_DETREND) != 0;
    if (!use_rs_path(nmax_all, std::max(ld, std::max(ld_p, ld_out)))) {
        // generic path: fit then predict through a transient state
        sd_bcsd_state* st = nullptr;
        SD_TRY(sd_bcsd_fit_dev(ctx, kind, X_dev, y_dev, ld, group_id, G, T, C, return_anoms, &st));
        int rc = sd_bcsd_predict_dev(ctx, st, Xp_dev, ld_p, group_id_p, Tp, out_dev, ld_out, cell_status);
        sd_bcsd_state_destroy(st);
        return rc;
    }
    // fused register/LDS path: no persisted quantile state, HBM traffic = 3 reads + 1 write per sample
    sd_scratch status_f, status_p, status_pub;
    SD_HIP(status_f.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    QTables qt;
    const bool identity = gf.host_off == gp.host_off;  // equal fit / predict group lengths: no inverse-CDF tables needed
    if (!identity) SD_TRY(build_q_tables(ctx, gf.host_off, gp.host_off, G, &qt));
    const double* first = X_dev ? X_dev : y_dev;
    SD_LAUNCH(ctx, "bcsd_mask_kernel", bcsd_mask_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, first, C,
              status_f.as<int32_t>());
    sdrs::Params p = {};
    p.kind = kind; p.G = G; p.return_anoms = return_anoms & SD_BCSD_RETURN_ANOMS; p.RS = sd_bcsd_rs_row_stride(nmax_all);
    p.C = C; p.Tf = T; p.ntiles = (C + 7) / 8;
    p.X = X_dev; p.y = y_dev; p.ld = ld;
    p.detrend = detrend ? 1 : 0;
    p.Xp = Xp_dev; p.ld_p = ld_p; p.out = out_dev; p.ld_out = ld_out;
    p.ord_f = (const int32_t*)gf.order.p; p.off_f = (const int32_t*)gf.off.p;
    p.ord_p = (const int32_t*)gp.order.p; p.off_p = (const int32_t*)gp.off.p;
    p.qidx = qt.idx.as<int32_t>(); p.qval = qt.val.as<double>();
    p.status_fit = status_f.as<int32_t>(); p.status_p = status_p.as<int32_t>();
    p.identity = identity ? 1 : 0;
    {
        // No persisted sorted state.  One fused kernel per segment (no hand-off at all); for the segments it hands back
        // RANK writes 2 bytes/sample (rank of every x_fut sample in its shifted
        // segment) + x_climo, APPLY sorts y_obs on chip, maps the ranks and restores the shift.
        const bool fused = use_fused_path(nmax_all, detrend);
        RsWorkspace w;
        SD_TRY(carve_workspace(ctx, nmax_all, C, G, fused, true, detrend, &w));
        p.ranks = w.ranks; p.shift = w.shift; p.x_climo = w.x_climo; p.trend_u = w.trend_u;
        p.worklist = w.worklist; p.work_count = w.work_count; p.work_cap = w.work_cap;
        p.worklist2 = w.worklist2; p.work_count2 = w.work_count2;
        const std::vector<int> glen = group_lengths(gf.host_off, &gp.host_off, G);
        SD_TRY(run_predict_kernels(ctx, p, fused, nmax_all, glen));
    }
    SD_LAUNCH(ctx, "nan_fill_kernel", nan_fill_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), 0, out_dev, ld_out, Tp, C,
              (const int32_t*)status_f.p, (const int32_t*)status_p.p);
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "status_public_kernel", status_public_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                  (const int32_t*)status_f.p, (const int32_t*)status_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_bcsd_fit(sd_ctx* ctx, int kind, const double* X, const double* y, const int32_t* group_id, int G, int64_t T,
                int64_t C, int return_anoms, sd_bcsd_state** out) {
    SD_CHECK_ARG(ctx && y && group_id && out, "sd_bcsd_fit: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0, "sd_bcsd_fit: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy;
    const size_t bytes = sizeof(double) * (size_t)T * (size_t)C;
    if (X) {
        SD_HIP(dX.alloc(ctx, bytes));
        SD_TRY(sd_copy_h2d(ctx, dX.p, X, bytes));
    }
    SD_HIP(dy.alloc(ctx, bytes));
    SD_TRY(sd_copy_h2d(ctx, dy.p, y, bytes));
    return sd_bcsd_fit_dev(ctx, kind, dX.as<double>(), dy.as<double>(), C, group_id, G, T, C, return_anoms, out);
}

// Host-buffer predict.  Large grids go through in blocks of cells, pipelined over the two directions of the PCIe link: while the
// kernels of block i run and block i - 1 of the result drains to the host (a second host thread, ring and stream:
// sd_copy_d2h_2d), block i + 1 of X comes in.  A block of cells is a column block of the row-major fields and a slice of the
// cell-major state.
int sd_bcsd_predict(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp, const int32_t* group_id_p, int64_t Tp,
                    double* out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xp && group_id_p && out, "sd_bcsd_predict: NULL argument");
    SD_CHECK_ARG(Tp > 0, "sd_bcsd_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C;
    const size_t bytes = sizeof(double) * (size_t)Tp * (size_t)C;
    const int nblk = (bytes >= ((size_t)256 << 20) && C >= 1024) ? 4 : 1;
    if (nblk == 1) {
        sd_scratch dX, dout;
        SD_HIP(dX.alloc(ctx, bytes));
        SD_HIP(dout.alloc(ctx, bytes));
        SD_TRY(sd_copy_h2d(ctx, dX.p, Xp, bytes));
        SD_TRY(sd_bcsd_predict_dev(ctx, st, dX.as<double>(), C, group_id_p, Tp, dout.as<double>(), C, cell_status));
        SD_TRY(sd_copy_d2h(ctx, out, dout.p, bytes));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    }
    sd_advise_result_buffer(out, bytes);
    const size_t pitch = sizeof(double) * (size_t)C;
    const int64_t per = ((C + nblk - 1) / nblk + 7) / 8 * 8;  // blocks of whole tiles
    // packed [Tp, cells of the block] device buffers: the transfers are linear, the kernels take the block's width as pitch
    sd_scratch dX, dout[2];
    SD_HIP(dX.alloc(ctx, sizeof(double) * (size_t)Tp * (size_t)per));
    SD_HIP(dout[0].alloc(ctx, sizeof(double) * (size_t)Tp * (size_t)per));
    SD_HIP(dout[1].alloc(ctx, sizeof(double) * (size_t)Tp * (size_t)per));
    std::thread drain;
    int drain_rc = SD_OK;
    std::string drain_err;
    int rc = SD_OK;
    int blk = 0;
    for (int64_t c0 = 0; c0 < C && rc == SD_OK; c0 += per, ++blk) {
        const int64_t cw = std::min(per, C - c0);
        const size_t width = sizeof(double) * (size_t)cw;
        const auto t_a = std::chrono::steady_clock::now();
        rc = sd_copy_h2d_2d(ctx, dX.p, width, Xp + c0, pitch, width, (size_t)Tp);
        if (rc != SD_OK) break;
        const auto t_b = std::chrono::steady_clock::now();
        sd_bcsd_state view = *st;  // the block's slice of the state (cell-major arrays; nothing owned)
        view.C = cw;
        view.ys = st->ys + c0 * st->T;
        if (st->x_climo) view.x_climo = st->x_climo + c0 * st->G;
        if (st->y_climo) view.y_climo = st->y_climo + c0 * st->G;
        if (st->y_trend) view.y_trend = st->y_trend + c0 * st->G * 2;
        view.status = st->status + c0;
        // (dout[blk & 1] is free: the drain of block blk - 2 was joined in iteration blk - 1, after that block's kernels; the drain
        // of block blk - 1 -- the other buffer -- keeps running beside this block's kernels)
        double* res = dout[blk & 1].as<double>();
        rc = sd_bcsd_predict_dev(ctx, &view, dX.as<double>(), cw, group_id_p, Tp, res, cw,
                                 cell_status ? cell_status + c0 : nullptr);  // (returns when the block's kernels are done)
        if (rc != SD_OK) break;
        const auto t_c = std::chrono::steady_clock::now();
        if (drain.joinable()) drain.join();
        const auto t_d = std::chrono::steady_clock::now();
        if (sd_dev_env("SD_TRACE_PIPE")) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "block %d: h2d %.2f ms, predict %.2f ms, wait for previous drain %.2f ms\n", blk, ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d));
        }
        if (drain_rc != SD_OK) break;
        double* dst = out + c0;
        drain = std::thread([=, &drain_rc, &drain_err]() {
            drain_rc = sd_copy_d2h_2d(ctx, dst, pitch, res, width, width, (size_t)Tp);
            if (drain_rc != SD_OK) drain_err = sd_last_error();  // (the message is thread-local)
        });
    }
    if (drain.joinable()) drain.join();
    SD_HIP(hipStreamSynchronize(ctx->stream));
    if (rc != SD_OK) return rc;
    if (drain_rc != SD_OK) return sd_set_error(drain_rc, "%s", drain_err.c_str());
    return SD_OK;
}

int sd_bcsd_fit_groups_dev(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld, const int32_t* group_order,
                           const int64_t* group_offsets, int G, int64_t T, int64_t C, int return_anoms, sd_bcsd_state** out) {
    SD_CHECK_ARG(ctx && y_dev && group_order && group_offsets && out, "sd_bcsd_fit_groups: NULL argument");
    SD_CHECK_ARG(kind == SD_BCSD_TAS || kind == SD_BCSD_PR, "sd_bcsd_fit_groups: unknown kind %d", kind);
    SD_CHECK_ARG(kind == SD_BCSD_PR || X_dev, "sd_bcsd_fit_groups: BcsdTemperature needs X");
    SD_CHECK_ARG(T > 0 && C > 0 && G > 0 && ld >= C, "sd_bcsd_fit_groups: bad sizes T=%lld C=%lld G=%d ld=%lld", (long long)T,
                 (long long)C, G, (long long)ld);
    *out = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    DevGroupTable gt;
    SD_TRY(upload_explicit_table(ctx, group_order, group_offsets, G, T, &gt));
    return fit_with_table(ctx, kind, X_dev, y_dev, ld, gt, G, T, C, return_anoms, out);
}

int sd_bcsd_fit_groups(sd_ctx* ctx, int kind, const double* X, const double* y, const int32_t* group_order,
                       const int64_t* group_offsets, int G, int64_t T, int64_t C, int return_anoms, sd_bcsd_state** out) {
    SD_CHECK_ARG(ctx && y && group_order && group_offsets && out, "sd_bcsd_fit_groups: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0, "sd_bcsd_fit_groups: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy;
    const size_t bytes = sizeof(double) * (size_t)T * (size_t)C;
    if (X) {
        SD_HIP(dX.alloc(ctx, bytes));
        SD_TRY(sd_copy_h2d(ctx, dX.p, X, bytes));
    }
    SD_HIP(dy.alloc(ctx, bytes));
    SD_TRY(sd_copy_h2d(ctx, dy.p, y, bytes));
    return sd_bcsd_fit_groups_dev(ctx, kind, dX.as<double>(), dy.as<double>(), C, group_order, group_offsets, G, T, C, return_anoms, out);
}

int sd_bcsd_predict_trend(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp, const int32_t* group_id_p,
                          const int32_t* trend_group_id, int G_trend, int64_t Tp, double* out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xp && group_id_p && trend_group_id && out, "sd_bcsd_predict_trend: NULL argument");
    SD_CHECK_ARG(Tp > 0, "sd_bcsd_predict_trend: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dout;
    const size_t bytes = sizeof(double) * (size_t)Tp * (size_t)st->C;
    SD_HIP(dX.alloc(ctx, bytes));
    SD_HIP(dout.alloc(ctx, bytes));
    SD_TRY(sd_copy_h2d(ctx, dX.p, Xp, bytes));
    SD_TRY(sd_bcsd_predict_trend_dev(ctx, st, dX.as<double>(), st->C, group_id_p, trend_group_id, G_trend, Tp, dout.as<double>(), st->C,
                                     cell_status));
    SD_TRY(sd_copy_d2h(ctx, out, dout.p, bytes));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_bcsd_state_set_tails(sd_bcsd_state* st, int extrapolate, int n_endpoints) {
    SD_CHECK_ARG(st, "sd_bcsd_state_set_tails: NULL state");
    SD_CHECK_ARG(extrapolate >= 0 && extrapolate <= (SD_QT_TAIL_LOWER | SD_QT_TAIL_UPPER), "sd_bcsd_state_set_tails: extrapolate = %d", extrapolate);
    SD_CHECK_ARG(n_endpoints >= 1, "sd_bcsd_state_set_tails: n_endpoints = %d (a line needs at least one point)", n_endpoints);
    st->qt_tails = extrapolate;
    st->qt_endpoints = n_endpoints;
    return SD_OK;
}

int sd_bcsd_state_info(const sd_bcsd_state* st, int* kind, int* G, int64_t* T, int64_t* C, int* return_anoms) {
    SD_CHECK_ARG(st, "state is NULL");
    if (kind) *kind = st->kind;
    if (G) *G = st->G;
    if (T) *T = st->T;
    if (C) *C = st->C;
    if (return_anoms) *return_anoms = (st->return_anoms ? SD_BCSD_RETURN_ANOMS : 0) | (st->detrend ? SD_BCSD_QM_DETREND : 0);
    return SD_OK;
}

int sd_bcsd_state_status(const sd_bcsd_state* st, int32_t* cell_status) {
    SD_CHECK_ARG(st && cell_status, "sd_bcsd_state_status: NULL argument");
    SD_HIP(hipSetDevice(st->ctx->device));
    SD_HIP(hipMemcpyAsync(cell_status, st->status, sizeof(int32_t) * st->C, hipMemcpyDeviceToHost, st->ctx->stream));
    SD_HIP(hipStreamSynchronize(st->ctx->stream));
    for (int64_t c = 0; c < st->C; ++c) cell_status[c] = sd_public_status(cell_status[c]);
    return SD_OK;
}

int sd_bcsd_state_export(const sd_bcsd_state* st, double* y_sorted, double* x_climo, double* y_climo,
                         int32_t* cell_status, int64_t* group_offsets) {
    SD_CHECK_ARG(st, "state is NULL");
    sd_ctx* ctx = st->ctx;
    SD_HIP(hipSetDevice(ctx->device));
    if (y_sorted) SD_HIP(hipMemcpyAsync(y_sorted, st->ys, sizeof(double) * st->T * st->C, hipMemcpyDeviceToHost, ctx->stream));
    if (x_climo) SD_HIP(hipMemcpyAsync(x_climo, st->x_climo, sizeof(double) * st->G * st->C, hipMemcpyDeviceToHost, ctx->stream));
    if (y_climo) SD_HIP(hipMemcpyAsync(y_climo, st->y_climo, sizeof(double) * st->G * st->C, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    if (cell_status) SD_TRY(sd_bcsd_state_status(st, cell_status));
    if (group_offsets)
        for (int g = 0; g <= st->G; ++g) group_offsets[g] = st->goff[g];
    return SD_OK;
}

int sd_bcsd_state_get_trend(const sd_bcsd_state* st, double* y_trend) {
    SD_CHECK_ARG(st && y_trend, "sd_bcsd_state_get_trend: NULL argument");
    sd_ctx* ctx = st->ctx;
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipMemcpyAsync(y_trend, st->y_trend, sizeof(double) * 2 * st->G * st->C, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_bcsd_state_set_trend(sd_bcsd_state* st, const double* y_trend) {
    SD_CHECK_ARG(st && y_trend, "sd_bcsd_state_set_trend: NULL argument");
    SD_CHECK_ARG(st->detrend, "sd_bcsd_state_set_trend: the state was not created with SD_BCSD_QM_DETREND");
    sd_ctx* ctx = st->ctx;
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipMemcpyAsync(st->y_trend, y_trend, sizeof(double) * 2 * st->G * st->C, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_bcsd_state_import(sd_ctx* ctx, int kind, int G, int64_t T, int64_t C, int return_anoms, const double* y_sorted,
                         const double* x_climo, const double* y_climo, const int32_t* cell_status,
                         const int64_t* group_offsets, sd_bcsd_state** out) {
    SD_CHECK_ARG(ctx && y_sorted && y_climo && group_offsets && out, "sd_bcsd_state_import: NULL argument");
    SD_CHECK_ARG(kind == SD_BCSD_PR || x_climo, "sd_bcsd_state_import: BcsdTemperature needs x_climo");
    SD_CHECK_ARG(T > 0 && C > 0 && G > 0 && group_offsets[0] == 0 && group_offsets[G] == T, "sd_bcsd_state_import: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_bcsd_state* st = nullptr;
    int rc = alloc_state(ctx, kind, G, T, C, return_anoms, &st);
    if (rc != SD_OK) {
        sd_bcsd_state_destroy(st);
        return rc;
    }
    st->goff.assign(group_offsets, group_offsets + G + 1);
    std::vector<int32_t> off32(G + 1), bits(C, 0);
    st->nmax = 0;
    for (int g = 0; g <= G; ++g) off32[g] = (int32_t)group_offsets[g];
    for (int g = 0; g < G; ++g) st->nmax = std::max(st->nmax, off32[g + 1] - off32[g]);
    if (cell_status)
        for (int64_t c = 0; c < C; ++c) bits[c] = sd_internal_status(cell_status[c]);
    auto body = [&]() -> int {
        SD_HIP(hipMemcpyAsync(st->ys, y_sorted, sizeof(double) * T * C, hipMemcpyHostToDevice, ctx->stream));
        if (x_climo) SD_HIP(hipMemcpyAsync(st->x_climo, x_climo, sizeof(double) * G * C, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipMemcpyAsync(st->y_climo, y_climo, sizeof(double) * G * C, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipMemcpyAsync(st->status, bits.data(), sizeof(int32_t) * C, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipMemcpyAsync(st->goff_dev, off32.data(), sizeof(int32_t) * (G + 1), hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    };
    rc = body();
    if (rc != SD_OK) {
        sd_bcsd_state_destroy(st);
        return rc;
    }
    *out = st;
    return SD_OK;
}

}  // extern "C"
