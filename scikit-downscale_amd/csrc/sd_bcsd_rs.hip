// BCSD quantile mapping, register/LDS merge-sort kernels with an explicit rank search: one 64-lane wave per
// (cell, month) segment, 8 adjacent cells per 512-thread workgroup, two workgroups per CU (both the 160 KB LDS --
// 16 rows of 10 KB -- and the 128-VGPR budget allow 16 waves per CU).  Three kernels share one body (segment_body):
//   MODE_RANK   x side.  x_hist rows are streamed (16-byte loads, 4 lanes per 64-byte row fragment) and reduced
//               to the per-cell climatology -- never stored; the x_fut tile is loaded the same way and transposed
//               through LDS into one zero-padded row per cell; the owning wave takes K consecutive samples per
//               lane, applies the 9-sample rolling-mean shift (bcsd.py:247-256), sorts a copy and ranks every
//               sample in it by a branch-free, bank-conflict-free binary search (np.interp's "last xp <= x"
//               rule); ranks leave as two 16-bit values per word in lane layout.
//   MODE_APPLY  y side.  y rows are loaded/transposed, reduced to y_climo and sorted in the wave's LDS row; every
//               rank is mapped through the fitted inverse CDF (the sorted value itself when fit and predict groups
//               have equal lengths, else the precomputed index + weight table with OLS tails); the x_fut tile is
//               read a second time to rebuild the shift; the result goes out transposed.
//   MODE_FIT    the y side alone, writing the fitted state.
// They serve every fit that keeps a state, QuantileMapper(detrend=True), segments of 1 537 .. 2 112 samples, and the
// segments the fused kernels of sd_bcsd_fx.hip hand back through their work list (exact ties: the search gives the
// largest rank among them).
// Sort and tile helpers: sd_wave.h / sd_sortnet.h.
#include "sd_bcsd_rs.h"
#include "sd_wave.h"

namespace sdrs {

using namespace sdw;

// The kernel reads its Params straight from the kernarg segment (scalar loads on demand, re-loadable).
typedef const Params __attribute__((address_space(4)))* ParamsPtr;

// OCC = waves per SIMD the register allocation is capped for: 4 -> 128 VGPRs (2 workgroups per CU),
// 2 -> 256 VGPRs (1 workgroup per CU).
// IDENT: every group has the same length in fit and predict, so the fitted inverse CDF evaluated at the
// Cunnane position of rank r is exactly the r-th sorted observation (np.interp exact hit): no table needed.
//
// Hand-off between MODE_RANK and MODE_APPLY (context workspace, one slab per (cell, group) segment, written
// and read with the same lane layout so every access is a fully coalesced 256-byte wave transaction):
//   ranks: [segment][(K+1)/2][64] u32 -- two 16-bit ranks per word, exactly the rank2[] registers
template <int K, int MODE, int KIND, bool IDENT>
__device__ __forceinline__ void segment_body(ParamsPtr p, int64_t tile_id, int g, char* smem_raw) {
    static_assert(MODE == MODE_FIT || MODE == MODE_RANK || MODE == MODE_APPLY, "unknown mode");
    constexpr bool kTas = KIND == SD_BCSD_TAS;
    constexpr int CH = Chunk<K>::CH;
    constexpr int NR = (K + 1) / 2;
    double* const scratch = reinterpret_cast<double*>(smem_raw);  // 64 doubles
    const double* const rcp = scratch + 64;                       // 16 doubles: correctly rounded 1/c, c = 1..9
    double* const tile = scratch + kHeadDoubles;
    const int RS = p->RS;
    const int64_t c0 = tile_id * kW;
    const int t_ = tid_now();
    const int wave = t_ / kWave, lane = t_ % kWave;
    const int64_t c = c0 + wave;
    const bool cell_ok = c < p->C;
    double* const row = tile + wave * RS;
    const int64_t seg = c * p->G + g;
    const int begf = p->off_f[g];
    const int n = p->off_f[g + 1] - begf;
    int begp = 0, m = 0;
    if (MODE != MODE_FIT) {
        begp = p->off_p[g];
        m = p->off_p[g + 1] - begp;
    }
    const bool vec_f = (p->ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->y) & 15) == 0) &&
                       (p->X == nullptr || (reinterpret_cast<uintptr_t>(p->X) & 15) == 0);
    const bool vec_p = MODE != MODE_FIT && (p->ld_p % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->Xp) & 15) == 0);
    if (MODE != MODE_FIT ? m == 0 : n == 0) return;

    unsigned rank2[NR];  // two 16-bit ranks per register (segments are < 65536 samples)
    double xc = 0.0;
    if (MODE == MODE_RANK) {
        // ---- x climatology (bcsd.py:222); PR only validates X ------------------------------------------
        // The x_hist rows and then the x_fut tile are requested back to back (loads return in order), so the
        // column sums are reduced while the tile is still in flight: one exposed memory latency instead of two.
        TileRegs<NR> xf;
        if (p->from_state) {
            if (kTas && cell_ok) xc = p->x_climo[seg];
            tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf);
        } else if (p->X != nullptr && n > 0) {
            TileRegs<NR> xh;
            tile_issue<NR>(p->X, p->ld, p->ord_f + begf, n, c0, p->C, vec_f, xh);
            tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf);
            xc = tile_reduce_mean<NR>(xh, n, c0, p->C, scratch, p->status_fit, wave, lane);
            if (kTas && lane == 0 && cell_ok) p->x_climo[seg] = xc;
        } else {
            tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf);
        }
        // ---- x_fut segment -> shifted series u -> rank of every sample in sort(u) --------------------
        tile_commit<NR>(xf, m, c0, p->C, tile + kPadFront, RS, p->status_p);
        if (kTas) zero_pads(row, m, lane, CH + 4);
        __syncthreads();
        double u[K];  // u = X - (rolling mean - x_climo) (bcsd.py:247-256); PR maps raw X (bcsd.py:167)
#pragma unroll
        for (int cbeg = 0; cbeg < K; cbeg += CH) {
            double mean[CH], xv[CH];
            if (kTas) {
                rolling_from_lds<CH>(row, K * lane + cbeg, m, rcp, mean, xv);
            } else {
                const double* src = row + kPadFront + (K * lane + cbeg < m ? K * lane + cbeg : 0);
#pragma unroll
                for (int ii = 0; ii < CH; ++ii) xv[ii] = src[ii];
            }
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                const int i = cbeg + ii;
                if (i < K) {
                    double uv = xv[ii];
                    if (kTas) {
                        const double shift = mean[ii] - xc;  // bcsd.py:253
                        uv = xv[ii] - shift;                 // bcsd.py:256
                    }
                    u[i] = K * lane + i < m ? uv : __builtin_inf();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p->detrend) {  // quantile.py:128-132: the series to rank is X minus its own least-squares line
            double a, b;
            trend_line<K>(u, m, lane, &a, &b);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * lane + i;
                u[i] = j < m ? u[i] - ((double)j * a + b) : u[i];  // trend.py:65,83
            }
            if (lane == 0 && cell_ok) {
                p->trend_u[2 * seg] = a;
                p->trend_u[2 * seg + 1] = b;
            }
        }
        wave_fence();
        {
            double s[K];
#pragma unroll
            for (int i = 0; i < K; ++i) s[i] = u[i];
            sort_segment<K>(s, row, m, lane);  // self ECDF: np.sort(u) (quantile.py:462 via 505-521)
        }
        // rank = (#sorted <= u) - 1: np.interp's exact-hit index = max rank among ties (quantile.py:488).
        // Branch-free binary search (len -> len - len/2 per step, the same wave-uniform stride for every lane),
        // CH independent chains at a time; positions are kept as LDS byte addresses (add, compare, select per
        // step).  A stride that is a multiple of 16 doubles would put the probes of all lanes on one or two
        // banks (the 2^k candidates of step k are whole strides apart): such strides are shortened by one.
#pragma unroll
        for (int i = 0; i < NR; ++i) rank2[i] = 0u;
        const unsigned rowb = lds_addr(row);
#pragma unroll
        for (int cbeg = 0; cbeg < K; cbeg += CH) {
            unsigned pb[CH];  // byte address of sorted[base - 1]
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) pb[ii] = rowb - 8u;
#pragma unroll 1
            for (int len = m; len > 1;) {
                int half = len >> 1;
                if ((half & 15) == 0) --half;
                len -= half;
                const unsigned h8 = (unsigned)half * 8u;
#pragma unroll
                for (int ii = 0; ii < CH; ++ii) {
                    const int i = cbeg + ii < K ? cbeg + ii : K - 1;
                    const unsigned t = pb[ii] + h8;
                    pb[ii] = lds_f64(t) <= u[i] ? t : pb[ii];
                }
            }
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                const int i = cbeg + ii;
                if (i < K) {
                    const int below = (int)(pb[ii] - rowb) >> 3;  // base - 1
                    const int r = below + (lds_f64(pb[ii] + 8u) <= u[i] ? 1 : 0);
                    const unsigned rk = (unsigned)(r > 0 ? r : 0);
                    rank2[i >> 1] |= (i & 1) ? (rk << 16) : rk;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cell_ok) {
            uint32_t* rk = p->ranks + (seg * p->slab_nr) * kWave + lane;
#pragma unroll
            for (int i = 0; i < NR; ++i) rk[i * kWave] = rank2[i];
        }
        return;
    }

    if (MODE == MODE_FIT && p->X != nullptr && n > 0) {
        // x climatology of the fitted state (bcsd.py:222); BcsdPrecipitation only validates X (bcsd.py:130-147)
        TileRegs<NR> xh;
        tile_issue<NR>(p->X, p->ld, p->ord_f + begf, n, c0, p->C, vec_f, xh);
        xc = tile_reduce_mean<NR>(xh, n, c0, p->C, scratch, p->status_fit, wave, lane);
        if (kTas && lane == 0 && cell_ok) p->x_climo[seg] = xc;
    }
    if (MODE == MODE_APPLY) {  // issued first: the loads fly while y is sorted
        if (kTas && cell_ok) xc = p->x_climo[seg];
        const uint32_t* rk = p->ranks + (seg * p->slab_nr) * kWave + lane;
#pragma unroll
        for (int i = 0; i < NR; ++i) rank2[i] = cell_ok ? rk[i * kWave] : 0u;
    }

    // ---- y: climatology + sorted segment in the wave's row ------------------------------------------
    double yc = 0.0;
    double icpt_fit = 0.0;  // detrend: intercept of the fitted segment's line
    if (!(MODE == MODE_APPLY && p->from_state)) {
        if (n > 0) {
            load_tile<NR>(p->y, p->ld, p->ord_f + begf, n, c0, p->C, vec_f, tile, RS, p->status_fit);
            __syncthreads();
            double v[K];
            load_blocked<K>(row, n, lane, 0.0, v);
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) s += v[i];
            yc = wave_sum(s) / (double)n;  // bcsd.py:223 / 138
            if (lane == 0 && cell_ok) {
                if (MODE == MODE_FIT) p->y_climo[seg] = yc;
                if (!kTas && p->return_anoms && yc <= 0.0) atomicOr(&p->status_fit[c], SDI_BAD_CLIMO);  // bcsd.py:140-141
            }
            if (p->detrend) {  // quantile.py:95-98: the CDF is fitted on y minus its least-squares line
                double a;
                trend_line<K>(v, n, lane, &a, &icpt_fit);
#pragma unroll
                for (int i = 0; i < K; ++i) v[i] = v[i] - ((double)(K * lane + i) * a + icpt_fit);  // trend.py:65,83
                if (MODE == MODE_FIT && lane == 0 && cell_ok) {
                    p->y_trend[2 * seg] = a;
                    p->y_trend[2 * seg + 1] = icpt_fit;
                }
            }
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = K * lane + i < n ? v[i] : __builtin_inf();
            wave_fence();
            sort_segment<K>(v, row, n, lane);  // quantile.py:462 np.sort
            if (MODE == MODE_FIT && cell_ok) {
                double* dst = p->ys + c * p->Tf + begf;
                for (int i = lane; i < n; i += kWave) dst[i] = row[i];
            }
        }
    } else {
        if (cell_ok) {
            yc = p->y_climo[seg];
            if (p->detrend) icpt_fit = p->y_trend[2 * seg + 1];
            const double* src = p->ys + c * p->Tf + begf;
            for (int i = lane; i < n; i += kWave) row[i] = src[i];
        }
        wave_fence();
    }
    if (MODE != MODE_APPLY) return;

    // The x_fut tile is read a second time (its RANK twin fetched it moments ago: L2 / Infinity Cache) to
    // recompute the rolling mean; the loads are issued here and fly during the lookups.
    TileRegs<NR> xf2;
    if (kTas) tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf2);

    // ---- map ranks through the fitted inverse CDF (quantile.py:523-545) ------------------------------
    double q[K];
    if (IDENT) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int r = (int)((i & 1) ? (rank2[i >> 1] >> 16) : (rank2[i >> 1] & 0xffffu));
            q[i] = row[r];
        }
    } else {
        double slo = 0.0, ilo = 0.0, shi = 0.0, ihi = 0.0;
        if (m > n && n > 0) {  // tails are reachable only when the predict segment is longer
            const int e = n < p->n_endpoints ? n : p->n_endpoints;
            const double dn = pp_denom(n);
            ols_line(row, 0, e, dn, &slo, &ilo);
            ols_line(row, n - e, e, dn, &shi, &ihi);
        }
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        const int32_t* qi = p->qidx + begp;
        const double* qv = p->qval + begp;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int r = (int)((i & 1) ? (rank2[i >> 1] >> 16) : (rank2[i >> 1] & 0xffffu));
            const int idx = qi[r];
            const double w = qv[r];
            double t;
            if (idx >= 0) {
                const double y0 = row[idx];
                const double y1 = row[idx + 1 < n ? idx + 1 : idx];
                t = w == 0.0 ? y0 : y0 + w * (y1 - y0);
            } else if (idx == -1) {
                t = w * slo + ilo;
            } else if (idx == -2) {
                t = w * shi + ihi;
            } else {
                t = nan;
            }
            q[i] = t;
            if ((i + 1) % CH == 0) __builtin_amdgcn_sched_barrier(0);  // keep later samples' loads from piling up
        }
    }

    if (p->detrend && cell_ok) {  // quantile.py:140-145: the predict line comes back, re-based on the fitted intercept
        const double a = p->trend_u[2 * seg], b = p->trend_u[2 * seg + 1];
        const double rebase = b - icpt_fit;
#pragma unroll
        for (int i = 0; i < K; ++i) q[i] = (q[i] + ((double)(K * lane + i) * a + b)) - rebase;
    }

    // ---- restore the climate-trend shift (bcsd.py:263-267) / ratio anomalies (bcsd.py:170-185) ------
    if (kTas) {
        __syncthreads();  // all lookups done: rows are free again
        tile_commit<NR>(xf2, m, c0, p->C, tile + kPadFront, RS, p->status_p);
        zero_pads(row, m, lane, CH + 4);
        __syncthreads();
#pragma unroll
        for (int cbeg = 0; cbeg < K; cbeg += CH) {
            double mean[CH], xv[CH];
            rolling_from_lds<CH>(row, K * lane + cbeg, m, rcp, mean, xv);
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                const int i = cbeg + ii;
                if (i < K) {
                    double res = (mean[ii] - xc) + q[i];  // bcsd.py:253,263
                    if (p->return_anoms) res = res - yc;   // bcsd.py:266-267
                    q[i] = res;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) q[i] = p->return_anoms ? q[i] / yc : q[i];  // bcsd.py:170-185
    }
    wave_fence();  // the wave's own row is rewritten in time order
    {
        const int base = K * lane;
#pragma unroll
        for (int i = 0; i < K; ++i) row[base + i < m ? base + i : m] = q[i];
    }
    __syncthreads();
    const bool vec_o = (p->ld_out % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->out) & 15) == 0);
    store_tile(p->out, p->ld_out, p->ord_p + begp, m, c0, p->C, vec_o, tile, RS);
}

template <int K, int MODE, int OCC, int KIND, bool IDENT>
__global__ void __launch_bounds__(kThreads, OCC) bcsd_rs_kernel(const Params) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();  // Params is the only kernel argument
    fill_rcp_table(reinterpret_cast<double*>(smem_raw) + 64);
    if (p->use_worklist) {
        // segments handed back by the fused kernel: a fixed grid walks the list
        int count = *p->work_count;
        if (count > p->work_cap) count = p->work_cap;
#pragma unroll 1
        for (int i = (int)blockIdx.x; i < count; i += (int)gridDim.x) {
            const int64_t item = p->worklist[i];
            segment_body<K, MODE, KIND, IDENT>(p, item / p->G, (int)(item % p->G), smem_raw);
            __syncthreads();  // the tile is reused by the next item
        }
        return;
    }
    int64_t tile_id;
    int g;
    xcd_tile_of_block(blockIdx.x, p->ntiles, &tile_id, &g);
    if (p->gmask != 0ull) g = nth_set_bit(p->gmask, g);  // this launch serves a subset of the groups
    if (tile_id >= p->ntiles || g < 0 || g >= p->G) return;
    segment_body<K, MODE, KIND, IDENT>(p, tile_id, g, smem_raw);
}

template <int K, int MODE, int OCC, int KIND, bool IDENT>
int launch_koki(sd_ctx* ctx, const Params& p, const char* name) {
    const size_t lds = ((size_t)kW * p.RS + sdw::kHeadDoubles) * sizeof(double);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_rs_kernel<K, MODE, OCC, KIND, IDENT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t tx = (p.ntiles + 7) / 8;
    int64_t nblocks = 8 * tx * (p.gmask ? __builtin_popcountll(p.gmask) : p.G);
    if (p.use_worklist) nblocks = 2 * (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256);
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "grid too large");
    SD_LAUNCH(ctx, name, (bcsd_rs_kernel<K, MODE, OCC, KIND, IDENT>), dim3((unsigned)nblocks), dim3(kThreads), lds, p);
    return SD_OK;
}

template <int K, int MODE, int OCC, int KIND>
int launch_kok(sd_ctx* ctx, const Params& p, const char* name) {
    if constexpr (MODE == MODE_APPLY) {  // IDENT only changes MODE_APPLY code
        if (p.identity) return launch_koki<K, MODE, OCC, KIND, true>(ctx, p, name);
    }
    return launch_koki<K, MODE, OCC, KIND, false>(ctx, p, name);
}

template <int K, int MODE>
int launch_k(sd_ctx* ctx, const Params& p, const char* name) {
    constexpr int OCC = K > 21 ? 2 : 4;
    return p.kind == SD_BCSD_TAS ? launch_kok<K, MODE, OCC, SD_BCSD_TAS>(ctx, p, name)
                                 : launch_kok<K, MODE, OCC, SD_BCSD_PR>(ctx, p, name);
}

template <int MODE>
int launch_mode(sd_ctx* ctx, const Params& p, int nmax, const char* name) {
    if (nmax <= 64 * 5) return launch_k<5, MODE>(ctx, p, name);
    if (nmax <= 64 * 13) return launch_k<13, MODE>(ctx, p, name);
    if (nmax <= 64 * 19) return launch_k<19, MODE>(ctx, p, name);
    if (nmax <= 64 * 21) return launch_k<21, MODE>(ctx, p, name);
    if (nmax <= 64 * 33) return launch_k<33, MODE>(ctx, p, name);
    return sd_set_error(SD_ERR_UNSUPPORTED, "segment of %d samples exceeds the register-sort path", nmax);
}

}  // namespace sdrs

// Entry points used by sd_bcsd.hip ---------------------------------------------------------------
bool sd_bcsd_rs_supported(int nmax) { return nmax >= 1 && nmax <= 64 * 33; }

int sd_bcsd_rs_width(int nmax) { return nmax <= 64 * 5 ? 5 : nmax <= 64 * 13 ? 13 : nmax <= 64 * 19 ? 19 : nmax <= 64 * 21 ? 21 : 33; }  // as in launch_mode

int sd_bcsd_rs_row_stride(int nmax) {
    const int K = sd_bcsd_rs_width(nmax);
    const int CH = K >= 14 ? (K + 2) / 3 : K;
    int rs = (nmax + K - 1) / K * K + 1;  // the sort stores the pads of the last run; one readable slot past the end
    const int roll = sdw::kPadFront + nmax + CH + 4;  // time-ordered segment with zero pads for the rolling windows
    if (rs < roll) rs = roll;
    while (rs % 4 != 2) ++rs;  // cell rows land 8 or 24 banks apart: conflict-free transposing stores
    return rs;
}

void sd_bcsd_rs_handoff_bytes(int nmax, int64_t C, int G, size_t* rank_bytes, size_t* shift_bytes) {
    const size_t K = (size_t)sd_bcsd_rs_width(nmax), segs = (size_t)C * (size_t)G;
    *rank_bytes = segs * ((K + 1) / 2) * 64 * sizeof(uint32_t);
    *shift_bytes = segs * K * 64 * sizeof(double);
}

static int rs_launch_one(sd_ctx* ctx, int mode, const sdrs::Params& p, int nmax) {
    switch (mode) {
        case sdrs::MODE_FIT: return sdrs::launch_mode<sdrs::MODE_FIT>(ctx, p, nmax, "bcsd_rs_fit_kernel");
        case sdrs::MODE_RANK: return sdrs::launch_mode<sdrs::MODE_RANK>(ctx, p, nmax, "bcsd_rs_rank_kernel");
        case sdrs::MODE_APPLY: return sdrs::launch_mode<sdrs::MODE_APPLY>(ctx, p, nmax, "bcsd_rs_apply_kernel");
        default: return sd_set_error(SD_ERR_INVALID, "unknown register-sort mode %d", mode);
    }
}

// The groups of a call that needs the 21-wide kernels but has groups fitting 19 samples per lane (30-day months of a
// daily series) are split over two launches: narrow = groups for the 19-wide kernels (0 when no split applies).
void sd_bcsd_rs_width_split(int nmax, int G, const int* group_len, unsigned long long* wide, unsigned long long* narrow) {
    *wide = *narrow = 0ull;
    if (sd_bcsd_rs_width(nmax) != 21 || group_len == nullptr || G > 64) return;
    const char* e = sd_dev_env("SD_RS_SPLIT");  // "0": one launch of the widest kernels for every group (A/B measurements)
    if (e && e[0] == '0') return;
    unsigned long long nw = 0ull, wd = 0ull;
    for (int g = 0; g < G; ++g) (group_len[g] <= 64 * 19 ? nw : wd) |= 1ull << g;
    if (nw != 0ull && wd != 0ull) {
        *wide = wd;
        *narrow = nw;
    }
}

int sd_bcsd_rs_launch(sd_ctx* ctx, int mode, const sdrs::Params& p, int nmax, const int* group_len) {
    sdrs::Params q = p;
    if (q.n_endpoints <= 0) q.n_endpoints = 10;
    const int kmax = sd_bcsd_rs_width(nmax);
    q.gmask = 0ull;
    q.slab_nr = (kmax + 1) / 2;
    q.slab_k = kmax;
    unsigned long long wide = 0ull, narrow = 0ull;
    if (!p.use_worklist) sd_bcsd_rs_width_split(nmax, p.G, group_len, &wide, &narrow);
    if (narrow != 0ull) {
        q.gmask = wide;
        SD_TRY(rs_launch_one(ctx, mode, q, nmax));
        q.gmask = narrow;
        return rs_launch_one(ctx, mode, q, 64 * 19);
    }
    return rs_launch_one(ctx, mode, q, nmax);
}
