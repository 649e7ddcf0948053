// BcsdTemperature fit + predict of one (cell, month) segment in ONE workgroup pass -- the headline kernel.
//
// Same decomposition as sd_bcsd_rs.hip (one 64-lane wave per segment, 8 adjacent cells per 512-thread workgroup,
// K consecutive samples per lane, two workgroups per CU), but the rank of a predict sample is not searched for:
// the sort itself carries it.  Reference semantics: bcsd.py:197-269, quantile.py:81-147, 438-545.
//
//   x side   x_hist rows are streamed and reduced to x_climo (bcsd.py:222); the x_fut tile is transposed into one
//            zero-padded LDS row per cell; the 9-sample rolling mean gives the shift (bcsd.py:247-253) and the shifted
//            series u = x - shift (bcsd.py:256).
//   tags     the low 16 mantissa bits of every u are replaced by 8 * (its position in the segment).  Truncation is
//            monotone, so the tagged values sort exactly like the u's unless two of them agree in the upper 48 bits.
//            After the wave's merge sort (v_min_f64 / v_max_f64 move the tag along with the key at no cost) the lane
//            holding sorted positions p .. p+K-1 reads the time positions of the samples with ranks p .. p+K-1 off
//            the tags: self-ECDF rank (quantile.py:505-521, np.interp's "last xp <= x" rule, quantile.py:488) without a
//            search.  Pads are distinct huge finite values (never +inf: a tagged inf would be a NaN).
//   ties     two samples that agree in the upper 48 bits (exact ties included) make the ranks ambiguous: the lanes
//            compare neighbouring sorted keys, and a workgroup that sees such a pair in any of its segments appends
//            its (tile, group) to the work list and stops; sd_bcsd.hip then runs the RANK / APPLY kernels (explicit
//            search, max rank among ties) over the list.  Random float64 data: ~1e-5 of the segments.
//   y side   y_obs rows are loaded / transposed, reduced to y_climo (bcsd.py:223) and sorted in the same LDS rows
//            (np.sort, quantile.py:462) -- or the sorted segment comes from a fitted state.
//   map      rank r -> fitted inverse CDF (quantile.py:523-545): with equal fit / predict group lengths this is the
//            r-th sorted observation, which the lane already holds in registers (it owns sorted positions p .. p+K-1
//            of *both* sorts); otherwise the per-rank (index, weight) table with 10-point OLS tails.  The value is
//            scattered to the sample's time position in the row, every lane reads back its K consecutive samples,
//            adds the shift (bcsd.py:263), removes y_climo (bcsd.py:266-267) and the tile goes out transposed.
//   shift    SLAB = false: the x_fut tile is read a second time (L2 / Infinity Cache) and the rolling mean recomputed;
//            SLAB = true: the x side parks the shift in a workspace slab in lane layout (coalesced 512-byte wave
//            transactions) and the map step reads it back.
// HBM traffic: 3 reads + 1 write per sample (+ the re-read or the slab); no rank slab, no second kernel.
#include "sd_bcsd_rs.h"
#include "sd_wave.h"

namespace sdfz {

using namespace sdw;
using sdrs::Params;

typedef const Params __attribute__((address_space(4)))* ParamsPtr;

constexpr unsigned kTagMask = 0xffffu;  // low 16 bits of the mantissa carry 8 * position (positions < 64 * 33)
constexpr int kPadHi = 0x7fe00000;      // pads: 2^1023 * (1 + position * 2^-36), tagged like data

__device__ __forceinline__ double from_words(unsigned lo, int hi) {
    return __hiloint2double(hi, (int)lo);
}

template <int K, bool IDENT, bool SLAB>
__global__ void __launch_bounds__(kThreads, 4) bcsd_fz_kernel(const Params) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();  // Params is the only kernel argument
    constexpr int CH = Chunk<K>::CH;
    constexpr int NR = (K + 1) / 2;
#ifdef SD_DEV
    const int abl = p->dev_flags;  // SD_FZ_ABLATE (low byte): 1 no u merge rounds, 2 no y merge rounds, 4 no x_hist, 8 no shift restore,
                                   // 16 no store, 32 no y load, 64 no x_fut load (timing only: results are wrong)
#else
    constexpr int abl = 0;
#endif
    double* const scratch = reinterpret_cast<double*>(smem_raw);  // 64 doubles (column-sum exchange)
    double* const rcp = scratch + 64;                             // 16 doubles: correctly rounded 1/c, c = 1..9
    int* const bad_cell = reinterpret_cast<int*>(rcp + 16);       // 8 ints: cell of the tile saw a non-finite sample
    double* const tile = scratch + kHeadDoubles;
    const int RS = p->RS;
    fill_rcp_table(rcp);
    if (threadIdx.x >= 32 && threadIdx.x < 32 + kW) bad_cell[threadIdx.x - 32] = 0;

#ifdef SD_DEV
    if ((abl >> 8) != 0 && blockIdx.x < 512u) {
        // SD_FZ_ABLATE bits 8.. = T: the first generation of workgroups starts spread over ~T microseconds (hash of the
        // workgroup id) instead of in lockstep; later generations inherit the offsets
        const unsigned h = (blockIdx.x * 2654435761u) >> 20;  // 12 bits
        const unsigned n = (h * (unsigned)(abl >> 8)) >> 12;  // 0 .. T-1 "microseconds"
        for (unsigned i = 0; i < 2u * n; ++i) __builtin_amdgcn_s_sleep(19);  // ~1216 clocks ~ 0.5 us
    }
#endif
    int64_t tile_id;
    int g;
    xcd_tile_of_block(blockIdx.x, p->ntiles, &tile_id, &g);
    if (p->gmask != 0ull) g = nth_set_bit(p->gmask, g);  // this launch serves a subset of the groups
    if (tile_id >= p->ntiles || g < 0 || g >= p->G) return;

    const int64_t c0 = tile_id * kW;
    const int wave = __builtin_amdgcn_readfirstlane(tid_now() / kWave);  // wave-uniform: what derives from it stays scalar
    // The lane id is re-read behind an opaque barrier in every phase (lane_now): otherwise the compiler keeps dozens of
    // per-sample index expressions (min(K * lane + i, m), ...) alive from the first phase to the last and spills them.
#define SD_LANE() const int lane = tid_now() % kWave
    const int64_t c = c0 + wave;
    const bool cell_ok = c < p->C;
    double* const row = tile + wave * RS;
    const int64_t seg = c * p->G + g;
    const int begf = p->off_f[g];
    const int n = p->off_f[g + 1] - begf;
    const int begp = p->off_p[g];
    const int m = p->off_p[g + 1] - begp;
    if (m == 0) return;
    const bool vec_f = (p->ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->y) & 15) == 0) &&
                       (p->X == nullptr || (reinterpret_cast<uintptr_t>(p->X) & 15) == 0);
    const bool vec_p = (p->ld_p % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->Xp) & 15) == 0);
    // masked cells (core.py:35-37) and cells already known to hold non-finite samples are overwritten with NaN
    // afterwards: whatever their rows contain must not send the tile to the work list
    const bool cell_live = cell_ok && p->status_fit[cell_ok ? c : 0] == 0;
    __syncthreads();  // bad_cell zeroed before the commits below may set it

    // ---- x climatology (bcsd.py:222) ----------------------------------------------------------------
    // The x_hist rows and then the x_fut tile are requested back to back (loads return in order), so the
    // column sums are reduced while the tile is still in flight: one exposed memory latency instead of two.
    double xc = 0.0;
    {
        SD_LANE();
        TileRegs<NR> xf;
        if (p->from_state) {
            if (cell_ok) xc = p->x_climo[seg];
            tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf);
        } else if (n > 0 && !(abl & 4)) {
            TileRegs<NR> xh;
            tile_issue<NR>(p->X, p->ld, p->ord_f + begf, n, c0, p->C, vec_f, xh);
            if (!(abl & 64)) tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf);
            xc = tile_reduce_mean<NR>(xh, n, c0, p->C, scratch, p->status_fit, wave, lane, bad_cell);
        } else if (!(abl & 64)) {
            tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf);
        }
        if (!(abl & 64)) tile_commit<NR>(xf, m, c0, p->C, tile + kPadFront, RS, p->status_p, bad_cell);
        zero_pads(row, m, lane, CH + 4);
    }
    __syncthreads();

    // ---- shifted series, tagged with the time position ------------------------------------------------
    double* const slab0 = SLAB ? p->shift + (seg * p->slab_k) * kWave : nullptr;
    const int np = (m + K - 1) / K * K;
    unsigned pos2[NR];  // two 16-bit tags (8 * time position) per register
    bool redo = false;
    // the y_obs tile: K <= 19 requests it right behind the sort of u (its latency then covers the tag bookkeeping and the
    // workgroup's vote: -2.4 % on those launches); at K = 21 the 44 registers in flight there cost more in spills than they hide
    TileRegs<NR> yt;
    {
        SD_LANE();
        double* const slab = slab0 + lane;
        double u[K];
#pragma unroll
        for (int cbeg = 0; cbeg < K; cbeg += CH) {
            double mean[CH], xv[CH];
            rolling_from_lds<CH>(row, K * lane + cbeg, m, rcp, mean, xv);
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                const int i = cbeg + ii;
                if (i < K) {
                    const double shift = mean[ii] - xc;            // bcsd.py:253
                    const double uv = (xv[ii] - shift) + 0.0;       // bcsd.py:256; -0.0 -> +0.0 (they tie in np.sort / np.interp)
                    if (SLAB && cell_ok) slab[i * kWave] = shift;
                    const int j = K * lane + i;
                    const unsigned tag = (unsigned)j * 8u;
                    const unsigned lo = ((unsigned)__double2loint(uv) & ~kTagMask) | tag;
                    const bool in = j < m;
                    u[i] = from_words(in ? lo : (((unsigned)j << 16) | tag), in ? __double2hiint(uv) : kPadHi);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wave_fence();
        sort_segment<K, true>(u, row, m, lane, !(abl & 1));  // u[] <- the sorted values of the positions this lane owns
        if (K <= 19 && !p->from_state && n > 0 && !(abl & 32)) tile_issue<NR>(p->y, p->ld, p->ord_f + begf, n, c0, p->C, vec_f, yt);
        // ---- ranks off the tags; near-tie detection ----------------------------------------------------
        const bool owner = K * lane < np;  // this lane owns sorted positions K*lane .. K*lane + K - 1
        long long key[K];  // upper 48 bits
#pragma unroll
        for (int i = 0; i < K; ++i) key[i] = __double_as_longlong(u[i]) & ~(long long)kTagMask;
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) redo |= key[i] == key[i + 1];
        const long long knext = __shfl_down(key[0], 1, kWave);
        redo |= (K * (lane + 1) < np) && key[K - 1] == knext;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned e = (unsigned)__double2loint(u[2 * i]) & kTagMask;
            const unsigned o = 2 * i + 1 < K ? (unsigned)__double2loint(u[2 * i + 1]) << 16 : 0u;
            pos2[i] = e | o;
        }
        // data at or above the pad range would sort behind pads: hand the segment back as well
        redo |= __double2hiint(row[m - 1]) >= kPadHi;
        redo = redo && owner && cell_live && bad_cell[wave] == 0 && (abl & 255) == 0;
    }
    // every wave is done with its row; a workgroup with an ambiguous segment leaves the (tile, group) to RANK / APPLY
    if (__syncthreads_or(redo ? 1 : 0)) {
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
        }
        return;
    }

    // ---- y: climatology + sorted segment in the wave's row ------------------------------------------
    double yc = 0.0;
    double t[K];  // IDENT: the sorted observations of the positions this lane owns (rank r <-> r-th sorted observation)
    if (!p->from_state) {
        if (n > 0) {
            SD_LANE();
            if (K > 19 && !(abl & 32)) tile_issue<NR>(p->y, p->ld, p->ord_f + begf, n, c0, p->C, vec_f, yt);
            if (!(abl & 32)) tile_commit<NR>(yt, n, c0, p->C, tile, RS, p->status_fit);
            __syncthreads();
            double v[K];
            load_blocked<K>(row, n, lane, 0.0, v);
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) s += v[i];
            yc = wave_sum(s) / (double)n;  // bcsd.py:223
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = K * lane + i < n ? v[i] : __builtin_inf();
            wave_fence();
            sort_segment<K, IDENT>(v, row, n, lane, !(abl & 2));  // quantile.py:462 np.sort
            if (IDENT) {
#pragma unroll
                for (int i = 0; i < K; ++i) t[i] = v[i];
            }
        }
    } else {
        SD_LANE();
        if (cell_ok) {
            yc = p->y_climo[seg];
            const double* src = p->ys + c * p->Tf + begf;
            for (int i = lane; i < n; i += kWave) row[i] = src[i];
        }
        wave_fence();
        if (IDENT) {
            const double* srow = row + (K * lane < np ? K * lane : 0);
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = srow[i];
        }
    }

    TileRegs<NR> xf2;  // SLAB = false: second read of the x_fut tile, in flight during the map step
    if (!SLAB && !(abl & 8)) tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0, p->C, vec_p, xf2);

    // ---- map ranks through the fitted inverse CDF (quantile.py:523-545), scatter to time positions -----
    {
        SD_LANE();
        const bool owner = K * lane < np;
        if (!IDENT) {
            double slo = 0.0, ilo = 0.0, shi = 0.0, ihi = 0.0;
            if (m > n && n > 0) {  // tails are reachable only when the predict segment is longer
                const int e = n < 10 ? n : 10;
                const double dn = pp_denom(n);
                ols_line(row, 0, e, dn, &slo, &ilo);
                ols_line(row, n - e, e, dn, &shi, &ihi);
            }
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
            const int r0 = owner ? K * lane : 0;
            const int32_t* qi = p->qidx + begp + r0;
            const double* qv = p->qval + begp + r0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const bool in = r0 + i < m;
                const int idx = in ? qi[i] : -3;
                const double w = in ? qv[i] : 0.0;
                double v;
                if (idx >= 0) {
                    const double y0 = row[idx];
                    const double y1 = row[idx + 1 < n ? idx + 1 : idx];
                    v = w == 0.0 ? y0 : y0 + w * (y1 - y0);
                } else if (idx == -1) {
                    v = w * slo + ilo;
                } else if (idx == -2) {
                    v = w * shi + ihi;
                } else {
                    v = nan;
                }
                t[i] = v;
                if ((i + 1) % CH == 0) __builtin_amdgcn_sched_barrier(0);  // keep later samples' loads from piling up
            }
        }
        wave_fence();  // every lane has read what it needs of the sorted row
        if (owner) {
            const unsigned rowb = lds_addr(row);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const unsigned tag = (i & 1) ? (pos2[i >> 1] >> 16) : (pos2[i >> 1] & kTagMask);
                lds_store_f64(rowb + tag, t[i]);
            }
        }
        wave_fence();
    }
    double q[K];
    {
        SD_LANE();
        const double* trow = row + (K * lane < np ? K * lane : 0);
#pragma unroll
        for (int i = 0; i < K; ++i) q[i] = trow[i];
    }

    // ---- restore the climate-trend shift (bcsd.py:263-267) --------------------------------------------
    if (SLAB) {
        if (cell_ok) {
            SD_LANE();
            const double* const slab = slab0 + lane;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                double res = slab[i * kWave] + q[i];  // bcsd.py:253,263
                if (p->return_anoms) res = res - yc;   // bcsd.py:266-267
                q[i] = res;
            }
        }
        wave_fence();
    } else if (!(abl & 8)) {
        __syncthreads();  // all rows read: free again
        tile_commit<NR>(xf2, m, c0, p->C, tile + kPadFront, RS, p->status_p);
        SD_LANE();
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
        wave_fence();  // the wave's own row is rewritten in time order
    }
    {
        SD_LANE();
        const int base = K * lane;
#pragma unroll
        for (int i = 0; i < K; ++i) row[base + i < m ? base + i : m] = q[i];
    }
    __syncthreads();
    const bool vec_o = (p->ld_out % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->out) & 15) == 0);
    if (!(abl & 16)) store_tile(p->out, p->ld_out, p->ord_p + begp, m, c0, p->C, vec_o, tile, RS);
#undef SD_LANE
}

template <int K, bool IDENT, bool SLAB>
int launch_kis(sd_ctx* ctx, const Params& p) {
    const size_t lds = ((size_t)kW * p.RS + kHeadDoubles) * sizeof(double);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fz_kernel<K, IDENT, SLAB>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t tx = (p.ntiles + 7) / 8;
    const int64_t nblocks = 8 * tx * (p.gmask ? __builtin_popcountll(p.gmask) : p.G);
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "grid too large");
    SD_LAUNCH(ctx, "bcsd_fz_kernel", (bcsd_fz_kernel<K, IDENT, SLAB>), dim3((unsigned)nblocks), dim3(kThreads), lds, p);
    return SD_OK;
}

template <int K>
int launch_k(sd_ctx* ctx, const Params& p) {
    if (p.shift != nullptr)
        return p.identity ? launch_kis<K, true, true>(ctx, p) : launch_kis<K, false, true>(ctx, p);
    return p.identity ? launch_kis<K, true, false>(ctx, p) : launch_kis<K, false, false>(ctx, p);
}

int launch_width(sd_ctx* ctx, const Params& p, int nmax) {
    if (nmax <= 64 * 5) return launch_k<5>(ctx, p);
    if (nmax <= 64 * 13) return launch_k<13>(ctx, p);
    if (nmax <= 64 * 19) return launch_k<19>(ctx, p);
    if (nmax <= 64 * 21) return launch_k<21>(ctx, p);
    return sd_set_error(SD_ERR_UNSUPPORTED, "segment of %d samples exceeds the fused register-sort path", nmax);
}

}  // namespace sdfz

int sd_bcsd_rs_width(int nmax);
void sd_bcsd_rs_width_split(int nmax, int G, const int* group_len, unsigned long long* wide, unsigned long long* narrow);

bool sd_bcsd_fz_supported(int nmax) { return nmax >= 1 && nmax <= 64 * 21; }

int sd_bcsd_fz_launch(sd_ctx* ctx, const sdrs::Params& p, int nmax, const int* group_len) {
    sdrs::Params q = p;
    const int kmax = sd_bcsd_rs_width(nmax);
    q.gmask = 0ull;
    q.slab_nr = (kmax + 1) / 2;
    q.slab_k = kmax;
    q.use_worklist = 0;
    unsigned long long wide = 0ull, narrow = 0ull;
    sd_bcsd_rs_width_split(nmax, p.G, group_len, &wide, &narrow);
    if (narrow != 0ull) {
        q.gmask = wide;
        SD_TRY(sdfz::launch_width(ctx, q, nmax));
        q.gmask = narrow;
        return sdfz::launch_width(ctx, q, 64 * 19);
    }
    return sdfz::launch_width(ctx, q, nmax);
}
