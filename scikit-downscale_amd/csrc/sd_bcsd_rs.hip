// BCSD quantile mapping, fast path: one 64-lane wave per (cell, month) segment, 8 adjacent cells per
// 512-thread workgroup, two workgroups per CU (both the 160 KB LDS -- 16 rows of 10 KB -- and the 128-VGPR
// budget allow 16 waves per CU).  Three kernels share one body (segment_body, MODE template):
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
//   MODE_FIT    the y side alone, writing the fitted state.  (MODE_BOTH = RANK then APPLY in one workgroup pass:
//               optional, see sd_bcsd.hip.)
// Sort (sd_sortnet.h): each lane sorts its K registers with a Batcher odd-even merge network
// (v_min_f64/v_max_f64), writes the run to its LDS row, then 6 merge rounds double the run length; in every
// round a lane finds its co-rank by binary search (merge path), loads its windows of the two runs and merges
// them in registers with a pruned bitonic merger before the wave writes them back in place (LDS requests of one
// wave are served in order, so a wavefront-scope fence replaces barriers inside the sort).  K is odd so that
// lane-strided LDS accesses are bank-conflict free.
//
// Workgroup ids are mapped XCD-aware (workgroup b runs on XCD b % 8): every XCD owns a contiguous range of cell
// tiles, so the two 64-byte halves of a 128-byte line are fetched by workgroups sharing an L2.
#include <cstdlib>

#include "sd_bcsd_rs.h"
#include "sd_sortnet.h"

namespace sdrs {

using namespace sdsort;

constexpr int kWave = 64;
constexpr int kW = 8;          // cells per workgroup
constexpr int kThreads = 512;  // 8 waves
constexpr int kRowsPerPass = kThreads / 4;  // 4 lanes (16 B each) cover the 8 cells of one row

// The thread index behind an opaque barrier (see SD_DERIVE in segment_body).
__device__ __forceinline__ int tid_now() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
// The kernel reads its Params straight from the kernarg segment (scalar loads on demand, re-loadable).
typedef const Params __attribute__((address_space(4)))* ParamsPtr;

__device__ __forceinline__ bool finite64(double v) {
    return (__double_as_longlong(v) & 0x7ff0000000000000ll) != 0x7ff0000000000000ll;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
// Lanes of one wave exchange data through LDS inside the sort.  The hardware serves a wave's LDS
// requests in order; for the compiler the exchange needs a wavefront-scope fence plus the wave barrier.
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- wave-level merge sort of row[0..n): runs of K per lane -> fully sorted, in place ------------
// Round r merges pairs of runs of length K << r.  Every lane owns K consecutive output positions of
// its pair: it finds its co-rank (merge path) by binary search, loads the matching windows of A and B
// (exactly one LDS read per element, all independent), merges them in registers and the wave writes
// the K outputs back in place.  LDS requests of one wave are served in order: no barrier needed.
template <int K>
__device__ __forceinline__ void merge_rounds(double* row, int np, int lane) {
    // np = number of slots being sorted, a multiple of K: the +inf pads that fill the last lane's run are
    // ordinary elements (they sort to the end), so every participating lane merges exactly K outputs and
    // no per-element validity test is needed; lanes past np sit out (one divergent branch per round).
    constexpr MergeNet<K> net{};
#pragma unroll 1
    for (int r = 0; r < 6; ++r) {
        const int L = K << r;
        if (L >= np) break;  // wave-uniform: a single run left
        const int gl = lane & ((2 << r) - 1);  // lane within its merge group
        const int base = (lane - gl) * K;
        const int a0 = base < np ? base : np;
        const int a1 = base + L < np ? base + L : np;
        const int b1 = base + 2 * L < np ? base + 2 * L : np;
        const int LA = a1 - a0, LB = b1 - a1;
        const int d0 = gl * K;
        const bool busy = d0 < LA + LB;  // this lane owns K outputs of the pair (LA + LB is a multiple of K)
        const int d = busy ? d0 : LA + LB;
        // co-rank: smallest i with A[i] > B[d-1-i]; ties go to A (stable merge)
        int lo = d - LB > 0 ? d - LB : 0, hi = d < LA ? d : LA;
        const int nsteps = r + ceil_log2(K + 1);
        const double* pa0 = row + a0;
        const double* pb0 = row + a1 + d - 1;
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const int mid = (lo + hi) >> 1;          // lo == hi (finished lane): reads stay inside the row, updates are no-ops
            const bool le = (pa0[mid] <= pb0[-mid]) && (lo < hi);
            lo = le ? mid + 1 : lo;
            hi = le ? hi : mid;
        }
        const int inext = __shfl_down(lo, 1, kWave);
        const int ihi = (d + K >= LA + LB) ? LA : inext;  // co-rank of the end of this lane's window
        const int acnt = ihi - lo;                         // elements taken from A; K - acnt from B
        double w[K];
        if (busy) {
            const double* pa = row + a0 + lo;                       // A window, ascending: pa[s], s < acnt
            const double* pq = row + a1 + (d - lo) + (K - acnt) - 1 + acnt;  // B window read backwards: pq[-s], s >= acnt
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const double* src = s < acnt ? pa : pq - 2 * s;     // (pq - 2s)[s] == pq[-s]
                w[s] = src[s];
                if (s % 7 == 6) __builtin_amdgcn_sched_barrier(0);  // issue the loads in batches
            }
#pragma unroll
            for (int c = 0; c < net.n; ++c) {
                const double mn = vmin(w[net.a[c]], w[net.b[c]]);
                const double mx = vmax(w[net.a[c]], w[net.b[c]]);
                w[net.a[c]] = mn;
                w[net.b[c]] = mx;
            }
        }
        wave_fence();
        if (busy) {
            double* dst = row + a0 + d;
#pragma unroll
            for (int s = 0; s < K; ++s) dst[s] = w[net.out[s]];
        }
        wave_fence();
    }
}

// sort the wave's segment: v[] = K consecutive samples per lane (pads = +inf), result in row[0..n); the
// row must have ceil(n / K) * K + 1 slots (the pads of the last run are stored and sorted like data).
template <int K>
__device__ __forceinline__ void sort_segment(double (&v)[K], double* row, int n, int lane) {
    sort_registers<K>(v);
    const int np = (n + K - 1) / K * K;
    if (K * lane < np) {
        double* dst = row + K * lane;
#pragma unroll
        for (int i = 0; i < K; ++i) dst[i] = v[i];
    }
    wave_fence();
    merge_rounds<K>(row, np, lane);
}

// ---- tile movement ------------------------------------------------------------------------------
// rows of one group for the 8 cells of the tile -> LDS rows (cell-major).  16-byte loads when possible.
// A thread owns rows rr, rr+128, ... (at most RPT of them); all of its loads are issued before the first
// use so the whole tile costs one memory latency, not one per batch.
// The two halves of a tile load can be separated (TileRegs): issue early, commit to LDS when the rows are free.
template <int RPT>
struct TileRegs {
    double v0[RPT], v1[RPT];
};

// address of row ti of a [rows, ld] field, cp = pointer to the tile's column in row 0: one v_mad_u64_u32 (the
// launcher guarantees 0 <= ti and 8 * ld < 2^32; the int64 product costs three quarter-rate multiplies per row)
__device__ __forceinline__ const double* row_of(const double* cp, int ti, int64_t ld) {
    const uint64_t off = (uint64_t)(uint32_t)ti * (uint64_t)(uint32_t)((uint32_t)ld * 8u);
    return reinterpret_cast<const double*>(reinterpret_cast<const char*>(cp) + off);
}
__device__ __forceinline__ double* row_of(double* cp, int ti, int64_t ld) {
    return const_cast<double*>(row_of(const_cast<const double*>(cp), ti, ld));
}

template <int RPT>
__device__ __forceinline__ void tile_issue(const double* __restrict__ src, int64_t ld, const int32_t* __restrict__ ord,
                                           int nrows, int64_t c0, int64_t C, bool vec_ok, TileRegs<RPT>& t, int rmask = -1) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    const bool full = vec_ok && c + 1 < C;
    int ti[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRowsPerPass;
        ti[k] = ord[r < nrows ? r : 0] & rmask;
    }
    if (full) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const double2 v = *reinterpret_cast<const double2*>(row_of(src + c, ti[k], ld));
            t.v0[k] = v.x;
            t.v1[k] = v.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const double* p = row_of(src + c, ti[k], ld);
            t.v0[k] = c < C ? p[0] : 0.0;
            t.v1[k] = c + 1 < C ? p[1] : 0.0;
        }
    }
}

template <int RPT>
__device__ __forceinline__ void tile_commit(const TileRegs<RPT>& t, int nrows, int64_t c0, int64_t C, double* tile, int RS,
                                            int32_t* status) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    double* d0 = tile + (2 * cp) * RS;
    double* d1 = d0 + RS;
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRowsPerPass;
        if (r < nrows) {
            bad0 |= !finite64(t.v0[k]);
            bad1 |= !finite64(t.v1[k]);
            d0[r] = t.v0[k];
            d1[r] = t.v1[k];
        }
    }
    if (bad0 && c < C) atomicOr(&status[c], SDI_NONFINITE);
    if (bad1 && c + 1 < C) atomicOr(&status[c + 1], SDI_NONFINITE);
}

template <int RPT>
__device__ __forceinline__ void load_tile(const double* __restrict__ src, int64_t ld, const int32_t* __restrict__ ord,
                                          int nrows, int64_t c0, int64_t C, bool vec_ok, double* tile, int RS,
                                          int32_t* status) {
    TileRegs<RPT> t;
    tile_issue<RPT>(src, ld, ord, nrows, c0, C, vec_ok, t);
    tile_commit<RPT>(t, nrows, c0, C, tile, RS, status);
}

__device__ __forceinline__ void store_tile(double* __restrict__ dst, int64_t ld, const int32_t* __restrict__ ord,
                                           int nrows, int64_t c0, int64_t C, bool vec_ok, const double* tile, int RS) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    const double* s0 = tile + (2 * cp) * RS;
    const double* s1 = s0 + RS;
    const bool full = vec_ok && c + 1 < C;
#pragma unroll 4
    for (int r = rr; r < nrows; r += kRowsPerPass) {
        double* p = row_of(dst + c, ord[r], ld);
        if (full) {
            *reinterpret_cast<double2*>(p) = make_double2(s0[r], s1[r]);
        } else {
            if (c < C) p[0] = s0[r];
            if (c + 1 < C) p[1] = s1[r];
        }
    }
}

// column means of one group's rows for the 8 cells from issued tile registers (x climatology; nothing stored)
template <int RPT>
__device__ __forceinline__ double tile_reduce_mean(const TileRegs<RPT>& t, int nrows, int64_t c0, int64_t C, double* scratch,
                                                   int32_t* status, int wave, int lane) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    double s0 = 0.0, s1 = 0.0;
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool in = rr + k * kRowsPerPass < nrows;
        bad0 |= in && !finite64(t.v0[k]);
        bad1 |= in && !finite64(t.v1[k]);
        s0 += in ? t.v0[k] : 0.0;
        s1 += in ? t.v1[k] : 0.0;
    }
    if (bad0 && c < C) atomicOr(&status[c], SDI_NONFINITE);
    if (bad1 && c + 1 < C) atomicOr(&status[c + 1], SDI_NONFINITE);
#pragma unroll
    for (int o = 4; o <= 32; o <<= 1) {  // lanes with equal (lane & 3) hold the same cell pair
        s0 += __shfl_xor(s0, o, kWave);
        s1 += __shfl_xor(s1, o, kWave);
    }
    if (lane < 4) {
        scratch[wave * kW + 2 * lane] = s0;
        scratch[wave * kW + 2 * lane + 1] = s1;
    }
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < kW; ++w) tot += scratch[w * kW + wave];  // wave <-> cell c0 + wave
    __syncthreads();
    return tot / (double)nrows;
}

template <int K>
__device__ __forceinline__ void load_blocked(const double* row, int cnt, int lane, double pad, double (&v)[K]) {
    const int base = K * lane;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = base + i;
        const double t = row[j < cnt ? j : 0];
        v[i] = j < cnt ? t : pad;
    }
}

// 9-sample centred rolling means (bcsd.py:247-250) for CH consecutive samples j0..j0+CH-1 of the wave's
// segment, which sits in its LDS row in time order *at offset 4 with zeros on both sides* (zero_pads): the
// CH+8 window values are plain reads at immediate offsets (lane stride K is odd: conflict-free), samples
// outside [0, m) contribute 0 and the divisor is the clipped window length.
constexpr int kPadFront = 4;
__device__ __forceinline__ void zero_pads(double* row, int m, int lane, int nback) {
    if (lane < kPadFront) row[lane] = 0.0;
    if (lane < nback) row[kPadFront + m + lane] = 0.0;
}
template <int CH>
__device__ __forceinline__ void rolling_from_lds(const double* row, int j0, int m, const double* rcp, double (&mean)[CH],
                                                 double (&centre)[CH]) {
    const double* win = row + (j0 < m ? j0 : 0);  // win[t] = sample j0 - 4 + t; lanes past the segment read in bounds
    double w[CH + 8];
#pragma unroll
    for (int t = 0; t < CH + 8; ++t) w[t] = win[t];
#pragma unroll
    for (int ii = 0; ii < CH; ++ii) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < 9; ++d) s += w[ii + d];
        const int j = j0 + ii;
        const int lo = j - 4 > 0 ? j - 4 : 0;
        const int hi = j + 5 < m ? j + 5 : m;
        const int c = hi - lo > 1 ? (hi - lo < 10 ? hi - lo : 9) : 1;
        const double cd = (double)c;
        const double rc = rcp[c];
        const double q = s * rc;
        mean[ii] = __builtin_fma(__builtin_fma(-cd, q, s), rc, q);  // correctly rounded s / c (Markstein step)
        centre[ii] = w[ii + 4];
    }
}

constexpr double kAlpha = 0.4, kBeta = 0.4;
__device__ __forceinline__ double pp_denom(int n) { return ((double)n + 1.0 - kAlpha) - kBeta; }
__device__ __forceinline__ double pp_at(int i, double denom) { return ((double)(i + 1) - kAlpha) / denom; }

__device__ void ols_line(const double* ysg, int first, int e, double denom, double* slope, double* icpt) {
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

// ------------------------------------------------------------------------------------------------
// Samples handled together in the rolling / search / lookup phases (bounded register pressure).
template <int K>
struct Chunk {
    static constexpr int CH = K >= 14 ? (K + 2) / 3 : K;
};

// LDS read at an absolute 32-bit LDS byte address (the search keeps positions as addresses)
typedef __attribute__((address_space(3))) const double lds_cdouble_t;
__device__ __forceinline__ double lds_f64(unsigned addr) { return *reinterpret_cast<lds_cdouble_t*>((uintptr_t)addr); }
__device__ __forceinline__ unsigned lds_addr(const void* generic_ptr_into_lds) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)generic_ptr_into_lds;
}

// OCC = waves per SIMD the register allocation is capped for: 4 -> 128 VGPRs (2 workgroups per CU),
// 2 -> 256 VGPRs (1 workgroup per CU).
// IDENT: every group has the same length in fit and predict, so the fitted inverse CDF evaluated at the
// Cunnane position of rank r is exactly the r-th sorted observation (np.interp exact hit): no table needed.
//
// Hand-off between MODE_RANK and MODE_APPLY (context workspace, one slab per (cell, group) segment, written
// and read with the same lane layout so every access is a fully coalesced 256/512-byte wave transaction):
//   ranks: [segment][(K+1)/2][64] u32 -- two 16-bit ranks per word, exactly the rank2[] registers
//   shift: [segment][K][64] f64       -- rolling mean - x_climo of every sample (TAS, optional: when absent
//                                        APPLY re-reads the x_fut tile and recomputes it)
// development (make trace -> lib/libsd_downscale_trace.so, run with SD_RS_TRACE=1): phase stamps (100 MHz wall
// clock) of every 1024th workgroup.  The stamps pin the instruction schedule (and cost registers), so the
// production library is built without them.
#ifdef SD_RS_TRACING
#define SD_TR(i)                                                                                         \
    do {                                                                                                 \
        if (p->trace != nullptr && lane == 0 && (blockIdx.x & 1023) == 7)                                \
            p->trace[((int64_t)(blockIdx.x >> 10) * kW + wave) * 16 + (i)] = wall_clock64();              \
    } while (0)
#else
#define SD_TR(i) do { } while (0)
#endif

template <int K, int MODE, int KIND, bool IDENT, bool SLAB>
__device__ __forceinline__ void segment_body(ParamsPtr p, int64_t tile_id, int g, char* smem_raw) {
    static_assert(MODE == MODE_FIT || MODE == MODE_RANK || MODE == MODE_APPLY || MODE == MODE_BOTH, "unknown mode");
    constexpr bool kRank = MODE == MODE_RANK || MODE == MODE_BOTH;    // ranks of the x_fut samples are computed here
    constexpr bool kApply = MODE == MODE_APPLY || MODE == MODE_BOTH;  // ... and mapped through the y CDF here
    constexpr bool kTas = KIND == SD_BCSD_TAS;
    constexpr int CH = Chunk<K>::CH;
    constexpr int NR = (K + 1) / 2;
    // per-segment quantities derived from (p, tile_id, g, thread id); MODE_BOTH derives them a second time from
    // laundered inputs between its two halves so that nothing but the ranks and x_climo stays live across them
    double *tile, *scratch, *row;
    const double* rcp;
    int RS, wave, lane, begf, n, begp, m;
    int64_t c0, c0l, c, seg;
    bool cell_ok, vec_f, vec_p;
#define SD_DERIVE()                                                                                              \
    do {                                                                                                         \
        tile = reinterpret_cast<double*>(smem_raw);                                                              \
        RS = p->RS;                                                                                              \
        scratch = tile + kW * RS; /* 64 doubles */                                                               \
        rcp = scratch + 64;       /* 16 doubles: correctly rounded 1/c, c = 1..9 */                              \
        c0 = tile_id * kW;                                                                                       \
        c0l = (p->ablate & 128) ? 0 : c0; /* dev: every tile loads the cells of tile 0 (cache-resident inputs) */ \
        const int t_ = tid_now();                                                                                \
        wave = t_ / kWave;                                                                                       \
        lane = t_ % kWave;                                                                                       \
        c = c0 + wave;                                                                                           \
        cell_ok = c < p->C;                                                                                      \
        row = tile + wave * RS;                                                                                  \
        seg = c * p->G + g;                                                                                      \
        begf = p->off_f[g];                                                                                      \
        n = p->off_f[g + 1] - begf;                                                                              \
        begp = 0;                                                                                                \
        m = 0;                                                                                                   \
        if (MODE != MODE_FIT) {                                                                                  \
            begp = p->off_p[g];                                                                                  \
            m = p->off_p[g + 1] - begp;                                                                          \
        }                                                                                                        \
        vec_f = (p->ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->y) & 15) == 0) &&                          \
                (p->X == nullptr || (reinterpret_cast<uintptr_t>(p->X) & 15) == 0);                             \
        vec_p = MODE != MODE_FIT && (p->ld_p % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->Xp) & 15) == 0);     \
    } while (0)
    SD_DERIVE();
    if (MODE != MODE_FIT ? m == 0 : n == 0) return;

    SD_TR(0);
    // ---- x climatology (bcsd.py:222); PR only validates X ------------------------------------------
    // RANK: the x_hist rows and then the x_fut tile are requested back to back (loads return in order), so the
    // column sums are reduced while the tile is still in flight: one exposed memory latency instead of two.
    double xc = 0.0;
    TileRegs<NR> xf;
    const bool dual = kRank && !(p->ablate & 16);
    if (MODE == MODE_APPLY || (kRank && p->from_state)) {
        if (kTas && cell_ok) xc = p->x_climo[seg];
        if (dual) tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0l, p->C, vec_p, xf, (p->ablate & 256) ? 63 : -1);
    } else if (p->X != nullptr && n > 0 && !(p->ablate & 32)) {
        TileRegs<NR> xh;
        tile_issue<NR>(p->X, p->ld, p->ord_f + begf, n, c0l, p->C, vec_f, xh, (p->ablate & 256) ? 63 : -1);
        if (dual) tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0l, p->C, vec_p, xf, (p->ablate & 256) ? 63 : -1);
        xc = tile_reduce_mean<NR>(xh, n, c0, p->C, scratch, p->status_fit, wave, lane);
        if (kTas && lane == 0 && cell_ok) p->x_climo[seg] = xc;
    } else if (dual) {
        tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0l, p->C, vec_p, xf, (p->ablate & 256) ? 63 : -1);
    }

    SD_TR(1);
    unsigned rank2[NR];  // two 16-bit ranks per register (segments are < 65536 samples)
    if (kRank) {
        // ---- x_fut segment -> shifted series u -> rank of every sample in sort(u) --------------------
        if (!dual) tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0l, p->C, vec_p, xf, (p->ablate & 256) ? 63 : -1);
        tile_commit<NR>(xf, m, c0, p->C, tile + kPadFront, RS, p->status_p);
        if (kTas) zero_pads(row, m, lane, CH + 4);
        __syncthreads();
        SD_TR(2);
        double u[K];  // u = X - (rolling mean - x_climo) (bcsd.py:247-256); PR maps raw X (bcsd.py:167)
        double* sh = (kTas && p->shift != nullptr && cell_ok) ? p->shift + (seg * p->slab_k) * kWave + lane : nullptr;
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
                        if (sh != nullptr) sh[i * kWave] = shift;
                    }
                    u[i] = K * lane + i < m ? uv : __builtin_inf();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wave_fence();
        SD_TR(3);
        if (!(p->ablate & 1)) {
            double s[K];
#pragma unroll
            for (int i = 0; i < K; ++i) s[i] = u[i];
            sort_segment<K>(s, row, m, lane);  // self ECDF: np.sort(u) (quantile.py:462 via 505-521)
        }
        SD_TR(4);
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
            for (int len = (p->ablate & 2) ? 1 : m; len > 1;) {
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
        SD_TR(5);
        if (MODE == MODE_RANK) {
            if (cell_ok) {
                uint32_t* rk = p->ranks + (seg * p->slab_nr) * kWave + lane;
#pragma unroll
                for (int i = 0; i < NR; ++i) rk[i * kWave] = rank2[i];
            }
            SD_TR(6);
            return;
        }
        __syncthreads();  // MODE_BOTH: every wave is done with its sorted x row, the tile is reused for y
        // pin the ranks (and x_climo) into registers here: otherwise the scheduler sinks the last search step behind
        // the y phase and keeps the searched values alive across it (200 bytes/lane of scratch)
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(rank2[i]));
        asm volatile("" : "+v"(xc));
        asm volatile("" : "+s"(p), "+s"(tile_id), "+s"(g));
        SD_DERIVE();
    }
#undef SD_DERIVE

    if (MODE == MODE_APPLY) {  // issued first: the loads fly while y is sorted
        const uint32_t* rk = p->ranks + (seg * p->slab_nr) * kWave + lane;
#pragma unroll
        for (int i = 0; i < NR; ++i) rank2[i] = cell_ok ? rk[i * kWave] : 0u;
    }

    // ---- y: climatology + sorted segment in the wave's row ------------------------------------------
    double yc = 0.0;
    if (!(kApply && p->from_state)) {
        if (n > 0) {
            load_tile<NR>(p->y, p->ld, p->ord_f + begf, n, c0l, p->C, vec_f, tile, RS, p->status_fit);
            __syncthreads();
            SD_TR(2);
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
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = K * lane + i < n ? v[i] : __builtin_inf();
            wave_fence();
            SD_TR(3);
            if (!(p->ablate & 4)) sort_segment<K>(v, row, n, lane);  // quantile.py:462 np.sort
            if (MODE == MODE_FIT && cell_ok) {
                double* dst = p->ys + c * p->Tf + begf;
                for (int i = lane; i < n; i += kWave) dst[i] = row[i];
            }
        }
    } else {
        if (cell_ok) {
            yc = p->y_climo[seg];
            const double* src = p->ys + c * p->Tf + begf;
            for (int i = lane; i < n; i += kWave) row[i] = src[i];
        }
        wave_fence();
    }
    SD_TR(4);
    if (!kApply) return;

    // No shift slab: the x_fut tile is read a second time (its RANK twin fetched it moments ago: L2 / Infinity
    // Cache) to recompute the rolling mean; the loads are issued here and fly during the lookups.
    constexpr bool reload = kTas && !SLAB;
    TileRegs<NR> xf2;
    if (reload) tile_issue<NR>(p->Xp, p->ld_p, p->ord_p + begp, m, c0l, p->C, vec_p, xf2);

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
            const int e = n < 10 ? n : 10;
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
            const int idx = (p->ablate & 8) ? 0 : qi[r];
            const double w = (p->ablate & 8) ? 0.0 : qv[r];
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

    SD_TR(5);
    // ---- restore the climate-trend shift (bcsd.py:263-267) / ratio anomalies (bcsd.py:170-185) ------
    if (kTas) {
        if (SLAB) {
            if (cell_ok) {
                const double* sh = p->shift + (seg * p->slab_k) * kWave + lane;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    double res = sh[i * kWave] + q[i];   // bcsd.py:253,263
                    if (p->return_anoms) res = res - yc;  // bcsd.py:266-267
                    q[i] = res;
                }
            }
        } else {
            __syncthreads();  // all lookups done: rows are free again
            tile_commit<NR>(xf2, m, c0, p->C, tile + kPadFront, RS, p->status_p);
            zero_pads(row, m, lane, CH + 4);
            __syncthreads();
            SD_TR(6);
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
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) q[i] = p->return_anoms ? q[i] / yc : q[i];  // bcsd.py:170-185
    }
    SD_TR(7);
    wave_fence();  // the wave's own row is rewritten in time order
    {
        const int base = K * lane;
#pragma unroll
        for (int i = 0; i < K; ++i) row[base + i < m ? base + i : m] = q[i];
    }
    __syncthreads();
    SD_TR(8);
    const bool vec_o = (p->ld_out % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->out) & 15) == 0);
    if (!(p->ablate & 64)) store_tile(p->out, p->ld_out, p->ord_p + begp, m, c0, p->C, vec_o, tile, RS);
    SD_TR(9);
}

template <int K, int MODE, int OCC, int KIND, bool IDENT, bool SLAB>
__global__ void __launch_bounds__(kThreads, OCC) bcsd_rs_kernel(const Params) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();  // Params is the only kernel argument
    {
        double* rcp = reinterpret_cast<double*>(smem_raw) + kW * p->RS + 64;
        if (threadIdx.x < 16) {
            const double tab[16] = {0.0, 1.0, 0.5, 1.0 / 3.0, 0.25, 0.2, 1.0 / 6.0, 1.0 / 7.0, 0.125, 1.0 / 9.0, 0, 0, 0, 0, 0, 0};
            rcp[threadIdx.x] = tab[threadIdx.x];
        }
    }
    // XCD-aware workgroup -> (tile, group): workgroup b runs on XCD b % 8; XCD x owns tiles [x*tx, (x+1)*tx)
    // and walks them tile-fastest, so the two 64-byte halves of a 128-byte line are fetched by workgroups
    // that are adjacent in time on the same L2.
    const int64_t tx = (p->ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7;
    const int64_t jb = blockIdx.x >> 3;
    const int64_t tile_id = xcd * tx + jb % tx;
    int g = (int)(jb / tx);
    if (p->gmask != 0ull) {  // this launch serves a subset of the groups: the g-th set bit
        unsigned long long m = p->gmask;
        for (int i = 0; i < g; ++i) m &= m - 1;
        if (m == 0ull) return;
        g = __builtin_ctzll(m);
    }
    if (tile_id >= p->ntiles || g >= p->G) return;
    segment_body<K, MODE, KIND, IDENT, SLAB>(p, tile_id, g, smem_raw);
}

template <int K, int MODE, int OCC, int KIND, bool IDENT, bool SLAB>
int launch_koki(sd_ctx* ctx, const Params& p, const char* name) {
    size_t lds = ((size_t)kW * p.RS + 64 + 16) * sizeof(double);
    if (const char* e = getenv("SD_RS_LDS_PAD")) lds += (size_t)atoi(e);  // dev: force one workgroup per CU
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_rs_kernel<K, MODE, OCC, KIND, IDENT, SLAB>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t tx = (p.ntiles + 7) / 8;
    const int64_t nblocks = 8 * tx * (p.gmask ? __builtin_popcountll(p.gmask) : p.G);
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "grid too large");
    Params q = p;
    const bool tracing = getenv("SD_RS_TRACE") != nullptr;
    const size_t nsamp = (size_t)(nblocks >> 10) + 1, trace_bytes = nsamp * kW * 16 * sizeof(long long);
    sd_scratch trace;
    if (tracing) {
        SD_HIP(trace.alloc(ctx, trace_bytes));
        SD_HIP(hipMemsetAsync(trace.p, 0, trace_bytes, ctx->stream));
        q.trace = trace.as<long long>();
    }
    SD_LAUNCH(ctx, name, (bcsd_rs_kernel<K, MODE, OCC, KIND, IDENT, SLAB>), dim3((unsigned)nblocks), dim3(kThreads), lds, q);
    if (tracing) {  // mean time between consecutive stamps over the sampled waves, in microseconds
        std::vector<long long> h(nsamp * kW * 16);
        SD_HIP(hipMemcpyAsync(h.data(), trace.p, trace_bytes, hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        double sum[16] = {};
        long long cnt[16] = {};
        for (size_t w = 0; w < nsamp * kW; ++w) {
            long long prev = 0;
            for (int i = 0; i < 16; ++i) {
                const long long t = h[w * 16 + i];
                if (t == 0) continue;
                if (prev) { sum[i] += (double)(t - prev) * 0.01; ++cnt[i]; }
                prev = t;
            }
        }
        fprintf(stderr, "[trace] %s:", name);
        for (int i = 1; i < 16; ++i)
            if (cnt[i]) fprintf(stderr, " ->%d %.2fus", i, sum[i] / (double)cnt[i]);
        fprintf(stderr, "\n");
    }
    return SD_OK;
}

template <int K, int MODE, int OCC, int KIND>
int launch_kok(sd_ctx* ctx, const Params& p, const char* name) {
    // IDENT and SLAB only change MODE_APPLY code (RANK tests p.shift at run time: one store per sample)
    if constexpr (MODE == MODE_APPLY || MODE == MODE_BOTH) {
        if constexpr (KIND == SD_BCSD_TAS && MODE == MODE_APPLY) {
            if (p.shift != nullptr)
                return p.identity ? launch_koki<K, MODE, OCC, KIND, true, true>(ctx, p, name)
                                  : launch_koki<K, MODE, OCC, KIND, false, true>(ctx, p, name);
        }
        if (p.identity) return launch_koki<K, MODE, OCC, KIND, true, false>(ctx, p, name);
    }
    return launch_koki<K, MODE, OCC, KIND, false, false>(ctx, p, name);
}

template <int K, int MODE, int OCC>
int launch_ko(sd_ctx* ctx, const Params& p, const char* name) {
    return p.kind == SD_BCSD_TAS ? launch_kok<K, MODE, OCC, SD_BCSD_TAS>(ctx, p, name)
                                 : launch_kok<K, MODE, OCC, SD_BCSD_PR>(ctx, p, name);
}

template <int K, int MODE>
int launch_k(sd_ctx* ctx, const Params& p, const char* name) {
    if constexpr (K == 13) {  // development: 80-VGPR build (6 waves/SIMD, three workgroups per CU when the rows fit)
        if (const char* e = getenv("SD_RS_OCC"))
            if (atoi(e) == 6) return launch_ko<K, MODE, 6>(ctx, p, name);
    }
    return launch_ko<K, MODE, (K > 21 ? 2 : 4)>(ctx, p, name);
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

static int rs_width(int nmax) { return nmax <= 64 * 5 ? 5 : nmax <= 64 * 13 ? 13 : nmax <= 64 * 19 ? 19 : nmax <= 64 * 21 ? 21 : 33; }  // as in launch_mode

int sd_bcsd_rs_row_stride(int nmax) {
    const int K = rs_width(nmax);
    const int CH = K >= 14 ? (K + 2) / 3 : K;
    int rs = (nmax + K - 1) / K * K + 1;  // the sort stores the +inf pads of the last run; one readable slot past the end
    const int roll = sdrs::kPadFront + nmax + CH + 4;  // time-ordered segment with zero pads for the rolling windows
    if (rs < roll) rs = roll;
    while (rs % 4 != 2) ++rs;  // cell rows land 8 or 24 banks apart: conflict-free transposing stores
    return rs;
}

void sd_bcsd_rs_handoff_bytes(int nmax, int64_t C, int G, size_t* rank_bytes, size_t* shift_bytes) {
    const size_t K = (size_t)rs_width(nmax), segs = (size_t)C * (size_t)G;
    *rank_bytes = segs * ((K + 1) / 2) * 64 * sizeof(uint32_t);
    *shift_bytes = segs * K * 64 * sizeof(double);
}

static int rs_launch_one(sd_ctx* ctx, int mode, const sdrs::Params& p, int nmax) {
    switch (mode) {
        case sdrs::MODE_FIT: return sdrs::launch_mode<sdrs::MODE_FIT>(ctx, p, nmax, "bcsd_rs_fit_kernel");
        case sdrs::MODE_RANK: return sdrs::launch_mode<sdrs::MODE_RANK>(ctx, p, nmax, "bcsd_rs_rank_kernel");
        case sdrs::MODE_APPLY: return sdrs::launch_mode<sdrs::MODE_APPLY>(ctx, p, nmax, "bcsd_rs_apply_kernel");
        case sdrs::MODE_BOTH: return sdrs::launch_mode<sdrs::MODE_BOTH>(ctx, p, nmax, "bcsd_rs_rank_apply_kernel");
        default: return sd_set_error(SD_ERR_INVALID, "unknown register-sort mode %d", mode);
    }
}

int sd_bcsd_rs_launch(sd_ctx* ctx, int mode, const sdrs::Params& p, int nmax, const int* group_len) {
    sdrs::Params q = p;
    const int kmax = rs_width(nmax);
    q.gmask = 0ull;
    q.slab_nr = (kmax + 1) / 2;
    q.slab_k = kmax;
    const char* e = getenv("SD_RS_SPLIT");  // "0": one launch of the widest kernels for every group (A/B testing)
    if (kmax == 21 && group_len != nullptr && p.G <= 64 && !(e && e[0] == '0')) {
        unsigned long long narrow = 0ull, wide = 0ull;
        for (int g = 0; g < p.G; ++g) (group_len[g] <= 64 * 19 ? narrow : wide) |= 1ull << g;
        if (narrow != 0ull && wide != 0ull) {
            q.gmask = wide;
            SD_TRY(rs_launch_one(ctx, mode, q, nmax));
            q.gmask = narrow;
            return rs_launch_one(ctx, mode, q, 64 * 19);
        }
    }
    return rs_launch_one(ctx, mode, q, nmax);
}
