// Wave-level building blocks of the BCSD kernels (device code shared by sd_bcsd_rs.hip and sd_bcsd_fx.hip): tile movement between time-major HBM fields and per-cell LDS rows, the per-wave merge sort, the
// 9-sample rolling mean (bcsd.py:247-250) and the Cunnane plotting-position helpers (quantile.py:23-43).
#pragma once
#include "sd_internal.h"
#include "sd_sortnet.h"

namespace sdw {

using namespace sdsort;

constexpr int kWave = 64;
constexpr int kW = 8;          // cells per workgroup
constexpr int kThreads = 512;  // 8 waves
constexpr int kRowsPerPass = kThreads / 4;  // 4 lanes (16 B each) cover the 8 cells of one row
// LDS layout of a workgroup: [column-sum exchange: 64 doubles][1/c table: 16][per-cell flags: 8][tile: kW rows of RS].
// The small areas come first so that no row starts at LDS address 0: the searches keep "address of element - 1"
// positions and compare them as unsigned numbers.
constexpr int kHeadDoubles = 64 + 16 + 8;

// The thread index behind an opaque barrier: keeps the compiler from hoisting everything derived from it to the
// top of the kernel (and keeping it alive in registers across the sorts).
__device__ __forceinline__ int tid_now() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ bool finite64(double v) {
    return (__double_as_longlong(v) & 0x7ff0000000000000ll) != 0x7ff0000000000000ll;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Least-squares line of a blocked series (lane holds samples j = K * lane + i, valid while j < n) on j = 0 .. n-1:
// trend.py:51 (LinearRegression on np.arange(len(X))); centred sums like sklearn's _preprocess_data + lstsq.
template <int K>
__device__ __forceinline__ void trend_line(const double (&v)[K], int n, int lane, double* slope, double* icpt) {
    const double tbar = 0.5 * (double)(n - 1);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) s += K * lane + i < n ? v[i] : 0.0;
    const double vbar = wave_sum(s) / (double)n;
    double sxy = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = K * lane + i;
        sxy += j < n ? ((double)j - tbar) * (v[i] - vbar) : 0.0;
    }
    sxy = wave_sum(sxy);
    const double dn = (double)n;
    const double sxx = dn * (dn * dn - 1.0) / 12.0;
    const double a = n > 1 ? sxy / sxx : 0.0;
    *slope = a;
    *icpt = vbar - a * tbar;
}
// Lanes of one wave exchange data through LDS inside the sort.  The hardware serves a wave's LDS
// requests in order; for the compiler the exchange needs a wavefront-scope fence plus the wave barrier.
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS access at an absolute 32-bit LDS byte address (searches keep positions as addresses: add, compare, select)
typedef __attribute__((address_space(3))) const double lds_cdouble_t;
__device__ __forceinline__ double lds_f64(unsigned addr) { return *reinterpret_cast<lds_cdouble_t*>((uintptr_t)addr); }
__device__ __forceinline__ unsigned lds_addr(const void* generic_ptr_into_lds) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)generic_ptr_into_lds;
}

// ---- wave-level merge sort of row[0..n): runs of K per lane -> fully sorted, in place ------------
// Round 0 (lane pairs) runs in registers, rounds 1..5 through the wave's LDS row.
//
// Round 0: odd lanes work on negated values, so that every lane executes the same instructions: after the register
// sort a lane holds its run ascending *in its own domain*; it fetches register j of its neighbour (DPP quad_perm
// [1,0,3,2], sign flipped = the neighbour's value seen from the own domain, which is the neighbour's run in
// descending order) and keeps z[j] = min(own[j], fetched[j]): the even lane is left with the K smallest of the 2K
// values, the odd lane with the negated K largest, both as an ascending-then-descending sequence that the pruned
// bitonic merger sorts.  The pair's run of 2K goes to LDS in true order (odd lanes store back to front).  Compared
// with a first round through LDS this saves the store of the unmerged runs, the co-rank search and the window loads.
//
// Rounds r >= 1 merge pairs of runs of length K << r.  Every lane owns K consecutive output positions of
// its pair: it finds its co-rank (merge path) by binary search, loads the matching windows of A and B
// (exactly one LDS read per element, all independent), merges them in registers and the wave writes
// the K outputs back in place.  LDS requests of one wave are served in order: no barrier needed.
//
// Co-rank = smallest i in [lo0, hi0] with i == hi0 or A[i] > B[d-1-i] (ties go to A: stable).  The predicate
// A[i] <= B[d-1-i] is monotone, so the branch-free search "answer in [base, base + len)" runs with the same
// wave-uniform stride sequence len -> len - len/2 for every lane (starting from the longest possible range, L + 1
// candidates); a probe past the lane's own range is switched off by one address comparison (t <= hi0), whatever it
// reads.  Positions are LDS byte addresses of A[base - 1]; the B probe sits at S - t.  Per step: add, subtract,
// two reads, two compares, one select.
//
// KEEP: the K sorted values of the positions a lane owns (K * lane ...) are also returned in v[] (taken from the
// registers of the last round; segments of at most 2K slots read them back).
typedef __attribute__((address_space(3))) double lds_double_t;
__device__ __forceinline__ void lds_store_f64(unsigned addr, double x) { *reinterpret_cast<lds_double_t*>((uintptr_t)addr) = x; }

__device__ __forceinline__ double flip_sign(double x, int mask) {  // mask = 0 or 0x80000000
    return __hiloint2double(__double2hiint(x) ^ mask, __double2loint(x));
}
// the value the neighbouring lane (lane ^ 1) holds in x, negated
__device__ __forceinline__ double neighbour_negated(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, true) ^ (int)0x80000000;
    return __hiloint2double(hi, lo);
}

template <int K, bool KEEP>
__device__ __forceinline__ void merge_rounds(double* row, int np, int lane, double (&v)[K]) {
    // np = number of slots being sorted, a multiple of K: the pads that fill the last lane's run are
    // ordinary elements (they sort to the end), so every participating lane merges exactly K outputs and
    // no per-element validity test is needed; lanes past np sit out (one divergent branch per round).
    constexpr MergeNet<K> net{};
    const unsigned rowb = lds_addr(row);
    double w[K];
#pragma unroll 1
    for (int r = 1; r < 6; ++r) {
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
        const int lo0 = d - LB > 0 ? d - LB : 0, hi0 = d < LA ? d : LA;
        const unsigned am8 = rowb + 8u * (unsigned)a0 - 8u;        // &A[-1]
        const unsigned hi_addr = am8 + 8u * (unsigned)hi0;          // t <= hi_addr  <=>  candidate index <= hi0
        const unsigned S = rowb + 8u * (unsigned)(a1 + d) + am8;    // &B[d - i] == S - &A[i - 1]
        unsigned pos = am8 + 8u * (unsigned)lo0;                    // &A[base - 1], base = lo0
        // The co-rank search and the window loads are chains of dependent LDS round trips: a wave in them gets issue priority
        // over the waves that grind through their merge networks (which have independent work to fill the gaps) -- 1.5 % of
        // the fused kernel's time on the bench, measured.
        __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (int len = L + 1; len > 1;) {
            int half = len >> 1;
            if ((half & 15) == 0) --half;  // strides that are multiples of 16 doubles pile the probes on two banks
            len -= half;
            const unsigned t = pos + 8u * (unsigned)half;
            const bool ok = (t <= hi_addr) && (lds_f64(t) <= lds_f64(S - t));
            pos = ok ? t : pos;
        }
        const int lo = (int)(pos - am8) >> 3;
        const int inext = __shfl_down(lo, 1, kWave);
        const int ihi = (d + K >= LA + LB) ? LA : inext;  // co-rank of the end of this lane's window
        const int acnt = ihi - lo;                         // elements taken from A; K - acnt from B
        if (busy) {
            const double* pa = row + a0 + lo;                       // A window, ascending: pa[s], s < acnt
            const double* pq = row + a1 + (d - lo) + (K - acnt) - 1 + acnt;  // B window read backwards: pq[-s], s >= acnt
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const double* src = s < acnt ? pa : pq - 2 * s;     // (pq - 2s)[s] == pq[-s]
                w[s] = src[s];
                if (s % 7 == 6) __builtin_amdgcn_sched_barrier(0);  // issue the loads in batches
            }
        }
        __builtin_amdgcn_s_setprio(0);  // (outside the divergent region: every wave drops back, busy lanes or not)
        if (busy) {
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
    if (KEEP) {
        if (np > 2 * K) {  // the last round covered the whole row: its registers are the lane's sorted positions
#pragma unroll
            for (int s = 0; s < K; ++s) v[s] = w[net.out[s]];
        } else {
            const double* src = row + (K * lane < np ? K * lane : 0);
#pragma unroll
            for (int s = 0; s < K; ++s) v[s] = src[s];
        }
    }
}

// sort the wave's segment: v[] = K consecutive samples per lane (pads sort last: every lane, also those past the
// segment, must hold K values that are >= all data), result in row[0..n); the row must have ceil(n / K) * K + 1
// slots (the pads of the last run are stored and sorted like data).
template <int K, bool KEEP = false>
__device__ __forceinline__ void sort_segment(double (&v)[K], double* row, int n, int lane, bool rounds = true) {
    constexpr MergeNet<K> net{};
    const int np = (n + K - 1) / K * K;
    const int neg = (lane & 1) << 31;
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = flip_sign(v[i], neg);
    sort_registers<K>(v);
    double z[K];
#pragma unroll
    for (int j = 0; j < K; ++j) z[j] = vmin(v[j], neighbour_negated(v[j]));
#pragma unroll
    for (int c = 0; c < net.n; ++c) {
        const double mn = vmin(z[net.a[c]], z[net.b[c]]);
        const double mx = vmax(z[net.a[c]], z[net.b[c]]);
        z[net.a[c]] = mn;
        z[net.b[c]] = mx;
    }
    if (K * lane < np) {
        // even lane: positions K*lane + s; odd lane (negated, so back to front): K*lane + K-1 - s
        unsigned a = lds_addr(row) + 8u * (unsigned)(K * lane) + ((lane & 1) ? 8u * (K - 1) : 0u);
        const unsigned step = (lane & 1) ? (unsigned)-8 : 8u;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            lds_store_f64(a, flip_sign(z[net.out[s]], neg));
            a += step;
        }
    }
    wave_fence();
    if (rounds) merge_rounds<K, KEEP>(row, np, lane, v);  // rounds == false: timing experiments of the development library only
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
                                           int nrows, int64_t c0, int64_t C, bool vec_ok, TileRegs<RPT>& t) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    const bool full = vec_ok && c + 1 < C;
    int ti[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRowsPerPass;
        ti[k] = ord[r < nrows ? r : 0];
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
                                            int32_t* status, int* bad_cell = nullptr) {
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
    if (bad_cell != nullptr) {  // workgroup-local copy of the verdict (LDS, [kW])
        if (bad0) bad_cell[2 * cp] = 1;
        if (bad1) bad_cell[2 * cp + 1] = 1;
    }
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
                                                   int32_t* status, int wave, int lane, int* bad_cell = nullptr) {
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
    if (bad_cell != nullptr) {
        if (bad0) bad_cell[2 * cp] = 1;
        if (bad1) bad_cell[2 * cp + 1] = 1;
    }
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
// table of correctly rounded reciprocals 1/c, c = 1..9 (16 doubles of LDS), written by the first threads of a workgroup
__device__ __forceinline__ void fill_rcp_table(double* rcp) {
    if (threadIdx.x < 16) {
        const double tab[16] = {0.0, 1.0, 0.5, 1.0 / 3.0, 0.25, 0.2, 1.0 / 6.0, 1.0 / 7.0, 0.125, 1.0 / 9.0, 0, 0, 0, 0, 0, 0};
        rcp[threadIdx.x] = tab[threadIdx.x];
    }
}

constexpr double kAlpha = 0.4, kBeta = 0.4;
__device__ __forceinline__ double pp_denom(int n) { return ((double)n + 1.0 - kAlpha) - kBeta; }
__device__ __forceinline__ double pp_at(int i, double denom) { return ((double)(i + 1) - kAlpha) / denom; }

// least-squares line through the e points (pp[first + i], ysg[first + i]) (quantile.py:532-543, centred form)
__device__ inline void ols_line(const double* ysg, int first, int e, double denom, double* slope, double* icpt) {
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

// Samples handled together in the rolling / search / lookup phases (bounded register pressure).
template <int K>
struct Chunk {
    static constexpr int CH = K >= 14 ? (K + 2) / 3 : K;
};

// XCD-aware workgroup -> (tile, group slot): workgroup b runs on XCD b % 8; XCD x owns tiles [x*tx, (x+1)*tx) and
// walks them tile-fastest, so the two 64-byte halves of a 128-byte line are fetched by workgroups that are adjacent
// in time on the same L2.
__device__ __forceinline__ void xcd_tile_of_block(unsigned b, int64_t ntiles, int64_t* tile_id, int* gslot) {
    const int64_t tx = (ntiles + 7) / 8;
    const int xcd = (int)(b & 7u);
    const int64_t jb = (int64_t)(b >> 3);
    *tile_id = xcd * tx + jb % tx;
    *gslot = (int)(jb / tx);
}
// the gslot-th set bit of mask (groups served by a launch); -1 if there is none
__device__ __forceinline__ int nth_set_bit(unsigned long long mask, int k) {
    for (int i = 0; i < k; ++i) mask &= mask - 1;
    return mask == 0ull ? -1 : __builtin_ctzll(mask);
}

}  // namespace sdw
