// EXPERIMENT (round 3) -- NOT BUILT, NOT PART OF THE LIBRARY.  Kept as the record of a measured negative result (DESIGN.md
// section 7): parity-green (tests/test_gpu_bcsd.py, bench parity check), 31.2 ms per step of BASELINE configs[1] against
// 20.0 ms for sd_bcsd_fz.hip.  With one workgroup per CU every wave is in the same phase at the same time: the two sorts
// take 14 us each per item at ~50 % vector-ALU utilisation, where two independent 512-thread workgroups of the one-wave-per-
// segment kernel overlap one's searches with the other's merge networks.  What it demonstrates and what later rounds can
// reuse: (1) amdgpu_waves_per_eu(6, 8) caps the compiler at 80 registers while inline assembly that names v80 .. v127 keeps
// the allocation at 128 -- hand-allocated registers for loads in flight and for values that live across the sorts (the
// register allocator spills both kinds to scratch whatever room there is); (2) rows requested one phase ahead through such
// banks arrive on time (the commit of the prefetched y_obs tile takes 0.85 us); (3) ordinary loads behind a bank request wait
// for the whole tile (loads return in order): row indices must be fetched before the first request.
// To build it again: add the file to SRC in the Makefile, restore Params::fz_version and the dispatch in sd_bcsd_fz_launch.
//
// BcsdTemperature fit + predict, second form of the fused kernel: TWO waves per (cell, month) segment, one persistent
// 1024-thread workgroup per CU, every tile requested one phase ahead.
//
// Reference semantics as in sd_bcsd_fz.hip (bcsd.py:197-269, quantile.py:81-147, 438-545); what changes is the shape:
//
//   * a segment (n ~ 1 240 samples of one cell and month) is owned by 128 lanes (waves 2s, 2s+1 of the workgroup) with
//     K = 11 consecutive samples per lane instead of 64 lanes with K = 21: every per-lane array halves, so the shift
//     (rolling mean - x_climo, bcsd.py:253) of a lane's samples STAYS IN REGISTERS from the x side to the last add
//     (bcsd.py:263): no second read of the x_fut tile, no second rolling mean, no spills;
//   * with the register file half empty, loads run one phase ahead of their use: the y_obs tile is requested before
//     the sort of the shifted series and committed to the (then free) LDS rows after it; the x_hist / x_fut tiles of the
//     NEXT (tile, month) item are requested before the sort of y_obs.  A workgroup is persistent (grid = one
//     workgroup per CU, items dealt out XCD-aware) so that there is a next item to prefetch;
//   * the merge sort runs rounds 0..5 inside a wave exactly like sd_wave.h and one more round across the two waves of
//     the segment (workgroup barriers, co-ranks exchanged through LDS).
//
// Ambiguous segments (two shifted samples with equal upper 48 bits) go to the work list like in sd_bcsd_fz.hip.
// States (from_state) stay on sd_bcsd_fz.hip.
#include "sd_bcsd_rs.h"
#include "sd_wave.h"

namespace sdfz2 {

#ifdef SD_DEV
__device__ long long g_trace[64 * 16];  // SD_FZ2_TRACE: phase stamps (100 MHz) of the first items of workgroup 0
#define SD_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && nitem < 64) g_trace[nitem * 16 + (k)] = wall_clock64(); } while (0)
#else
#define SD_STAMP(k) do { } while (0)
#endif

using namespace sdw;
using sdrs::Params;

typedef const Params __attribute__((address_space(4)))* ParamsPtr;

constexpr int kThreads2 = 1024;           // 16 waves: two per cell of the tile
constexpr int kSegLanes = 128;            // lanes per segment
constexpr int kRows2 = kThreads2 / 4;     // rows per tile pass: 4 lanes (16 B each) cover the 8 cells of one row
constexpr unsigned kTagMask = 0xffffu;    // low 16 mantissa bits carry 8 * position (positions < 128 * 11)
constexpr int kPadHi = 0x7fe00000;        // pads: 2^1023 * (1 + position * 2^-36), tagged like data
// LDS: [column sums: 16 waves x 8][1/c table: 16][flags: 8][co-rank exchange: 8 x 128 ints][tile: 8 rows of RS]
constexpr int kSumDoubles = 16 * kW;
constexpr int kXchDoubles = kW * kSegLanes / 2;
constexpr int kHead2 = kSumDoubles + 16 + 8 + kXchDoubles;

__device__ __forceinline__ double from_words(unsigned lo, int hi) { return __hiloint2double(hi, (int)lo); }

// ---- hand-allocated registers ------------------------------------------------------------------------------------------
// The compiler keeps its own values in v0 .. v79 (amdgpu_waves_per_eu(6, 8) on the kernel; the workgroup still gets 128
// registers per lane because the inline assembly below names v80 .. v127).  Those 48 registers are allocated by hand:
//   v80 .. v99     bank 0: the 5 rows x 16 bytes a thread loads of one tile, requested one phase ahead
//   v100 .. v119   bank 1: the same for a second tile in flight
//   v120 .. v123   the shifts (bcsd.py:253) of the lane's last two samples; the other nine are parked in LDS
// The register allocator spills such long-lived values to scratch although the sorts leave room for them -- and it
// cannot wait for a load it does not know about, so in-flight rows must not be its values at all.  Loads return in order
// and the hardware counter also counts the compiler's own memory operations: "s_waitcnt vmcnt(0)" before the first read
// of a bank is always sufficient.
// Row indices of a thread's 5 rows (ordinary loads: they must be complete before the first request into a bank -- loads
// return in order, so a value loaded behind a bank request would only arrive after the whole tile).
__device__ __forceinline__ void bank_rows(const int32_t* __restrict__ ord, int nrows, int (&ti)[5]) {
    const int rr = tid_now() >> 2;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int r = rr + k * kRows2;
        ti[k] = ord[r < nrows ? r : 0];
    }
}
template <int BANK>
__device__ __forceinline__ void bank_issue(const double* __restrict__ src, int64_t ld, const int (&ti)[5], int64_t c0, int64_t C) {
    const int tid = tid_now();
    const int cp = tid & 3;
    const int64_t c = c0 + 2 * cp;
    const double* colp = src + (c + 1 < C ? c : c0);  // (C is even: a pair past the last cell re-reads pair 0)
    const double* a0 = row_of(colp, ti[0], ld);
    const double* a1 = row_of(colp, ti[1], ld);
    const double* a2 = row_of(colp, ti[2], ld);
    const double* a3 = row_of(colp, ti[3], ld);
    const double* a4 = row_of(colp, ti[4], ld);
    if (BANK == 0) {
        asm volatile("global_load_dwordx4 v[80:83], %0, off" : : "v"(a0) : "memory", "v80", "v81", "v82", "v83");
        asm volatile("global_load_dwordx4 v[84:87], %0, off" : : "v"(a1) : "memory", "v84", "v85", "v86", "v87");
        asm volatile("global_load_dwordx4 v[88:91], %0, off" : : "v"(a2) : "memory", "v88", "v89", "v90", "v91");
        asm volatile("global_load_dwordx4 v[92:95], %0, off" : : "v"(a3) : "memory", "v92", "v93", "v94", "v95");
        asm volatile("global_load_dwordx4 v[96:99], %0, off" : : "v"(a4) : "memory", "v96", "v97", "v98", "v99");
    } else {
        asm volatile("global_load_dwordx4 v[100:103], %0, off" : : "v"(a0) : "memory", "v100", "v101", "v102", "v103");
        asm volatile("global_load_dwordx4 v[104:107], %0, off" : : "v"(a1) : "memory", "v104", "v105", "v106", "v107");
        asm volatile("global_load_dwordx4 v[108:111], %0, off" : : "v"(a2) : "memory", "v108", "v109", "v110", "v111");
        asm volatile("global_load_dwordx4 v[112:115], %0, off" : : "v"(a3) : "memory", "v112", "v113", "v114", "v115");
        asm volatile("global_load_dwordx4 v[116:119], %0, off" : : "v"(a4) : "memory", "v116", "v117", "v118", "v119");
    }
}

// the bank's rows as ordinary values (after everything requested so far has arrived)
template <int BANK>
__device__ __forceinline__ void bank_take(TileRegs<5>& t) {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    if (BANK == 0) {
        asm volatile("v_mov_b64 %0, v[80:81]\n\tv_mov_b64 %1, v[82:83]" : "=v"(t.v0[0]), "=v"(t.v1[0]));
        asm volatile("v_mov_b64 %0, v[84:85]\n\tv_mov_b64 %1, v[86:87]" : "=v"(t.v0[1]), "=v"(t.v1[1]));
        asm volatile("v_mov_b64 %0, v[88:89]\n\tv_mov_b64 %1, v[90:91]" : "=v"(t.v0[2]), "=v"(t.v1[2]));
        asm volatile("v_mov_b64 %0, v[92:93]\n\tv_mov_b64 %1, v[94:95]" : "=v"(t.v0[3]), "=v"(t.v1[3]));
        asm volatile("v_mov_b64 %0, v[96:97]\n\tv_mov_b64 %1, v[98:99]" : "=v"(t.v0[4]), "=v"(t.v1[4]));
    } else {
        asm volatile("v_mov_b64 %0, v[100:101]\n\tv_mov_b64 %1, v[102:103]" : "=v"(t.v0[0]), "=v"(t.v1[0]));
        asm volatile("v_mov_b64 %0, v[104:105]\n\tv_mov_b64 %1, v[106:107]" : "=v"(t.v0[1]), "=v"(t.v1[1]));
        asm volatile("v_mov_b64 %0, v[108:109]\n\tv_mov_b64 %1, v[110:111]" : "=v"(t.v0[2]), "=v"(t.v1[2]));
        asm volatile("v_mov_b64 %0, v[112:113]\n\tv_mov_b64 %1, v[114:115]" : "=v"(t.v0[3]), "=v"(t.v1[3]));
        asm volatile("v_mov_b64 %0, v[116:117]\n\tv_mov_b64 %1, v[118:119]" : "=v"(t.v0[4]), "=v"(t.v1[4]));
    }
}

// The shift of sample i of a lane, parked between the x side and the last add: samples 0 .. 8 in LDS (park[i][thread]:
// lane-contiguous, conflict-free), samples 9 and 10 in v120 .. v123.  (i is a constant once the loops are unrolled.)
constexpr int kParkLds = 9;
__device__ __forceinline__ void shift_park(double* park, int i, double v) {
    if (i < kParkLds) {
        park[i * kThreads2] = v;
    } else if (i == kParkLds) {
        asm volatile("v_mov_b64 v[120:121], %0" : : "v"(v) : "v120", "v121");
    } else {
        asm volatile("v_mov_b64 v[122:123], %0" : : "v"(v) : "v122", "v123");
    }
}
__device__ __forceinline__ double shift_take(const double* park, int i) {
    double v;
    if (i < kParkLds) {
        v = park[i * kThreads2];
    } else if (i == kParkLds) {
        asm volatile("v_mov_b64 %0, v[120:121]" : "=v"(v));
    } else {
        asm volatile("v_mov_b64 %0, v[122:123]" : "=v"(v));
    }
    return v;
}

// ---- tile movement for 1024 threads (same scheme as sd_wave.h: a thread owns rows rr, rr + 256, ...) ---------------
template <int RPT>
__device__ __forceinline__ void issue2(const double* __restrict__ src, int64_t ld, const int32_t* __restrict__ ord, int nrows,
                                       int64_t c0, int64_t C, bool vec_ok, TileRegs<RPT>& t) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    const bool full = c + 1 < C;  // C is even here (vec_ok), so a pair is inside or outside as a whole
    (void)vec_ok;
    int ti[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRows2;
        ti[k] = ord[r < nrows ? r : 0];
    }
    // 16-byte loads only (launcher: even pitches, 16-byte aligned fields); a pair past the last cell re-reads pair 0
    const double* colp = src + (full ? c : c0);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const double2 v = *reinterpret_cast<const double2*>(row_of(colp, ti[k], ld));
        t.v0[k] = v.x;
        t.v1[k] = v.y;
    }
}

template <int RPT>
__device__ __forceinline__ void commit2(const TileRegs<RPT>& t, int nrows, int64_t c0, int64_t C, double* tile, int RS,
                                        int32_t* status, int* bad_cell) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    double* d0 = tile + (2 * cp) * RS;
    double* d1 = d0 + RS;
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRows2;
        if (r < nrows) {
            bad0 |= !finite64(t.v0[k]);
            bad1 |= !finite64(t.v1[k]);
            d0[r] = t.v0[k];
            d1[r] = t.v1[k];
        }
    }
    if (bad0 && c < C) atomicOr(&status[c], SDI_NONFINITE);
    if (bad1 && c + 1 < C) atomicOr(&status[c + 1], SDI_NONFINITE);
    if (bad_cell != nullptr) {
        if (bad0) bad_cell[2 * cp] = 1;
        if (bad1) bad_cell[2 * cp + 1] = 1;
    }
}

// column means of the issued rows for the 8 cells (x climatology); the caller's wave serves cell `seg`
template <int RPT>
__device__ __forceinline__ double reduce_mean2(const TileRegs<RPT>& t, int nrows, int64_t c0, int64_t C, double* sums,
                                               int32_t* status, int wave, int lane, int seg, int* bad_cell) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    double s0 = 0.0, s1 = 0.0;
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool in = rr + k * kRows2 < nrows;
        bad0 |= in && !finite64(t.v0[k]);
        bad1 |= in && !finite64(t.v1[k]);
        s0 += in ? t.v0[k] : 0.0;
        s1 += in ? t.v1[k] : 0.0;
    }
    if (bad0 && c < C) atomicOr(&status[c], SDI_NONFINITE);
    if (bad1 && c + 1 < C) atomicOr(&status[c + 1], SDI_NONFINITE);
    if (bad0) bad_cell[2 * cp] = 1;
    if (bad1) bad_cell[2 * cp + 1] = 1;
#pragma unroll
    for (int o = 4; o <= 32; o <<= 1) {  // lanes with equal (lane & 3) hold the same cell pair
        s0 += __shfl_xor(s0, o, kWave);
        s1 += __shfl_xor(s1, o, kWave);
    }
    if (lane < 4) {
        sums[wave * kW + 2 * lane] = s0;
        sums[wave * kW + 2 * lane + 1] = s1;
    }
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += sums[w * kW + seg];
    __syncthreads();
    return tot / (double)nrows;
}

__device__ __forceinline__ void store2(double* __restrict__ dst, int64_t ld, const int32_t* __restrict__ ord, int nrows, int64_t c0,
                                       int64_t C, bool vec_ok, const double* tile, int RS) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    const double* s0 = tile + (2 * cp) * RS;
    const double* s1 = s0 + RS;
    (void)vec_ok;
    if (c + 1 < C) {
#pragma unroll 2
        for (int r = rr; r < nrows; r += kRows2) {
            double* q = row_of(dst + c, ord[r], ld);
            *reinterpret_cast<double2*>(q) = make_double2(s0[r], s1[r]);
        }
    }
}

// ---- merge sort of a segment by its 128 lanes -----------------------------------------------------------------------
// One merge round (pairs of runs of length K << r).  sl = lane of the segment (0..127).  Rounds whose merge groups fit
// a wave (r <= 5) are those of sd_wave.h; the round across the two waves (r == 6) synchronises with workgroup barriers
// (every wave of the workgroup runs the same rounds: np is the same for all segments of a tile) and passes the
// neighbour's co-rank through xch (the segment's 128 ints).
template <int K, bool STORE>
__device__ __forceinline__ void merge_round2(double* row, unsigned rowb, int np, int sl, int r, int* xch, double (&w)[K]) {
    constexpr MergeNet<K> net{};
    const bool cross = r >= 6;  // workgroup-uniform
    const int L = K << r;
    const int gl = sl & ((2 << r) - 1);  // lane within its merge group
    const int base = (sl - gl) * K;
    const int a0 = base < np ? base : np;
    const int a1 = base + L < np ? base + L : np;
    const int b1 = base + 2 * L < np ? base + 2 * L : np;
    const int LA = a1 - a0, LB = b1 - a1;
    const int d0 = gl * K;
    const bool busy = d0 < LA + LB;
    const int d = busy ? d0 : LA + LB;
    const int lo0 = d - LB > 0 ? d - LB : 0, hi0 = d < LA ? d : LA;
    const unsigned am8 = rowb + 8u * (unsigned)a0 - 8u;
    const unsigned hi_addr = am8 + 8u * (unsigned)hi0;
    const unsigned S = rowb + 8u * (unsigned)(a1 + d) + am8;
    unsigned pos = am8 + 8u * (unsigned)lo0;
    if (cross) __syncthreads();  // the runs of the previous round, written by both waves
#pragma unroll 1
    for (int len = L + 1; len > 1;) {
        int half = len >> 1;
        if ((half & 15) == 0) --half;
        len -= half;
        const unsigned t = pos + 8u * (unsigned)half;
        const bool ok = (t <= hi_addr) && (lds_f64(t) <= lds_f64(S - t));
        pos = ok ? t : pos;
    }
    const int lo = (int)(pos - am8) >> 3;
    int inext;
    if (cross) {
        xch[sl] = lo;
        __syncthreads();
        inext = xch[sl + 1 < kSegLanes ? sl + 1 : sl];
    } else {
        inext = __shfl_down(lo, 1, kWave);
    }
    const int ihi = (d + K >= LA + LB) ? LA : inext;
    const int acnt = busy ? ihi - lo : K;
    const double* pa = busy ? row + a0 + lo : row;
    const double* pq = row + a1 + (d - lo) + (K - acnt) - 1 + acnt;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const double* src = s < acnt ? pa : pq - 2 * s;
        w[s] = src[s];
    }
#pragma unroll
    for (int c = 0; c < net.n; ++c) {
        const double mn = vmin(w[net.a[c]], w[net.b[c]]);
        const double mx = vmax(w[net.a[c]], w[net.b[c]]);
        w[net.a[c]] = mn;
        w[net.b[c]] = mx;
    }
    if (cross) __syncthreads(); else wave_fence();  // every lane has read its windows
    if (STORE) {
        if (busy) {
            double* dst = row + a0 + d;
#pragma unroll
            for (int s = 0; s < K; ++s) dst[s] = w[net.out[s]];
        }
        if (cross) __syncthreads(); else wave_fence();
    }
}

// v[] = K consecutive samples per lane (pads sort last), result in row[0..np) (unless !STORE_LAST) and, KEEP, the K sorted
// values of the positions a lane owns in v[].  Called by every wave of the workgroup with the same n.
template <int K, bool KEEP, bool STORE_LAST>
__device__ __forceinline__ void sort_segment2(double (&v)[K], double* row, int n, int sl, int* xch) {
    constexpr MergeNet<K> net{};
    const int np = (n + K - 1) / K * K;
    const int neg = (sl & 1) << 31;
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
    if (K * sl < np) {
        unsigned a = lds_addr(row) + 8u * (unsigned)(K * sl) + ((sl & 1) ? 8u * (K - 1) : 0u);
        const unsigned step = (sl & 1) ? (unsigned)-8 : 8u;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            lds_store_f64(a, flip_sign(z[net.out[s]], neg));
            a += step;
        }
    }
    wave_fence();
    const unsigned rowb = lds_addr(row);
    int last = 0;  // the last round r with (K << r) < np
#pragma unroll
    for (int r = 1; r <= 6; ++r) last = (K << r) < np ? r : last;
#pragma unroll 1
    for (int r = 1; r < last; ++r) {
        double w[K];
        merge_round2<K, true>(row, rowb, np, sl, r, xch, w);
    }
    if (last >= 1) {  // workgroup-uniform
        double w[K];
        merge_round2<K, STORE_LAST || !KEEP>(row, rowb, np, sl, last, xch, w);
        if (KEEP) {
#pragma unroll
            for (int s = 0; s < K; ++s) v[s] = w[net.out[s]];
        }
    } else if (KEEP) {
        const double* src = row + (K * sl < np ? K * sl : 0);
#pragma unroll
        for (int s = 0; s < K; ++s) v[s] = src[s];
    }
}

// ---- the kernel -----------------------------------------------------------------------------------------------------
// item q of XCD x (workgroup b runs on XCD b % 8): tile x * tx + q % tx, group slot q / tx -- tile-fastest, so that the
// two 64-byte halves of a 128-byte line are fetched by workgroups of one L2 at about the same time.
struct Item {
    int64_t tile;
    int g, begf, n, begp, m;
    bool valid;
};
__device__ __forceinline__ Item decode_item(ParamsPtr p, int64_t q, int xcd, int64_t tx, int nslots) {
    Item it;
    it.valid = false;
    it.tile = xcd * tx + q % tx;
    const int gs = (int)(q / tx);
    it.g = gs < nslots ? (p->gmask != 0ull ? nth_set_bit(p->gmask, gs) : gs) : -1;
    it.begf = it.n = it.begp = it.m = 0;
    if (it.g >= 0 && it.g < p->G) {
        it.begf = p->off_f[it.g];
        it.n = p->off_f[it.g + 1] - it.begf;
        it.begp = p->off_p[it.g];
        it.m = p->off_p[it.g + 1] - it.begp;
        it.valid = it.tile < p->ntiles && it.m > 0;
    }
    return it;
}

template <int K, bool IDENT, int NR>
__global__ void __attribute__((amdgpu_flat_work_group_size(1024, 1024), amdgpu_waves_per_eu(6, 8))) bcsd_fz2_kernel(const Params) {
    static_assert(NR == 5 && K == 11, "the hand-allocated registers hold 5 rows per thread and 11 shifts per lane");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    // NR = rows a thread loads of one tile (256 rows per pass)
    constexpr int CH = 4;                                      // samples per rolling-mean chunk
    constexpr int NP2 = (K + 1) / 2;
    double* const sums = reinterpret_cast<double*>(smem_raw);
    double* const rcp = sums + kSumDoubles;
    int* const bad_cell = reinterpret_cast<int*>(rcp + 16);
    int* const xch_all = reinterpret_cast<int*>(rcp + 16 + 8);
    double* const tile = sums + kHead2;
    double* const park = tile + kW * p->RS + tid_now();  // parked shifts: [9][1024 threads]
    const int RS = p->RS;
    fill_rcp_table(rcp);

    const int wave = __builtin_amdgcn_readfirstlane(tid_now() / kWave);
    const int seg = wave >> 1;  // cell of the tile this wave works for
#define SD_LANE() const int lane = tid_now() % kWave; const int sl = ((wave & 1) << 6) + lane
    double* const row = tile + seg * RS;
    int* const xch = xch_all + seg * kSegLanes;
    const bool vec_f = (p->ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->y) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(p->X) & 15) == 0);
    const bool vec_p = (p->ld_p % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->Xp) & 15) == 0);
    const bool vec_o = (p->ld_out % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->out) & 15) == 0);

    const int xcd = (int)(blockIdx.x & 7u);
    const int64_t tx = (p->ntiles + 7) / 8;
    const int nslots = p->gmask ? __builtin_popcountll(p->gmask) : p->G;
    const int64_t nq = tx * nslots;                  // items of one XCD
    const int64_t qstep = (int64_t)(gridDim.x >> 3);  // workgroups per XCD
    int64_t q = (int64_t)(blockIdx.x >> 3);
    Item it = decode_item(p, q < nq ? q : 0, xcd, tx, nslots);
    while (q < nq && !it.valid) {
        q += qstep;
        it = decode_item(p, q < nq ? q : 0, xcd, tx, nslots);
    }
    if (q >= nq) return;

    // The x_hist rows of an item are requested (and reduced to x_climo) during the previous item's first sort, its
    // x_fut rows during the previous item's second sort: xc_cur / xf travel around the loop.
    // Bank 0: the y_obs tile of the current item, then the x_fut tile of the next one; bank 1: x_hist of the next item.
    double xc_cur = 0.0;
    if (threadIdx.x < kW) bad_cell[threadIdx.x] = 0;
    __syncthreads();
    {
        int th[5], tf[5];
        bank_rows(it.n > 0 ? p->ord_f + it.begf : p->ord_f, it.n > 0 ? it.n : 1, th);
        bank_rows(p->ord_p + it.begp, it.m, tf);
        if (it.n > 0) bank_issue<1>(p->X, p->ld, th, it.tile * kW, p->C);
        bank_issue<0>(p->Xp, p->ld_p, tf, it.tile * kW, p->C);
        const int lane0 = tid_now() % kWave;
        if (it.n > 0) {
            TileRegs<NR> xh;
            bank_take<1>(xh);
            xc_cur = reduce_mean2<NR>(xh, it.n, it.tile * kW, p->C, sums, p->status_fit, wave, lane0, seg, bad_cell);
        }
    }

    int nitem = 0;
    while (true) {  // one (tile, group) item per turn; workgroup-uniform control flow
        SD_STAMP(0);
        const int64_t c0 = it.tile * kW;
        const int64_t c = c0 + seg;
        const bool cell_ok = c < p->C;
        const int n = it.n, m = it.m, begf = it.begf, begp = it.begp, g = it.g;
        const bool cell_live = cell_ok && p->status_fit[cell_ok ? c : 0] == 0;
        __syncthreads();  // previous item: rows read by its stores

        // ---- x climatology (bcsd.py:222, reduced one item ahead), x_fut tile into the zero-padded rows -----------
        const double xc = xc_cur;
        {
            SD_LANE();
            TileRegs<NR> xf;
            bank_take<0>(xf);
            commit2<NR>(xf, m, c0, p->C, tile + kPadFront, RS, p->status_p, bad_cell);
            if (sl < kPadFront) row[sl] = 0.0;
            if (sl < CH + 4) row[kPadFront + m + sl] = 0.0;
        }
        __syncthreads();
        SD_STAMP(1);

        // ---- the y_obs tile is requested now and committed when the rows are free again -------------------------
        int64_t qn = q + qstep;
        Item nx = decode_item(p, qn < nq ? qn : 0, xcd, tx, nslots);
        while (qn < nq && !nx.valid) {
            qn += qstep;
            nx = decode_item(p, qn < nq ? qn : 0, xcd, tx, nslots);
        }
        const bool more = qn < nq;
        {
            int ty[5], th[5];
            bank_rows(n > 0 ? p->ord_f + begf : p->ord_f, n > 0 ? n : 1, ty);
            bank_rows(more && nx.n > 0 ? p->ord_f + nx.begf : p->ord_f, more && nx.n > 0 ? nx.n : 1, th);
            if (n > 0) bank_issue<0>(p->y, p->ld, ty, c0, p->C);
            if (more && nx.n > 0) bank_issue<1>(p->X, p->ld, th, nx.tile * kW, p->C);  // x_hist of the next item
        }

        // ---- shifted series (bcsd.py:247-256), tagged with the time position; the shift stays in registers ------
        const int np = (m + K - 1) / K * K;
        unsigned pos2[NP2];
        bool redo = false;
        {
            SD_LANE();
            double u[K];
#pragma unroll
            for (int cbeg = 0; cbeg < K; cbeg += CH) {
                double mean[CH], xv[CH];
                rolling_from_lds<CH>(row, K * sl + cbeg, m, rcp, mean, xv);
#pragma unroll
                for (int ii = 0; ii < CH; ++ii) {
                    const int i = cbeg + ii;
                    if (i < K) {
                        const double shift = mean[ii] - xc;        // bcsd.py:253
                        const double uv = (xv[ii] - shift) + 0.0;   // bcsd.py:256; -0.0 -> +0.0
                        shift_park(park, i, shift);
                        const int j = K * sl + i;
                        const unsigned tag = (unsigned)j * 8u;
                        const unsigned lo = ((unsigned)__double2loint(uv) & ~kTagMask) | tag;
                        const bool in = j < m;
                        u[i] = from_words(in ? lo : (((unsigned)j << 16) | tag), in ? __double2hiint(uv) : kPadHi);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();  // both waves of a segment have read their windows of the row
            SD_STAMP(2);
            sort_segment2<K, true, true>(u, row, m, sl, xch);
            SD_STAMP(3);
            if (np <= (K << 6)) __syncthreads();  // (the round across the waves ends with a barrier of its own)
            // ranks off the tags; two neighbouring sorted values with equal upper 48 bits make them ambiguous
            const bool owner = K * sl < np;
            unsigned amb = ~0u;
#pragma unroll
            for (int i = 0; i + 1 < K; ++i) {
                const unsigned dh = (unsigned)(__double2hiint(u[i]) ^ __double2hiint(u[i + 1]));
                const unsigned dl = (unsigned)(__double2loint(u[i]) ^ __double2loint(u[i + 1]));
                const unsigned d = dh | (dl & ~kTagMask);
                amb = d < amb ? d : amb;
            }
            {
                const double nx = row[K * (sl + 1) < np ? K * (sl + 1) : 0];  // first sorted value of the next lane
                const unsigned d = (unsigned)(__double2hiint(u[K - 1]) ^ __double2hiint(nx)) |
                                   ((unsigned)(__double2loint(u[K - 1]) ^ __double2loint(nx)) & ~kTagMask);
                redo = (K * (sl + 1) < np) && d == 0u;
            }
            redo |= amb == 0u;
#pragma unroll
            for (int i = 0; i < NP2; ++i) {
                const unsigned e = (unsigned)__double2loint(u[2 * i]) & kTagMask;
                const unsigned o = 2 * i + 1 < K ? (unsigned)__double2loint(u[2 * i + 1]) << 16 : 0u;
                pos2[i] = e | o;
            }
            redo |= __double2hiint(row[m - 1]) >= kPadHi;  // data in the pad range would sort behind pads
            redo = redo && owner && cell_live && bad_cell[seg] == 0;
        }
        const bool handed_back = __syncthreads_or(redo ? 1 : 0) != 0;  // also: every wave is done with its row
        if (handed_back && threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = it.tile * p->G + g;
        }
        // the flags of this item are consumed: those of the next item's x side start here
        if (threadIdx.x < kW) bad_cell[threadIdx.x] = 0;
        __syncthreads();
        double xc_next = 0.0;
        if (more && nx.n > 0) {
            const int lane1 = tid_now() % kWave;
            TileRegs<NR> xh;
            bank_take<1>(xh);
            xc_next = reduce_mean2<NR>(xh, nx.n, nx.tile * kW, p->C, sums, p->status_fit, wave, lane1, seg, bad_cell);
        }
        SD_STAMP(4);

        // ---- y: climatology (bcsd.py:223) + sort (quantile.py:462) -----------------------------------------------
        double yc = 0.0;
        double t[K];
        if (n > 0) {
            TileRegs<NR> yt;
            bank_take<0>(yt);
            commit2<NR>(yt, n, c0, p->C, tile, RS, p->status_fit, nullptr);
        }
        __syncthreads();
        SD_STAMP(5);

        // ---- the next item's x_fut tile is requested before the second sort ------------------------------------
        if (more) {
            int tf[5];
            bank_rows(p->ord_p + nx.begp, nx.m, tf);
            bank_issue<0>(p->Xp, p->ld_p, tf, nx.tile * kW, p->C);
        }

        if (!handed_back) {
            if (n > 0) {
                SD_LANE();
                double v[K];
                load_blocked<K>(row, n, sl, 0.0, v);
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < K; ++i) s += v[i];
                s = wave_sum(s);
                if (lane == 0) sums[wave] = s;
                __syncthreads();
                yc = (sums[wave & ~1] + sums[wave | 1]) / (double)n;  // bcsd.py:223
#pragma unroll
                for (int i = 0; i < K; ++i) v[i] = K * sl + i < n ? v[i] : __builtin_inf();
                wave_fence();  // (a lane reads and then overwrites its own slots only)
                SD_STAMP(6);
                sort_segment2<K, IDENT, !IDENT>(v, row, n, sl, xch);
                SD_STAMP(7);
                if (IDENT) {
#pragma unroll
                    for (int i = 0; i < K; ++i) t[i] = v[i];
                }
            }
            __syncthreads();

            // ---- ranks -> fitted inverse CDF (quantile.py:523-545), scattered to the time positions -------------
            {
                SD_LANE();
                const bool owner = K * sl < np;
                if (!IDENT) {
                    double slo = 0.0, ilo = 0.0, shi = 0.0, ihi = 0.0;
                    if (m > n && n > 0) {
                        const int e = n < 10 ? n : 10;
                        const double dn = pp_denom(n);
                        ols_line(row, 0, e, dn, &slo, &ilo);
                        ols_line(row, n - e, e, dn, &shi, &ihi);
                    }
                    const double nan = __longlong_as_double(0x7ff8000000000000ll);
                    const int r0 = owner ? K * sl : 0;
                    const int32_t* qi = p->qidx + begp + r0;
                    const double* qv = p->qval + begp + r0;
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        const bool in = r0 + i < m;
                        const int idx = in ? qi[i] : -3;
                        const double wq = in ? qv[i] : 0.0;
                        double v;
                        if (idx >= 0) {
                            const double y0 = row[idx];
                            const double y1 = row[idx + 1 < n ? idx + 1 : idx];
                            v = wq == 0.0 ? y0 : y0 + wq * (y1 - y0);
                        } else if (idx == -1) {
                            v = wq * slo + ilo;
                        } else if (idx == -2) {
                            v = wq * shi + ihi;
                        } else {
                            v = nan;
                        }
                        t[i] = v;
                    }
                    __syncthreads();  // every lane has read what it needs of the sorted rows
                }
                if (owner) {
                    const unsigned rowb = lds_addr(row);
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        const unsigned tag = (i & 1) ? (pos2[i >> 1] >> 16) : (pos2[i >> 1] & kTagMask);
                        lds_store_f64(rowb + tag, t[i]);
                    }
                }
            }
            __syncthreads();
            SD_STAMP(8);
            // ---- back in time order: restore the shift (bcsd.py:263), anomalies (bcsd.py:266-267) ---------------
            {
                SD_LANE();
                const int base = K * sl;
                double* trow = row + (base < np ? base : 0);
                double qv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) qv[i] = trow[i];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    double res = shift_take(park, i) + qv[i];
                    if (p->return_anoms) res = res - yc;
                    qv[i] = res;
                }
                if (base < np) {  // own slots only (slots m .. np-1 are never stored to the field)
#pragma unroll
                    for (int i = 0; i < K; ++i) trow[i] = qv[i];
                }
            }
            __syncthreads();
            SD_STAMP(9);
            store2(p->out, p->ld_out, p->ord_p + begp, m, c0, p->C, vec_o, tile, RS);
            SD_STAMP(10);
        }
        ++nitem;
        if (!more) break;
        q = qn;
        it = nx;
        xc_cur = xc_next;
    }
#undef SD_LANE
}

template <int K, bool IDENT, int NR>
int launch_kin(sd_ctx* ctx, const Params& p0, int nmax) {
    Params p = p0;
    int rs = (nmax + K - 1) / K * K + 1;           // the sort stores the pads of the last run; one readable slot past the end
    const int roll = kPadFront + nmax + 4 + 4;     // time-ordered segment with zero pads for the rolling windows (CH = 4)
    if (rs < roll) rs = roll;
    while (rs % 4 != 2) ++rs;                      // cell rows land 8 or 24 banks apart: conflict-free transposing stores
    p.RS = rs;
    const size_t lds = ((size_t)kW * rs + kHead2 + (size_t)kParkLds * kThreads2) * sizeof(double);
    SD_CHECK_ARG(lds <= ctx->lds_max || ctx->lds_max == 0, "segment too long for the two-wave kernel");
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fz2_kernel<K, IDENT, NR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t tx = (p.ntiles + 7) / 8;
    const int64_t items = tx * (p.gmask ? __builtin_popcountll(p.gmask) : p.G);  // per XCD
    int per_xcd = ctx->cu_count > 0 ? (ctx->cu_count + 7) / 8 : 32;            // one workgroup per CU
    if ((int64_t)per_xcd > items) per_xcd = (int)items;
    if (per_xcd < 1) per_xcd = 1;
    SD_LAUNCH(ctx, "bcsd_fz2_kernel", (bcsd_fz2_kernel<K, IDENT, NR>), dim3((unsigned)(8 * per_xcd)), dim3(kThreads2), lds, p);
#ifdef SD_DEV
    if (sd_dev_env("SD_FZ2_TRACE")) {  // mean phase durations (microseconds) of the first items of workgroup 0
        static long long h[64 * 16];
        SD_HIP(hipStreamSynchronize(ctx->stream));
        SD_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace), sizeof(h)));
        double acc[11] = {0};
        int cnt = 0;
        for (int i = 8; i < 56; ++i, ++cnt)
            for (int k = 0; k < 10; ++k) acc[k] += (double)(h[i * 16 + k + 1] - h[i * 16 + k]) * 0.01;
        for (int i = 8; i < 55; ++i) acc[10] += (double)(h[(i + 1) * 16] - h[i * 16]) * 0.01;
        fprintf(stderr, "fz2 trace (us):");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " p%d-%d %.2f", k, k + 1, acc[k] / cnt);
        fprintf(stderr, " | item %.2f\n", acc[10] / (cnt - 1));
    }
#endif
    return SD_OK;
}

template <int K, bool IDENT>
int launch_ki(sd_ctx* ctx, const Params& p, int nmax) {
    return launch_kin<K, IDENT, 5>(ctx, p, nmax);
}

}  // namespace sdfz2

bool sd_bcsd_fz2_supported(int nmax, const sdrs::Params& p) {
    const bool vec = p.C % 2 == 0 && p.ld % 2 == 0 && p.ld_p % 2 == 0 && p.ld_out % 2 == 0 &&
                     ((reinterpret_cast<uintptr_t>(p.X) | reinterpret_cast<uintptr_t>(p.y) | reinterpret_cast<uintptr_t>(p.Xp) |
                       reinterpret_cast<uintptr_t>(p.out)) & 15) == 0;
    return nmax >= 1 && nmax <= 5 * sdfz2::kRows2 && !p.from_state && p.X != nullptr && p.shift == nullptr && vec;
}

int sd_bcsd_fz2_launch(sd_ctx* ctx, const sdrs::Params& p, int nmax) {
    sdrs::Params q = p;
    q.gmask = 0ull;
    q.use_worklist = 0;
    return p.identity ? sdfz2::launch_ki<11, true>(ctx, q, nmax) : sdfz2::launch_ki<11, false>(ctx, q, nmax);
}
