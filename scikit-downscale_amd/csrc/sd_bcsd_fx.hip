// BcsdTemperature fit + predict of one (cell, month) segment in ONE workgroup pass -- the headline kernel, round 4.
//
// Same decomposition as sd_bcsd_rs.hip (one 64-lane wave per segment, 8 adjacent cells per 512-thread workgroup, K
// consecutive samples per lane, two workgroups per CU) and the same reference semantics as the fused kernel it replaces
// (bcsd.py:197-269, quantile.py:81-147, 438-545), but neither sort touches LDS memory any more:
//
//   keys     a sample becomes a 32-bit key: 21 bits of q = floor((v - lo) * QD / (hi - lo)) (lo, hi = extremes of the segment;
//            monotone in v) above the 11-bit LDS slot of the sample.  Pads get keys above every data key.
//   sort     sdws::wave_sort: a bitonic network over 64 lane-blocks held in registers (DPP / ds_swizzle partner fetch +
//            v_med3_u32, compile-time comparator networks inside a lane).  It orders by (q, slot), which differs from the
//            true order only inside runs of equal q (1.3 pairs per 1 240-sample segment).
//   fix-up   neighbours with equal q are compared by their float64 values, which sit in the wave's LDS row by slot, and
//            swap tags when inverted (wave-uniform branches; a second pass only when three or more samples share a q).
//            Exactly tied predict samples (np.interp wants the largest rank among ties, quantile.py:488) send the
//            (tile, group) to the RANK / APPLY work list as before.
//   x side   x_hist rows streamed and reduced to x_climo (bcsd.py:222); the x_fut tile transposed into the LDS rows; the
//            9-sample rolling mean gives the shift (bcsd.py:247-253), which now STAYS IN REGISTERS until the end (the sort
//            needs 20 key registers instead of 42 + 84 for values and merge windows): no second read of the x_fut tile, no
//            second rolling mean, no scratch.  The shifted series u = x - shift (bcsd.py:256) replaces x in the row.
//   y side   y_obs tile transposed into the same rows, y_climo (bcsd.py:223), keys, sort, fix-up; the lane owning sorted
//            positions r reads the r-th smallest observations from the row by tag (np.sort, quantile.py:462).
//   map      rank r -> fitted inverse CDF (quantile.py:523-545): identity for equal fit / predict group lengths, else the
//            per-rank (index, weight) table with 10-point OLS tails; the value is scattered to the time slot named by the
//            tag of the rank-r predict sample, every lane reads back its K consecutive samples, adds the shift
//            (bcsd.py:263), removes y_climo (bcsd.py:266-267) and the tile goes out transposed.
//
// LDS row layout: sample j of the segment sits in slot 4 + j + j / (K * P), P = 64 / gcd(2K, 64): every K*P samples one
// slot is skipped, so that the lanes' blocks of K doubles start on distinct bank pairs (K = 20 is even: stride 40 words
// would otherwise put lanes l and l + 8 on the same banks).  Slots 0..3 and the four slots behind the segment hold zeros
// for the rolling window.
// HBM traffic: 3 reads + 1 write per sample.
#include "sd_bcsd_rs.h"
#include "sd_wave.h"
#include "sd_wsort.h"

#ifdef SD_PHASE_MARKS
#define SDPH(name) asm volatile("; SDPHASE " name)
#else
#define SDPH(name)
#endif

namespace sdfx {

using namespace sdw;

// Development library: phase clocks (s_memtime) of sampled workgroups of the FULL temperature kernel (SD_FX_TRACE,
// tools/dev/trace_fx.py): [kTraceWgs][kTraceSlots], written by wave 0 of every kTraceStride-th workgroup.
#ifdef SD_DEV
constexpr int kTraceWgs = 256, kTraceSlots = 16, kTraceStride = 449;
__device__ long long sd_fx_trace[kTraceWgs * kTraceSlots];
#define SDT(slot)                                                                                              \
    do {                                                                                                       \
        if (traced) tstamp[slot] = (long long)__builtin_amdgcn_s_memtime();                                    \
    } while (0)
#else
#define SDT(slot)
#endif
using sdrs::Params;

typedef const Params __attribute__((address_space(4)))* ParamsPtr;

constexpr int kFront = 4;
constexpr unsigned kTagBits = 11, kTagMask = 2047u;
constexpr unsigned kQD = (1u << 21) - 2048u;  // data keys use q <= kQD; pads q = kQD + 1 + slot

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }

template <int K>
struct Lay {
    static constexpr int P = 64 / cgcd(2 * K, 64);
    static constexpr int KP = K * P;
    __device__ __host__ static int slot(int j) { return kFront + j + j / KP; }  // j >= 0
    __device__ static int own(int lane) { return kFront + K * lane + lane / P; }  // slot of sample K * lane
};

// slots a row needs: a partly filled lane reads its whole block and the rolling window behind it, so the row reaches
// sample K * ceil(len / K) + 3 of the longest group; one spare slot takes the stores of positions past the segment
template <int K>
int row_slots(int nmax) {
    int need = Lay<K>::slot((nmax + K - 1) / K * K + 3) + 2;
    while (need % 4 != 2) ++need;  // cell rows land 8 or 24 banks apart: conflict-free transposing stores
    return need;
}

__device__ __forceinline__ unsigned lds_u32(unsigned a) { return *reinterpret_cast<__attribute__((address_space(3))) unsigned*>((uintptr_t)a); }

// ---- wave reductions ----------------------------------------------------------------------------------------------
// In registers (DPP row shifts, then the two row broadcasts; the result is read off lane 63): six dependent steps of three
// instructions each instead of six LDS round trips of ~130 clocks -- these reductions sit on the critical path of a
// workgroup between two barriers.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v) {  // lanes without a source (or masked out) keep their own value
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
#define SD_DPP_REDUCE(OP)                        \
    v = OP(v, dpp_f64<0x111, 0xF>(v)); /* row_shr:1 */  \
    v = OP(v, dpp_f64<0x112, 0xF>(v)); /* row_shr:2 */  \
    v = OP(v, dpp_f64<0x114, 0xF>(v)); /* row_shr:4 */  \
    v = OP(v, dpp_f64<0x118, 0xF>(v)); /* row_shr:8 */  \
    v = OP(v, dpp_f64<0x142, 0xA>(v)); /* row_bcast15 -> rows 1, 3 */ \
    v = OP(v, dpp_f64<0x143, 0xC>(v)); /* row_bcast31 -> rows 2, 3 */ \
    return lane63_f64(v)
__device__ __forceinline__ double wave_min_f64(double v) { SD_DPP_REDUCE(vmin); }
__device__ __forceinline__ double wave_max_f64(double v) { SD_DPP_REDUCE(vmax); }
#undef SD_DPP_REDUCE
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp0_f64(double v) {  // lanes without a source (or masked out) read 0.0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {  // the same six steps; the sum of all 64 lanes, on every lane
    v += dpp0_f64<0x111, 0xF>(v);
    v += dpp0_f64<0x112, 0xF>(v);
    v += dpp0_f64<0x114, 0xF>(v);
    v += dpp0_f64<0x118, 0xF>(v);
    v += dpp0_f64<0x142, 0xA>(v);
    v += dpp0_f64<0x143, 0xC>(v);
    return lane63_f64(v);
}
// inclusive prefix sums over the 64 lanes (DPP row shifts, then the two row broadcasts)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp0_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, false); }
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += dpp0_i32<0x111, 0xF>(v);
    v += dpp0_i32<0x112, 0xF>(v);
    v += dpp0_i32<0x114, 0xF>(v);
    v += dpp0_i32<0x118, 0xF>(v);
    v += dpp0_i32<0x142, 0xA>(v);
    v += dpp0_i32<0x143, 0xC>(v);
    return v;
}

// the keys of the neighbouring lanes: wave shifts by one lane (lanes without a neighbour get `edge`)
__device__ __forceinline__ unsigned from_next_lane(unsigned v, unsigned edge) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xF, 0xF, false);  // wave_shl:1: lane i <- lane i + 1
}
__device__ __forceinline__ unsigned from_prev_lane(unsigned v, unsigned edge) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false);  // wave_shr:1: lane i <- lane i - 1
}

// ---- tile movement with the swizzled row layout ---------------------------------------------------------------
// one v_cmp_class_f64 per value (NaN or +-inf) instead of the and + compare of finite64
__device__ __forceinline__ bool nonfinite64(double v) { return __builtin_amdgcn_class(v, 0x207); }

// FULL (template parameter of the kernels): every segment the launch serves has a multiple of K samples -- a lane is all
// data or all pad -- and at least kRowsPerPass * (K / 2 - 1) + 1 of them -- every thread's first K / 2 - 1 rows of a tile
// exist, only the last pass is predicated.  10 of the 12 months of a daily series qualify at K = 20 (1 200 / 1 240 samples).
template <int K>
constexpr int full_min_len() { return kRowsPerPass * (K / 2 - 1) + 1; }

// slot of row r (< 2 048) without the division by K * P: floor(r / d) = (r * ceil(2^21 / d)) >> 21 for the divisors in use
// (checked exhaustively by the static_assert below); one 24-bit multiply, one shift
template <int K>
struct SlotDiv {
    static constexpr unsigned D = (unsigned)Lay<K>::KP;
    static constexpr unsigned M = ((1u << 21) + D - 1) / D;
    static constexpr bool ok() {
        for (unsigned r = 0; r < 2048; ++r)
            if (((r * M) >> 21) != r / D) return false;
        return true;
    }
    __device__ static int slot(int r) { return kFront + r + (int)(__umul24((unsigned)r, M) >> 21); }
};

// The time indices of a thread's rows (ord[rr + k * kRowsPerPass]) loaded on their own: the tile loads depend on them, so
// they are requested as early as registers allow (a table of 64-bit row byte offsets instead -- one add per row -- was
// measured 5 % slower: twice the registers, the compiler no longer hoisted the table loads of the second tile above the
// loads of the first).
template <int RPT, bool FULL>
__device__ __forceinline__ void rows_load(const int32_t* __restrict__ ord, int nrows, int (&ti)[RPT]) {
    const int rr = tid_now() >> 2;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRowsPerPass;
        ti[k] = ord[(FULL && k + 1 < RPT) || r < nrows ? r : 0];
    }
}
template <int RPT>
__device__ __forceinline__ void tile_issue_ti(const double* __restrict__ src, int64_t ld, const int (&ti)[RPT], int64_t c0, int64_t C,
                                              bool vec_ok, TileRegs<RPT>& t) {
    const int cp = tid_now() & 3;
    const int64_t c = c0 + 2 * cp;
    const bool full = vec_ok && c + 1 < C;
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
            const double* q = row_of(src + c, ti[k], ld);
            t.v0[k] = c < C ? q[0] : 0.0;
            t.v1[k] = c + 1 < C ? q[1] : 0.0;
        }
    }
}
// the tile out through row indices loaded earlier
template <int K, bool FULL>
__device__ __forceinline__ void store_tile_ti(double* __restrict__ dst, int64_t ld, const int (&ti)[K / 2], int nrows, int64_t c0, int64_t C,
                                              bool vec_ok, const double* tile, int RS) {
    constexpr int RPT = K / 2;
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    const double* s0 = tile + (2 * cp) * RS;
    const double* s1 = s0 + RS;
    const bool full = vec_ok && c + 1 < C;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRowsPerPass;
        if ((FULL && k + 1 < RPT) || r < nrows) {
            double* q = row_of(dst + c, ti[k], ld);
            const int s = SlotDiv<K>::slot(r);
            if (full) {
                *reinterpret_cast<double2*>(q) = make_double2(s0[s], s1[s]);
            } else {
                if (c < C) q[0] = s0[s];
                if (c + 1 < C) q[1] = s1[s];
            }
        }
    }
}

template <int RPT, int K, bool FULL = false>
__device__ __forceinline__ void tile_commit_sw(const TileRegs<RPT>& t, int nrows, int64_t c0, int64_t C, double* tile, int RS,
                                               int32_t* status, int* bad_cell) {
    static_assert(SlotDiv<K>::ok(), "slot multiplier");
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    double* d0 = tile + (2 * cp) * RS;
    double* d1 = d0 + RS;
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rr + k * kRowsPerPass;
        // (rows past the segment were loaded from row 0 of the segment: testing them too flags nothing new)
        bad0 |= nonfinite64(t.v0[k]);
        bad1 |= nonfinite64(t.v1[k]);
        if ((FULL && k + 1 < RPT) || r < nrows) {
            const int s = SlotDiv<K>::slot(r);
            d0[s] = t.v0[k];
            d1[s] = t.v1[k];
        }
    }
    if (bad0 && c < C) atomicOr(&status[c], SDI_NONFINITE);
    if (bad1 && c + 1 < C) atomicOr(&status[c + 1], SDI_NONFINITE);
    if (bad_cell != nullptr) {
        if (bad0) bad_cell[2 * cp] = 1;
        if (bad1) bad_cell[2 * cp + 1] = 1;
    }
}

template <int K>
__device__ __forceinline__ void store_tile_sw(double* __restrict__ dst, int64_t ld, const int32_t* __restrict__ ord, int nrows,
                                              int64_t c0, int64_t C, bool vec_ok, const double* tile, int RS) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    const double* s0 = tile + (2 * cp) * RS;
    const double* s1 = s0 + RS;
    const bool full = vec_ok && c + 1 < C;
#pragma unroll 4
    for (int r = rr; r < nrows; r += kRowsPerPass) {
        double* p = row_of(dst + c, ord[r], ld);
        const int s = Lay<K>::slot(r);
        if (full) {
            *reinterpret_cast<double2*>(p) = make_double2(s0[s], s1[s]);
        } else {
            if (c < C) p[0] = s0[s];
            if (c + 1 < C) p[1] = s1[s];
        }
    }
}
// column sums of one group's rows for the 8 cells of the tile from issued tile registers, as far as one wave gets: the wave's
// partial sums go to scratch[wave][cell]; the caller adds the 8 partials of its cell behind its next barrier (sd_wave.h's
// tile_reduce_mean does the same with two barriers of its own)
template <int RPT, bool FULL = false>
__device__ __forceinline__ void tile_reduce_partials(const TileRegs<RPT>& t, int nrows, int64_t c0, int64_t C, double* scratch,
                                                     int32_t* status, int wave, int lane, int* bad_cell) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    double s0 = 0.0, s1 = 0.0;
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        // (rows past the segment were loaded from row 0 of the segment: testing them too flags nothing new)
        bad0 |= nonfinite64(t.v0[k]);
        bad1 |= nonfinite64(t.v1[k]);
        if (FULL && k + 1 < RPT) {
            s0 += t.v0[k];
            s1 += t.v1[k];
        } else {
            const bool in = rr + k * kRowsPerPass < nrows;
            s0 += in ? t.v0[k] : 0.0;
            s1 += in ? t.v1[k] : 0.0;
        }
    }
    if (bad0 && c < C) atomicOr(&status[c], SDI_NONFINITE);
    if (bad1 && c + 1 < C) atomicOr(&status[c + 1], SDI_NONFINITE);
    if (bad0) bad_cell[2 * cp] = 1;
    if (bad1) bad_cell[2 * cp + 1] = 1;
    // lanes with equal (lane & 3) hold the same cell pair: lanes l, l + 4, l + 8, l + 12 of a row by rotations of the row,
    // the four rows through the LDS crossbar
    s0 += dpp_f64<0x124, 0xF>(s0);  // row_ror:4
    s1 += dpp_f64<0x124, 0xF>(s1);
    s0 += dpp_f64<0x128, 0xF>(s0);  // row_ror:8
    s1 += dpp_f64<0x128, 0xF>(s1);
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        s0 += __shfl_xor(s0, o, kWave);
        s1 += __shfl_xor(s1, o, kWave);
    }
    if (lane < 4) {
        scratch[wave * kW + 2 * lane] = s0;
        scratch[wave * kW + 2 * lane + 1] = s1;
    }
}

// non-finite samples of an issued tile -> per-cell status (BcsdPrecipitation only validates x_hist: bcsd.py:130-147): no sums,
// no exchange, no barrier of its own -- the flags are read behind the barrier that follows the x_fut commit
template <int RPT>
__device__ __forceinline__ void tile_check_finite(const TileRegs<RPT>& t, int nrows, int64_t c0, int64_t C, int32_t* status, int* bad_cell) {
    const int tid = tid_now();
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t c = c0 + 2 * cp;
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool in = rr + k * kRowsPerPass < nrows;
        bad0 |= in && !finite64(t.v0[k]);
        bad1 |= in && !finite64(t.v1[k]);
    }
    if (bad0 && c < C) atomicOr(&status[c], SDI_NONFINITE);
    if (bad1 && c + 1 < C) atomicOr(&status[c + 1], SDI_NONFINITE);
    if (bad0) bad_cell[2 * cp] = 1;
    if (bad1) bad_cell[2 * cp + 1] = 1;
}

// ---- keys ---------------------------------------------------------------------------------------------------------
// v[i] = sample K * lane + i of the segment (any value where that position is past m); key = (q << 11) | slot
// ZC (BcsdPrecipitation): exact zeros form class q = 0 of their own -- they are all tied, whatever order the sort leaves
// them in -- and every other sample gets q >= 1 (lo must not be negative: the caller hands such segments back).
// FULL: the segment length is a multiple of K, so a lane is all data or all pad: one select per key instead of a compare
// and a select, and no masking in the search for the extremes (lanes past the segment hold a copy of lane 0's block).
// PLAIN: the samples sit in the row without the skipped slots (compacted wet days: slot = sample index).
template <int K, bool ZC, bool FULL, bool PLAIN = false>
__device__ __forceinline__ void keys_from_range(const double (&v)[K], int m, int lane, double lo, double hi, unsigned (&key)[K]) {
    const int j0 = K * lane;
    const double sc = (double)(kQD - (ZC ? 1u : 0u)) / (hi - lo);  // +inf when every sample is equal: all keys tie, the fix-up sorts it out
    const double off = -lo * sc + (ZC ? 1.0 : 0.0);
    const unsigned tag0 = PLAIN ? (unsigned)j0 : (unsigned)Lay<K>::own(lane);
    const unsigned pad0 = ((kQD + 1u + tag0) << kTagBits) | tag0;
    const bool lane_in = j0 < m;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        unsigned q = (unsigned)__builtin_fma(v[i], sc, off);  // v_cvt_u32_f64: truncates, saturates, NaN -> 0
        q = q < kQD ? q : kQD;
        if (ZC) q = v[i] == 0.0 ? 0u : (q > 1u ? q : 1u);
        const unsigned dk = (q << kTagBits) | (tag0 + (unsigned)i);
        const unsigned pk = pad0 + (unsigned)i * ((1u << kTagBits) + 1u);
        key[i] = (FULL ? lane_in : j0 + i < m) ? dk : pk;
    }
}
template <int K, bool ZC, bool FULL, bool PLAIN = false>
__device__ __forceinline__ double make_keys_impl(const double (&v)[K], int m, int lane, unsigned (&key)[K]) {
    const int j0 = K * lane;
    double lo = __builtin_inf(), hi = -__builtin_inf();
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const bool in = FULL || j0 + i < m;
        lo = vmin(lo, in ? v[i] : lo);
        hi = vmax(hi, in ? v[i] : hi);
    }
    lo = wave_min_f64(lo);
    keys_from_range<K, ZC, FULL, PLAIN>(v, m, lane, lo, wave_max_f64(hi), key);
    return lo;
}
// keys of a compacted series of m samples, K consecutive ones per lane at row[K * lane ..] (tags = sample indices)
template <int K>
__device__ __forceinline__ double make_keys_plain(const double (&v)[K], int m, int lane, unsigned (&key)[K]) {
    return make_keys_impl<K, false, false, true>(v, m, lane, key);
}
template <int K, bool ZC = false, bool FULL = false>
__device__ __forceinline__ double make_keys(const double (&v)[K], int m, int lane, unsigned (&key)[K]) {
    if (FULL || m % K == 0) return make_keys_impl<K, ZC, true>(v, m, lane, key);  // wave-uniform (the whole launch, in fact)
    return make_keys_impl<K, ZC, false>(v, m, lane, key);
}

// ---- exact order inside runs of equal q -------------------------------------------------------------------------------
// k[] = the lane's K sorted keys (lane l owns sorted positions K*l ..).  Neighbouring keys with equal q are compared by
// the float64 values behind their tags (rowb + 8 * slot) and exchange tags when inverted.  Returns kTie if two of the
// compared values are equal (an exact tie), kUnsorted if the passes did not converge (a long run of equal q).
constexpr int kTie = 1, kUnsorted = 2;
template <int K, bool ZC = false>
__device__ __forceinline__ int fix_equal_q(unsigned (&k)[K], unsigned rowb, int lane) {
    constexpr unsigned kQ = 1u << kTagBits;  // (ZC: keys below kQ are the zero class, all exactly tied: left alone)
    bool tie = false;
#pragma unroll 1
    for (int pass = 0; pass < 6; ++pass) {
        bool swapped = false;
        // the pair (last key of lane l, first key of lane l + 1) first, on both lanes' values as the pass finds them
        const unsigned knext = from_next_lane(k[0], 0xffffffffu);
        const bool eqx = (k[K - 1] ^ knext) < kQ && (!ZC || k[K - 1] >= kQ);
        const unsigned long long bx = __ballot(eqx);
        if (bx != 0ull) {  // wave-uniform, rare
            const double va = lds_f64(rowb + 8u * (k[K - 1] & kTagMask));
            const double vb = lds_f64(rowb + 8u * (knext & kTagMask));
            const bool sw = eqx && va > vb;
            tie |= eqx && va == vb;
            const unsigned d = sw ? (k[K - 1] ^ knext) : 0u;  // equal q: the xor exchanges the tags
            k[K - 1] ^= d;
            k[0] ^= from_prev_lane(d, 0u);
            swapped |= sw;
        }
        // pairs inside the lane, in ascending order
        unsigned long long prev = 0ull, b0 = 0ull, multi = 0ull;
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) {
            const bool eq = (k[i] ^ k[i + 1]) < kQ && (!ZC || k[i] >= kQ);
            const unsigned long long b = __ballot(eq);
            multi |= b & prev;
            if (i == 0) b0 = b;
            prev = b;
            if (b != 0ull) {
                const double va = lds_f64(rowb + 8u * (k[i] & kTagMask));
                const double vb = lds_f64(rowb + 8u * (k[i + 1] & kTagMask));
                const bool sw = eq && va > vb;
                tie |= eq && va == vb;
                const unsigned d = sw ? (k[i] ^ k[i + 1]) : 0u;
                k[i] ^= d;
                k[i + 1] ^= d;
                swapped |= sw;
            }
        }
        // a key in two equal-q pairs (a run of three or more) may need another pass
        multi |= (bx & prev) | ((bx << 1) & b0);
        if (multi == 0ull || !__any(swapped)) return __any(tie) ? kTie : 0;
    }
    return kUnsorted;
}

template <int K, bool IDENT, bool FULL>
__global__ void __launch_bounds__(kThreads, 4) bcsd_fx_kernel(const Params) {
    static_assert(!FULL || IDENT, "FULL launches serve groups of equal fit / predict length");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();  // Params is the only kernel argument
    constexpr int NR = K / 2;  // rows per thread of a tile (64 K rows, 128 per pass)
    constexpr int CH = K >= 20 ? K / 4 : K >= 14 ? K / 2 : K;  // samples per rolling-mean chunk (bounded register pressure)
    static_assert(K % 2 == 0 && K % CH == 0, "even K");
    using L = Lay<K>;
    // the row indices of the y tile and of the output tile are requested ahead of the barriers in front of their use (10
    // registers each, for which only the FULL instantiation has room)
#ifdef SD_FX_LATEROWS
    constexpr bool kEarlyRows = false;
#else
    constexpr bool kEarlyRows = FULL;
#endif
#ifdef SD_DEV
    const int abl = p->dev_flags;  // SD_FZ_ABLATE: 1 no sort of u, 2 no sort of y, 4 no x_hist, 8 no fix-ups, 16 no store, 32 no y load,
                                   // 64 no x_fut load, 128 no rolling mean (timing only: results are wrong)
#else
    constexpr int abl = 0;
#endif
    double* const scratch = reinterpret_cast<double*>(smem_raw);  // 64 doubles (column-sum exchange)
    double* const rcp = scratch + 64;                             // 16 doubles: correctly rounded 1/c, c = 1..9
    int* const bad_cell = reinterpret_cast<int*>(rcp + 16);       // 8 ints: cell of the tile saw a non-finite sample
    int* const redo_flag = bad_cell + kW;                         // 8 ints: the wave's segment must go to the work list
    double* const tile = scratch + kHeadDoubles;
    const int RS = p->RS;
    fill_rcp_table(rcp);
    if (threadIdx.x >= 32 && threadIdx.x < 32 + kW) bad_cell[threadIdx.x - 32] = 0;

#ifdef SD_DEV
    const bool traced = FULL && (abl & 0x800) != 0 && blockIdx.x % kTraceStride == 7 && blockIdx.x / kTraceStride < (unsigned)kTraceWgs;  // (uniform)
    long long tstamp[kTraceSlots] = {};
    SDT(0);
    if ((abl >> 12) != 0 && blockIdx.x < 512u) {
        // SD_FZ_ABLATE bits 12.. = T: the first generation of workgroups starts spread over ~T microseconds
        const unsigned h = (blockIdx.x * 2654435761u) >> 20;  // 12 bits
        const unsigned nn = (h * (unsigned)(abl >> 12)) >> 12;  // 0 .. T-1 "microseconds"
        for (unsigned i = 0; i < 2u * nn; ++i) __builtin_amdgcn_s_sleep(19);  // ~1216 clocks ~ 0.5 us
    }
#endif
    int64_t tile_id;
    int g;
    xcd_tile_of_block(blockIdx.x, p->ntiles, &tile_id, &g);
    if (p->gmask != 0ull) g = nth_set_bit(p->gmask, g);
    if (tile_id >= p->ntiles || g < 0 || g >= p->G) return;

    const int64_t c0 = tile_id * kW;
    const int wave = __builtin_amdgcn_readfirstlane(tid_now() / kWave);
#define SD_LANE() const int lane = tid_now() % kWave
    const int64_t c = c0 + wave;
    const bool cell_ok = c < p->C;
    double* const row = tile + wave * RS;
    const unsigned rowb = lds_addr(row);
    const int64_t seg = c * p->G + g;
    const int begf = p->off_f[g];
    const int n = p->off_f[g + 1] - begf;
    const int begp = p->off_p[g];
    const int m = p->off_p[g + 1] - begp;
    if (m == 0) return;
    const bool vec_f = (p->ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->y) & 15) == 0) &&
                       (p->X == nullptr || (reinterpret_cast<uintptr_t>(p->X) & 15) == 0);
    const bool vec_p = (p->ld_p % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->Xp) & 15) == 0);
    const bool cell_live = cell_ok && p->status_fit[cell_ok ? c : 0] == 0;
    const unsigned spare = rowb + 8u * (unsigned)(RS - 1);  // slot that takes the stores of positions past the segment
    __syncthreads();  // bad_cell zeroed before the commits below may set it

    // ---- x climatology (bcsd.py:222) + the x_fut tile ---------------------------------------------------
    SDPH("x_tiles");
    double xc = 0.0;
    bool xc_from_partials = false;  // (workgroup-uniform)
    {
        SD_LANE();
        TileRegs<NR> xf;
        int tp[NR];
        rows_load<NR, FULL>(p->ord_p + begp, m, tp);
        if (p->from_state) {
            if (cell_ok) xc = p->x_climo[seg];
            tile_issue_ti<NR>(p->Xp, p->ld_p, tp, c0, p->C, vec_p, xf);
        } else if (n > 0 && !(abl & 4)) {
            TileRegs<NR> xh;
            int tf[NR];
            rows_load<NR, FULL>(p->ord_f + begf, n, tf);
            tile_issue_ti<NR>(p->X, p->ld, tf, c0, p->C, vec_f, xh);
            if (!(abl & 64)) tile_issue_ti<NR>(p->Xp, p->ld_p, tp, c0, p->C, vec_p, xf);
            tile_reduce_partials<NR, FULL>(xh, n, c0, p->C, scratch, p->status_fit, wave, lane, bad_cell);
            xc_from_partials = true;
        } else if (!(abl & 64)) {
            tile_issue_ti<NR>(p->Xp, p->ld_p, tp, c0, p->C, vec_p, xf);
        }
        SDT(1);  // x tiles requested, column sums reduced
        if (!(abl & 64)) tile_commit_sw<NR, K, FULL>(xf, m, c0, p->C, tile, RS, p->status_p, bad_cell);
        if (lane < kFront) {
            row[lane] = 0.0;
            row[L::slot(m + lane)] = 0.0;
        }
    }
    __syncthreads();
    SDT(2);  // x_fut tile in the rows
    if (xc_from_partials) {  // x_climo (bcsd.py:222): the waves' partial column sums, in wave order
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kW; ++w) tot += scratch[w * kW + wave];  // wave <-> cell c0 + wave
        xc = tot / (double)n;
    }

    // ---- shift (kept), shifted series -> row, keys -> sort ------------------------------------------------
    SDPH("rolling");
    double shift[K];
    unsigned ku[K];  // sorted keys of the shifted series: tag = time slot of the sample with that rank
    bool redo = false;
    int ty[NR];  // time indices of this thread's rows of the y tile
    {
        SD_LANE();
        const bool has = K * lane < m;  // the lane's block starts inside the segment
        const int bl = has ? lane : 0;  // lanes past the segment read lane 0's block (values unused)
        const bool first_lane = bl == 0, last_lane = K * (bl + 1) == m;  // (FULL: the only lanes with clipped windows)
        const double* ob = row + L::own(bl);
        const double* pb = row + (kFront + K * bl + (bl > 0 ? (bl - 1) / L::P : 0));  // pb[-k] = sample K*bl - k
        const double* nb = row + (kFront + K * (bl + 1) + (bl + 1) / L::P);              // nb[k]  = sample K*(bl+1) + k
        double u[K];
#pragma unroll
        for (int cbeg = 0; cbeg < K; cbeg += CH) {
            double w[CH + 8];
#pragma unroll
            for (int t = 0; t < CH + 8; ++t) {
                const int idx = cbeg - 4 + t;  // sample K*bl + idx
                w[t] = idx < 0 ? pb[idx] : idx >= K ? nb[idx - K] : ob[idx];
            }
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                // the nine samples are added in time order, like the oracle and the RANK kernel do: samples whose shifted values
                // are one ulp apart then rank the same way on every path (sharing partial sums between windows would save four
                // additions per sample and decide such pairs differently)
                double s = 0.0;
#pragma unroll
                for (int d = 0; d < 9; ++d) s += w[ii + d];
                double cd, rc;
                if constexpr (FULL) {
                    // the window is clipped for the first four samples of the first lane and the last four of the last
                    // lane only (a lane is all data, the segment has more than one lane): constants elsewhere
                    const int i = cbeg + ii;  // (a constant once the loops are unrolled)
                    constexpr double kRc[10] = {0.0, 1.0, 0.5, 1.0 / 3.0, 0.25, 0.2, 1.0 / 6.0, 1.0 / 7.0, 0.125, 1.0 / 9.0};  // = fill_rcp_table
                    cd = 9.0;
                    rc = kRc[9];
                    if (i < 4) {
                        cd = first_lane ? (double)(5 + i) : cd;
                        rc = first_lane ? kRc[5 + i] : rc;
                    }
                    if (i >= K - 4) {  // (K = 4: every sample is in both classes, a lane in at most one)
                        cd = last_lane ? (double)(K + 4 - i) : cd;
                        rc = last_lane ? kRc[K + 4 - i] : rc;
                    }
                } else {
                    const int j = K * bl + cbeg + ii;  // (lanes past the segment redo lane 0's samples: in-range values for the extremes)
                    const int lo = j - 4 > 0 ? j - 4 : 0;
                    const int hi = j + 5 < m ? j + 5 : m;
                    const int cnt = hi - lo > 1 ? (hi - lo < 10 ? hi - lo : 9) : 1;
                    cd = (double)cnt;
                    rc = rcp[cnt];
                }
                const double q = s * rc;
                const double mean = __builtin_fma(__builtin_fma(-cd, q, s), rc, q);  // correctly rounded s / cnt
                const double sh = mean - xc;                                           // bcsd.py:253
                shift[cbeg + ii] = sh;
                u[cbeg + ii] = (w[ii + 4] - sh) + 0.0;  // bcsd.py:256; -0.0 -> +0.0 (they tie in np.sort / np.interp)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wave_fence();  // every lane has read its window
        SDT(3);  // rolling mean
        SDPH("u_store_keys");
        {
            const unsigned a0 = rowb + 8u * (unsigned)L::own(lane);
            if constexpr (FULL) {
                if (has) {
#pragma unroll
                    for (int i = 0; i < K; ++i) lds_store_f64(a0 + 8u * (unsigned)i, u[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) lds_store_f64(K * lane + i < m ? a0 + 8u * (unsigned)i : spare, u[i]);
            }
        }
        make_keys<K, false, FULL>(u, m, lane, ku);
        wave_fence();
        SDT(4);  // u stored, keys
        SDPH("u_sort");
        if (!(abl & 1)) sdws::wave_sort<K>(ku, lane, (m + K - 1) / K);
        SDT(5);  // sort of u
        SDPH("u_fix");
        // the rows of the y tile are named while the fix-up runs: the tile loads behind the vote go out at once
        if (kEarlyRows && !p->from_state && n > 0) rows_load<NR, FULL>(p->ord_f + begf, n, ty);
        const bool tie = (abl & 8) ? false : fix_equal_q<K>(ku, rowb, lane) != 0;
        redo = tie && cell_live && bad_cell[wave] == 0 && (abl & 0x7ff) == 0;  // wave-uniform
        SDT(6);  // fix-up of u
    }
    SDPH("vote");
    // every wave is done with its row; a workgroup with an ambiguous segment leaves the (tile, group) to RANK / APPLY
    // (not __syncthreads_or: its library reduction takes 256 bytes of static LDS, which costs the second workgroup per CU)
    redo_flag[wave] = redo ? 1 : 0;
    __syncthreads();
    int any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
        }
        return;
    }
    SDT(7);  // vote

    // ---- y: climatology + sorted observations -----------------------------------------------------------------
    SDPH("y_tile");
    double yc = 0.0;
    double t[K];  // IDENT: the sorted observations of the ranks this lane owns
    bool redo_y = false;  // a run of equal-q observations too long for the fix-up passes: RANK / APPLY take the (tile, group)
    if (!p->from_state) {
        if (n > 0) {
            SD_LANE();
            // (requesting this tile ahead of the sort of u, of its fix-up or of the vote keeps 40 registers in flight where the
            // compiler has none to spare: 28 - 30 spilled registers, measured slower)
            TileRegs<NR> yt;
            if (!kEarlyRows) rows_load<NR, FULL>(p->ord_f + begf, n, ty);
            if (!(abl & 32)) tile_issue_ti<NR>(p->y, p->ld, ty, c0, p->C, vec_f, yt);
            SDT(8);  // y tile requested
            if (!(abl & 32)) tile_commit_sw<NR, K, FULL>(yt, n, c0, p->C, tile, RS, p->status_fit, nullptr);
            __syncthreads();
            SDT(9);  // y tile in the rows
            SDPH("y_keys");
            unsigned ky[K];
            {
                const int bl = K * lane < n ? lane : 0;
                const double* ob = row + L::own(bl);
                double v[K];
#pragma unroll
                for (int i = 0; i < K; ++i) v[i] = ob[i];
                double s = 0.0;
                if constexpr (FULL) {
#pragma unroll
                    for (int i = 0; i < K; ++i) s += v[i];
                    s = K * lane < n ? s : 0.0;
                } else {
#pragma unroll
                    for (int i = 0; i < K; ++i) s += K * lane + i < n ? v[i] : 0.0;
                }
                yc = wave_sum_f64(s) / (double)n;  // bcsd.py:223
                make_keys<K, false, FULL>(v, n, lane, ky);
            }
            SDT(10);  // y_climo, keys
            SDPH("y_sort");
            if (!(abl & 2)) sdws::wave_sort<K>(ky, lane, (n + K - 1) / K);
            SDT(11);  // sort of y
            SDPH("y_fix");
            if (!(abl & 8)) redo_y = (fix_equal_q<K>(ky, rowb, lane) & kUnsorted) != 0 && cell_live;  // tied observations are interchangeable
            SDPH("y_gather");
            if constexpr (FULL) {  // (the pad keys of the lanes past the segment carry slots beyond the row: those lanes read slot 0)
                const unsigned tm = K * lane < n ? kTagMask : 0u;
#pragma unroll
                for (int i = 0; i < K; ++i) t[i] = lds_f64(rowb + 8u * (ky[i] & tm));
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) t[i] = lds_f64(rowb + 8u * (K * lane + i < n ? (ky[i] & kTagMask) : (unsigned)(RS - 1)));
            }
            if (!IDENT) {
                wave_fence();  // all reads by tag done: the row becomes the sorted segment, plain indexing
#pragma unroll
                for (int i = 0; i < K; ++i) lds_store_f64(K * lane + i < n ? rowb + 8u * (unsigned)(K * lane + i) : spare, t[i]);
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
            const double* srow = row + (K * lane < n ? K * lane : 0);
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = srow[i];
        }
    }

    // ---- map ranks through the fitted inverse CDF (quantile.py:523-545), scatter to time slots ------------------
    SDT(12);  // fix-up of y, sorted observations gathered
    SDPH("map_scatter");
    int to[NR];  // time indices of this thread's rows of the output tile
    {
        SD_LANE();
        if (!IDENT) {
            wave_fence();
            double slo = 0.0, ilo = 0.0, shi = 0.0, ihi = 0.0;
            if (m > n && n > 0) {  // tails are reachable only when the predict segment is longer
                const int e = n < p->n_endpoints ? n : p->n_endpoints;
                const double dn = pp_denom(n);
                ols_line(row, 0, e, dn, &slo, &ilo);
                ols_line(row, n - e, e, dn, &shi, &ihi);
            }
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
            const int r0 = K * lane < m ? K * lane : 0;
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
                if ((i + 1) % CH == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
        wave_fence();  // every lane has read what it needs of the row
        const bool has = K * lane < m;
        if constexpr (FULL) {
            if (has) {
#pragma unroll
                for (int i = 0; i < K; ++i) lds_store_f64(rowb + 8u * (ku[i] & kTagMask), t[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) lds_store_f64(K * lane + i < m ? rowb + 8u * (ku[i] & kTagMask) : spare, t[i]);
        }
        wave_fence();
        // ---- restore the climate-trend shift (bcsd.py:263-267), in place ---------------------------------------
        SDPH("restore");
        if (kEarlyRows) rows_load<NR, FULL>(p->ord_p + begp, m, to);  // (named ahead of the last barrier: the stores go out behind it at once)
        double* ob = row + L::own(has ? lane : 0);
        double q[K];
#pragma unroll
        for (int i = 0; i < K; ++i) q[i] = ob[i];
        const double ycr = p->return_anoms ? yc : 0.0;  // bcsd.py:266-267 (res - 0.0 == res, also for -0.0)
        if constexpr (FULL) {
            if (has) {
#pragma unroll
                for (int i = 0; i < K; ++i) ob[i] = (shift[i] + q[i]) - ycr;  // bcsd.py:253,263
            }
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const double res = (shift[i] + q[i]) - ycr;
                lds_store_f64(K * lane + i < m ? rowb + 8u * (unsigned)(L::own(lane) + i) : spare, res);
            }
        }
    }
    SDPH("store");
    SDT(13);  // scatter, shift restored
    redo_flag[wave] = redo_y ? 1 : 0;
    __syncthreads();
    SDT(14);  // last barrier
    any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {  // nothing has been written yet
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
        }
        return;
    }
    const bool vec_o = (p->ld_out % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->out) & 15) == 0);
        if (!kEarlyRows) rows_load<NR, FULL>(p->ord_p + begp, m, to);
    if (!(abl & 16)) store_tile_ti<K, FULL>(p->out, p->ld_out, to, m, c0, p->C, vec_o, tile, RS);
#ifdef SD_DEV
    SDT(15);  // stores issued
    if (traced && threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kTraceSlots; ++i) sd_fx_trace[(blockIdx.x / kTraceStride) * kTraceSlots + i] = tstamp[i];
    }
#endif
#undef SD_LANE
}

// ---- round 6: the tiles land by LDS-DMA, time-major ------------------------------------------------------------------------
// bcsd_fd_kernel<K>: the whole-lane BcsdTemperature fit + predict pass of bcsd_fx_kernel<K, true, true> (same reference
// semantics, bcsd.py:197-269, quantile.py:81-147, 438-545; same keys, sort and work list) with a different life of the tile:
//
//   landing   global_load_lds_dwordx4: one wave instruction fetches 16 rows x 64 B (the 8 cells of the tile) and the hardware
//             writes them lane-linear into LDS -- a 1 KB chunk of a TIME-MAJOR tile [t][8 cells].  No tile registers, no
//             ds_write pass, and a request in flight owns no VGPR: half of the y tile is requested BEFORE the sort of u and
//             lands while the wave sorts (the old kernel could not afford the 40 registers).  The base of every
//             chunk is skewed by 16 B (chunk stride 1 040 B): a wave reads a COLUMN of the tile (its cell; lane l owns rows
//             K l .. K l + K - 1), and without the skew the lane stride K x 64 B would put every lane on the same bank.
//   u side    the shifted series is not written back as float64: the key generation (one v_add_f64 with 2^21: q and 31
//             further bits of the fixed-point quotient come out of the mantissa) leaves a 32-bit SECOND-LEVEL key u2 per
//             sample, stored cell-major in the first 39 KB of the tile once every wave has read its windows; the fix-up of
//             equal-q neighbours compares u2 (equal u2 = equal or indistinguishable values: work list).  The rest of the
//             tile is free during the sort of u: that is where the early half of the y tile lands.
//   y side    the late half of y is requested behind the vote barrier (the u2 area is dead then); column reads, keys, sort,
//             fix-up on the float64 observations by tag, gather, scatter to the tags of u, shift restored in place.
//   output    the tile is already time-major: ds_read_b128 of a chunk, lane-linear, and 64-byte row fragments to memory.
//
// Row indices of the requests: loaded once per order table into registers (element 16 k + i of a wave's chunk list in register
// (16 k + i) / 64, lane (16 k + i) % 64) and fetched per request by ds_bpermute -- an ordinary global load between two requests
// would make the compiler wait for vmcnt(0), i.e. for the DMA queue.  The requests themselves are inline assembly for the same
// reason: the compiler treats a pending __builtin_amdgcn_global_load_lds as an LDS write and puts s_waitcnt vmcnt(0) in front of
// every DS instruction -- the ds_swizzle / ds_bpermute partner fetches of the sort included.  Waits are counted by hand.
namespace tmj {
// Layout of a tile (K = 20): a chunk = what one request lands = 16 row fragments of 64 B, and the REQUEST chooses which 16 rows
// (the source address is per lane).  Lane l of the wave that owns a cell works on rows 20 l .. 20 l + 19 of the segment:
//   chunk l             rows 20 l .. 20 l + 15          (slot i = row 20 l + i)
//   chunk nl + t(l)     the tails, rows 20 l + 16 + e    (slot s2(l) + 4 e), t(l) = (l & 7) + 8 (l >> 5), s2(l) = (l >> 3) & 3
// (nl = lanes with data = m / 20), and chunk bases are 1 032 B apart -- 8 bytes of skew, which the DMA accepts (measured:
// csrc/microbench/tile_dma).  A column read of a wave (its cell, the same row number i in every lane) then touches
// (base + 1 032 l + 64 i) / 8 mod 32 = (l + 8 i) mod 32: 32 consecutive lanes on 32 different bank pairs, conflict-free for
// ds_read_b64 and ds_read2_b64 alike; the tails likewise ((nl + t) + 8 s2 = nl + l mod 32).  (Rows in time order with the
// chunks skewed by 16 B -- the first version -- put lanes 16 chunks apart on the same banks: 3-way conflicts, and the column
// phases ran at half the speed of the old cell-major rows.)  A sample's tag = its position 16 chunk + slot (11 bits).
constexpr int kChunkStride = 1024 + 8;
constexpr int kBlock = 20;  // rows per lane (the kernel's K)
__host__ __device__ constexpr int chunks_of(int n) { return n / kBlock + 16; }  // n = whole lanes of 20, more than 32 of them
__device__ __forceinline__ int tail_chunk(int l) { return (l & 7) + 8 * (l >> 5); }
__device__ __forceinline__ int tail_slot(int l) { return (l >> 3) & 3; }
// row of the segment that slot S of chunk Q holds (n - 1 where the slot holds none)
__device__ __forceinline__ int row_of_slot(int Q, int S, int nl, int n) {
    int r;
    if (Q < nl) {
        r = kBlock * Q + S;
    } else {
        const int t = Q - nl;
        const int l = (t & 7) + 8 * (S & 3) + 32 * (t >> 3);
        r = l < nl ? kBlock * l + 16 + (S >> 2) : n - 1;
    }
    return r < n ? r : n - 1;
}
__device__ __forceinline__ unsigned tag_off(unsigned tag) { return (tag << 6) + ((tag >> 4) << 3); }  // position -> byte offset (cell 0)

template <int NREG>
struct RowIdx {
    int v[NREG];
};
// rows of the chunks q0 + wave + 8 k, k = 0 .. 4 NREG - 1
template <int NREG>
__device__ __forceinline__ RowIdx<NREG> rows_of_wave(const int32_t* __restrict__ ord, int n, int q0, int wave, int lane) {
    RowIdx<NREG> t;
    const int nl = n / kBlock;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const int e = 64 * j + lane;
        t.v[j] = ord[row_of_slot(q0 + wave + kW * (e >> 4), e & 15, nl, n)];
    }
    return t;
}
// The indices are "used" here, so the compiler's wait for their loads sits here -- ahead of the requests -- and not between
// them, where it would count only its own loads and drain the DMA queue with them.
template <int NREG>
__device__ __forceinline__ void rows_ready(RowIdx<NREG>& t) {
#pragma unroll
    for (int j = 0; j < NREG; ++j) asm volatile("" : "+v"(t.v[j]));
}
template <int NREG>
__device__ __forceinline__ int row_of_request(const RowIdx<NREG>& rows, int k, int lane) {
    return __builtin_amdgcn_ds_bpermute(4 * (16 * (k & 3) + (lane >> 2)), rows.v[k >> 2]);
}
// requests for the chunks q = q0 + wave + 8 k < q1 of a tile (NK = requests per wave at most)
template <int NK, int NREG>
__device__ __forceinline__ void dma_chunks(const double* __restrict__ src, int64_t ld, const RowIdx<NREG>& rows, int q0, int q1,
                                           int64_t c0, unsigned tile_b, int wave, int lane) {
    static_assert(NK <= 4 * NREG, "row registers");
    const char* colp = reinterpret_cast<const char*>(src + c0) + 16 * (lane & 3);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int q = q0 + wave + kW * k;
        if (q < q1) {  // (wave-uniform)
            const int ti = row_of_request(rows, k, lane);
            const char* g = colp + (uint64_t)(uint32_t)ti * (uint64_t)(uint32_t)((uint32_t)ld * 8u);
            const int dst = __builtin_amdgcn_readfirstlane((int)(tile_b + (unsigned)q * (unsigned)kChunkStride));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(g), "s"(dst)
                         : "memory");
        }
    }
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Column access of lane block bl of the wave's cell (rows 20 bl + idx, idx = -4 .. 23): four bases, immediate offsets
//   idx -4 .. -1  the tail of lane bl - 1      idx 0 .. 15  the lane's chunk      idx 16 .. 19  its tail
//   idx 20 .. 23  the head of lane bl + 1's chunk (one chunk stride further)
// (pointer arithmetic: an in-bounds element offset becomes the DS instruction's immediate; an unsigned sum does not)
struct Col {
    unsigned mainb, tailb, prevb;
    __device__ __forceinline__ unsigned at(int idx) const {  // idx: compile-time after unrolling
        return idx < 0 ? prevb + (unsigned)(256 * (idx + 4)) : idx < 16 ? mainb + (unsigned)(64 * idx)
             : idx < 20 ? tailb + (unsigned)(256 * (idx - 16)) : mainb + (unsigned)(kChunkStride + 64 * (idx - 20));
    }
    __device__ __forceinline__ double get(int idx) const {
        if (idx < 0) return reinterpret_cast<lds_cdouble_t*>((uintptr_t)prevb)[32 * (idx + 4)];
        if (idx < 16) return reinterpret_cast<lds_cdouble_t*>((uintptr_t)mainb)[8 * idx];
        if (idx < 20) return reinterpret_cast<lds_cdouble_t*>((uintptr_t)tailb)[32 * (idx - 16)];
        return reinterpret_cast<lds_cdouble_t*>((uintptr_t)mainb)[kChunkStride / 8 + 8 * (idx - 20)];
    }
    __device__ __forceinline__ void put(int idx, double x) const {  // own rows only (0 <= idx < 20)
        if (idx < 16) reinterpret_cast<lds_double_t*>((uintptr_t)mainb)[8 * idx] = x;
        else reinterpret_cast<lds_double_t*>((uintptr_t)tailb)[32 * (idx - 16)] = x;
    }
};
__device__ __forceinline__ Col col_of(unsigned tile_b, int col, int bl, int nl) {
    const unsigned cb = tile_b + 8u * (unsigned)col;
    const int lp = (bl - 1) & 63;  // (lane 0 has no predecessor: it reads -- and drops -- the tail slot of lane 63)
    Col c;
    c.mainb = cb + (unsigned)bl * (unsigned)kChunkStride;
    c.tailb = cb + (unsigned)(nl + tail_chunk(bl)) * (unsigned)kChunkStride + 64u * (unsigned)tail_slot(bl);
    c.prevb = cb + (unsigned)(nl + tail_chunk(lp)) * (unsigned)kChunkStride + 64u * (unsigned)tail_slot(lp);
    return c;
}
// tags (positions) of the lane's samples: i < 16: 16 lane + i; tails: 16 (nl + t) + s2 + 4 e
struct Tags {
    unsigned main0, tail0;
    __device__ __forceinline__ unsigned of(int i) const { return i < 16 ? main0 + (unsigned)i : tail0 + (unsigned)(4 * (i - 16)); }
};
__device__ __forceinline__ Tags tags_of(int lane, int nl) {
    return Tags{16u * (unsigned)lane, 16u * (unsigned)(nl + tail_chunk(lane)) + (unsigned)tail_slot(lane)};
}

// keys + second-level keys of the shifted series: t = (v - lo) * sc + 1 in [1, kQD - 1]; t + 2^21 has the exponent of 2^21, so its
// mantissa is t in units of 2^-31: 21 bits of q above 31 further bits.  Monotone in v (fma, add: correctly rounded).
// key = (q << 11) | sample index; u2 = the low mantissa word (bit 31 = the lowest bit of q: equal wherever q is).
template <int K, class GetU>
__device__ __forceinline__ void keys_u2(const GetU& u_of, double lo, double hi, int m, int lane, const Tags& tg, unsigned (&key)[K], unsigned (&u2)[K]) {
    const double sc = (double)(kQD - 2u) / (hi - lo);  // +inf when every sample is equal: t = NaN everywhere, all keys and u2 tie
    const double off = -lo * sc + 1.0;
    const unsigned tag0 = (unsigned)(K * lane);
    const unsigned pad0 = ((kQD + 1u + tag0) << kTagBits) | tag0;
    const bool lane_in = K * lane < m;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const double t = __builtin_fma(u_of(i), sc, off) + 2097152.0;
        const unsigned w0 = (unsigned)__double2loint(t), w1 = (unsigned)__double2hiint(t);
        unsigned q = __builtin_amdgcn_alignbit(w1, w0, 31) & 0x1fffffu;
        q = q < kQD ? q : kQD;
        const unsigned dk = (q << kTagBits) | tg.of(i);
        const unsigned pk = pad0 + (unsigned)i * ((1u << kTagBits) + 1u);
        key[i] = lane_in ? dk : pk;
        u2[i] = w0;
    }
}
// keys of the observations (the cvt path of keys_from_range, whole lanes) with position tags
template <int K>
__device__ __forceinline__ void keys_pos(const double (&v)[K], int m, int lane, const Tags& tg, unsigned (&key)[K]) {
    double lo = __builtin_inf(), hi = -__builtin_inf();
#pragma unroll
    for (int i = 0; i < K; ++i) {
        lo = vmin(lo, v[i]);
        hi = vmax(hi, v[i]);
    }
    lo = wave_min_f64(lo);
    hi = wave_max_f64(hi);
    const double sc = (double)kQD / (hi - lo);
    const double off = -lo * sc;
    const unsigned tag0 = (unsigned)(K * lane);
    const unsigned pad0 = ((kQD + 1u + tag0) << kTagBits) | tag0;
    const bool lane_in = K * lane < m;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        unsigned q = (unsigned)__builtin_fma(v[i], sc, off);
        q = q < kQD ? q : kQD;
        const unsigned dk = (q << kTagBits) | tg.of(i);
        const unsigned pk = pad0 + (unsigned)i * ((1u << kTagBits) + 1u);
        key[i] = lane_in ? dk : pk;
    }
}

// exact order inside runs of equal q (see fix_equal_q): the values behind two tags are fetched and compared by `cmp`
// (tag_a, tag_b, &greater, &equal)
template <int K, class Cmp>
__device__ __forceinline__ int fix_equal_q_by(unsigned (&k)[K], int lane, const Cmp& cmp) {
    constexpr unsigned kQ = 1u << kTagBits;
    bool tie = false;
#pragma unroll 1
    for (int pass = 0; pass < 6; ++pass) {
        bool swapped = false;
        const unsigned knext = from_next_lane(k[0], 0xffffffffu);
        const bool eqx = (k[K - 1] ^ knext) < kQ;
        const unsigned long long bx = __ballot(eqx);
        if (bx != 0ull) {  // wave-uniform, rare
            bool gt, eq;
            cmp(k[K - 1] & kTagMask, knext & kTagMask, &gt, &eq);
            const bool sw = eqx && gt;
            tie |= eqx && eq;
            const unsigned d = sw ? (k[K - 1] ^ knext) : 0u;
            k[K - 1] ^= d;
            k[0] ^= from_prev_lane(d, 0u);
            swapped |= sw;
        }
        unsigned long long prev = 0ull, b0 = 0ull, multi = 0ull;
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) {
            const bool e = (k[i] ^ k[i + 1]) < kQ;
            const unsigned long long b = __ballot(e);
            multi |= b & prev;
            if (i == 0) b0 = b;
            prev = b;
            if (b != 0ull) {
                bool gt, eq;
                cmp(k[i] & kTagMask, k[i + 1] & kTagMask, &gt, &eq);
                const bool sw = e && gt;
                tie |= e && eq;
                const unsigned d = sw ? (k[i] ^ k[i + 1]) : 0u;
                k[i] ^= d;
                k[i + 1] ^= d;
                swapped |= sw;
            }
        }
        multi |= (bx & prev) | ((bx << 1) & b0);
        if (multi == 0ull || !__any(swapped)) return __any(tie) ? kTie : 0;
    }
    return kUnsorted;
}
}  // namespace tmj

// LDS of a workgroup: [head: kHeadDoubles][tile: chunks_of(nmax) x 1 032 B]; the u2 area (8 cells x RSU 32-bit words, indexed by
// tag) overlays the first chunks of the tile
__host__ __device__ constexpr int fd_u2_stride(int nmax) { return 16 * tmj::chunks_of(nmax); }
__host__ __device__ constexpr int fd_late_chunks(int rsu) { return (kW * 4 * rsu + tmj::kChunkStride - 1) / tmj::kChunkStride; }  // rsu = fd_u2_stride(nmax)
inline size_t fd_lds_bytes(int nmax) { return (size_t)kHeadDoubles * sizeof(double) + (size_t)tmj::chunks_of(nmax) * tmj::kChunkStride; }

// EARLY: half of the y tile is requested ahead of the sort of u (the second-level keys are compacted into the other half behind a
// workgroup barrier); !EARLY: the second-level keys stay in the wave's own column (low words of its slots: no barrier), the whole y
// tile is requested behind the vote.
template <int K, bool EARLY>
__global__ void __launch_bounds__(kThreads, 4) bcsd_fd_kernel(const Params) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NR = K / 2;
    constexpr int CH = K >= 20 ? K / 4 : K >= 14 ? K / 2 : K;
    constexpr int NKX = 10;  // requests per wave and tile at most: 80 chunks = 1 280 rows
    static_assert(K == tmj::kBlock && K % CH == 0, "the chunk layout is that of 20 rows per lane");
#ifdef SD_DEV
    const int abl = p->dev_flags;
#else
    constexpr int abl = 0;
#endif
    double* const scratch = reinterpret_cast<double*>(smem_raw);
    int* const bad_cell = reinterpret_cast<int*>(scratch + 64 + 16);
    int* const redo_flag = bad_cell + kW;
    const unsigned tile_b = lds_addr(smem_raw) + (unsigned)(kHeadDoubles * sizeof(double));
    if (threadIdx.x >= 32 && threadIdx.x < 32 + kW) bad_cell[threadIdx.x - 32] = 0;
#ifdef SD_DEV
    const bool traced = (abl & 0x800) != 0 && blockIdx.x % kTraceStride == 7 && blockIdx.x / kTraceStride < (unsigned)kTraceWgs;
    long long tstamp[kTraceSlots] = {};
    SDT(0);
#endif
    int64_t tile_id;
    int g;
    xcd_tile_of_block(blockIdx.x, p->ntiles, &tile_id, &g);
    if (p->gmask != 0ull) g = nth_set_bit(p->gmask, g);
    if (tile_id >= p->ntiles || g < 0 || g >= p->G) return;

    const int64_t c0 = tile_id * kW;
    const int wave = __builtin_amdgcn_readfirstlane(tid_now() / kWave);
#define SD_LANE() const int lane = tid_now() % kWave
    const int64_t c = c0 + wave;
    const bool cell_ok = c < p->C;
    const int begf = p->off_f[g];
    const int m = p->off_f[g + 1] - begf;  // == the predict segment's length (whole-lane groups of equal length)
    const int begp = p->off_p[g];
    if (m == 0) return;
    const int nl = m / K;                     // lanes with data
    const int nch = tmj::chunks_of(m);
    const int RSU = p->RS;                    // 32-bit words per cell of the u2 area
    const int nlate = fd_late_chunks(RSU) < nch ? fd_late_chunks(RSU) : nch;  // chunks under the u2 area: the late half of y
    const bool cell_live = cell_ok && p->status_fit[cell_ok ? c : 0] == 0;
    // a tile whose last cells lie past the grid (C is even: pairs of cells are whole) fetches its last whole pair instead
    const int64_t cfetch = c0 + kW <= p->C ? c0 : p->C - kW;  // (C >= 8: the launcher's condition)
    const int cshift = (int)(c0 - cfetch);                     // cells the fetched tile is shifted by: wave w's cell sits in column w + cshift
    const int col = wave + cshift < kW ? wave + cshift : wave + cshift - kW;  // (waves past the grid get the columns in front: cells of the previous tile, nothing of them is kept)
    __syncthreads();  // bad_cell zeroed before the checks below may set it

    // ---- x climatology (bcsd.py:222) from registers; the x_fut tile by DMA ------------------------------------------------
    SDPH("x_tiles");
    double xc = 0.0;
    tmj::RowIdx<EARLY ? 2 : 3> ry_late;
    tmj::RowIdx<2> ry_early;
    {
        SD_LANE();
        tmj::RowIdx<3> rp = tmj::rows_of_wave<3>(p->ord_p + begp, m, 0, wave, lane);  // (in one batch with the rows of the x_hist tile)
        TileRegs<NR> xh;
        int tf[NR];
        rows_load<NR, true>(p->ord_f + begf, m, tf);
        // the DMA requests first, the register loads of x_hist behind them (the compiler's waits for the latter then cover both;
        // in the other order it drains its own loads before the first request goes out)
        tmj::rows_ready(rp);
#pragma unroll
        for (int k = 0; k < NR; ++k) asm volatile("" : "+v"(tf[k]));  // (likewise: the wait for these sits ahead of the requests)
        if (!(abl & 64)) tmj::dma_chunks<NKX, 3>(p->Xp, p->ld_p, rp, 0, nch, cfetch, tile_b, wave, lane);
        if (!(abl & 4)) {
            // (tile_issue_ti without its scalar branch for cells at an odd boundary: the launcher admits whole pairs only, a pair
            // past the grid re-reads the tile's first one -- its sums belong to no cell.  The sibling branch's un-waited loads would
            // make the compiler drain vmcnt -- the DMA queue -- on the way into this one.)
            const int cp = lane & 3;
            const double* src = p->X + (c0 + 2 * cp + 1 < p->C ? c0 + 2 * cp : c0);
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const double2 v = *reinterpret_cast<const double2*>(row_of(src, tf[k], p->ld));
                xh.v0[k] = v.x;
                xh.v1[k] = v.y;
            }
        }
        if (!(abl & 4)) tile_reduce_partials<NR, true>(xh, m, c0, p->C, scratch, p->status_fit, wave, lane, bad_cell);
        SDT(1);  // x_hist summed, x_fut requested
        tmj::dma_wait_all();
    }
    __syncthreads();
    SDT(2);  // x_fut tile landed
    {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kW; ++w) tot += scratch[w * kW + wave];
        xc = tot / (double)m;
    }

    // ---- shift (kept), second-level keys -> u2 area, keys -> sort ---------------------------------------------------------
    SDPH("rolling");
    double shift[K];
    unsigned ku[K];
    bool redo = false;
    {
        SD_LANE();
        const bool has = K * lane < m;
        const int bl = has ? lane : 0;
        const bool first_lane = bl == 0, last_lane = K * (bl + 1) == m;
        const tmj::Col cz = tmj::col_of(tile_b, col, bl, nl);
        // (the shifted samples themselves are not kept -- 40 registers beside the 40 of the shifts --: their extremes are taken
        // here, the key generation below re-reads the samples from the column and subtracts the kept shifts again)
        double ulo = __builtin_inf(), uhi = -__builtin_inf();
        bool bad = false;
#pragma unroll
        for (int cbeg = 0; cbeg < K; cbeg += CH) {
            double w[CH + 8];
#pragma unroll
            for (int t = 0; t < CH + 8; ++t) {
                const int idx = cbeg - 4 + t;  // row K * bl + idx of the cell's column
                // (every lane reads, also where the row lies before or behind the segment -- some other slot of the tile -- and
                // drops the value: a select on the address made the compiler wrap every such read in a divergent branch)
                const double x = cz.get(idx);
                w[t] = idx < 0 ? (first_lane ? 0.0 : x) : idx >= K ? (last_lane ? 0.0 : x) : x;
            }
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                double s = 0.0;  // the nine samples in time order, like the oracle and every other path
#pragma unroll
                for (int d = 0; d < 9; ++d) s += w[ii + d];
                const int i = cbeg + ii;
                constexpr double kRc[10] = {0.0, 1.0, 0.5, 1.0 / 3.0, 0.25, 0.2, 1.0 / 6.0, 1.0 / 7.0, 0.125, 1.0 / 9.0};
                double cd = 9.0, rc = kRc[9];
                if (i < 4) {
                    cd = first_lane ? (double)(5 + i) : cd;
                    rc = first_lane ? kRc[5 + i] : rc;
                }
                if (i >= K - 4) {
                    cd = last_lane ? (double)(K + 4 - i) : cd;
                    rc = last_lane ? kRc[K + 4 - i] : rc;
                }
                const double q = s * rc;
                const double mean = __builtin_fma(__builtin_fma(-cd, q, s), rc, q);  // correctly rounded s / count
                const double sh = mean - xc;                                           // bcsd.py:253
                shift[i] = sh;
                bad |= nonfinite64(w[ii + 4]);
                const double ui = (w[ii + 4] - sh) + 0.0;  // bcsd.py:256; -0.0 -> +0.0
                ulo = vmin(ulo, ui);
                uhi = vmax(uhi, ui);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (__any(bad && has)) {  // (the x_fut tile is validated by the wave that owns the cell: bcsd.py:244)
            if (lane == 0) {
                if (cell_ok) atomicOr(&p->status_p[c], SDI_NONFINITE);
                bad_cell[wave] = 1;
            }
        }
        SDT(3);  // rolling mean
        // the rows of the y requests (the loads return under the key generation)
        ry_late = tmj::rows_of_wave<EARLY ? 2 : 3>(p->ord_f + begf, m, 0, wave, lane);
        if (EARLY) ry_early = tmj::rows_of_wave<2>(p->ord_f + begf, m, nlate, wave, lane);
        SDPH("u_keys");
        unsigned u2[K];
        {
            ulo = wave_min_f64(ulo);
            uhi = wave_max_f64(uhi);
            const auto u_of = [&](int i) { return (cz.get(i) - shift[i]) + 0.0; };
            tmj::keys_u2<K>(u_of, ulo, uhi, m, lane, tmj::tags_of(lane, nl), ku, u2);
        }
        tmj::rows_ready(ry_late);
        unsigned ub;
        if constexpr (EARLY) {
            // every wave has read its column for the last time: the tile is dead, the y requests and the u2 area may overwrite it
            tmj::rows_ready(ry_early);
            __syncthreads();
            if (!(abl & 32)) tmj::dma_chunks<5, 2>(p->y, p->ld, ry_early, nlate, nch, cfetch, tile_b, wave, lane);
            ub = tile_b + 4u * (unsigned)(col * RSU);
        } else {
            wave_fence();  // every lane of the wave has read its rows: their low words take the second-level keys
            ub = 0u;
        }
        if (!EARLY) {
            if (has) {
                const tmj::Col cw = tmj::col_of(tile_b, col, lane, nl);
#pragma unroll
                for (int i = 0; i < K; ++i) *reinterpret_cast<__attribute__((address_space(3))) unsigned*>((uintptr_t)cw.at(i)) = u2[i];
            }
        } else if (has) {  // by tag: the lane's chunk is 16 consecutive words, its tail four words 16 bytes apart
            const tmj::Tags tg = tmj::tags_of(lane, nl);
            typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
            __attribute__((address_space(3))) u32x4* um = reinterpret_cast<__attribute__((address_space(3))) u32x4*>((uintptr_t)(ub + 4u * tg.main0));
#pragma unroll
            for (int i = 0; i < 16; i += 4) um[i / 4] = u32x4{u2[i], u2[i + 1], u2[i + 2], u2[i + 3]};
            __attribute__((address_space(3))) unsigned* ut = reinterpret_cast<__attribute__((address_space(3))) unsigned*>((uintptr_t)(ub + 4u * tg.tail0));
#pragma unroll
            for (int i = 16; i < K; ++i) ut[4 * (i - 16)] = u2[i];
        }
        wave_fence();
        SDT(4);  // u2 stored, keys
        SDPH("u_sort");
        if (!(abl & 1)) sdws::wave_sort<K>(ku, lane, (m + K - 1) / K);
        SDT(5);  // sort of u
        SDPH("u_fix");
        const unsigned colb_u = tile_b + 8u * (unsigned)col;
        const auto cmp_u2 = [ub, colb_u](unsigned ta, unsigned tb, bool* gt, bool* eq) {
            const unsigned a = lds_u32(EARLY ? ub + 4u * ta : colb_u + tmj::tag_off(ta)), b = lds_u32(EARLY ? ub + 4u * tb : colb_u + tmj::tag_off(tb));
            *gt = a > b;
            *eq = a == b;
        };
        const bool tie = (abl & 8) ? false : tmj::fix_equal_q_by<K>(ku, lane, cmp_u2) != 0;
        redo = tie && cell_live && bad_cell[wave] == 0 && (abl & 0x7ff) == 0;  // wave-uniform
        SDT(6);  // fix-up of u
    }
    SDPH("vote");
    redo_flag[wave] = redo ? 1 : 0;
    __syncthreads();
    int any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {
        tmj::dma_wait_all();  // (no request of this workgroup may land in the LDS of its successor)
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
        }
        return;
    }
    SDT(7);  // vote

    // ---- y: the late half of the tile, climatology, sorted observations ---------------------------------------------------
    SDPH("y_tile");
    double yc = 0.0;
    double t[K];
    bool redo_y = false;
    {
        SD_LANE();
        if (!(abl & 32)) tmj::dma_chunks<EARLY ? 5 : NKX, EARLY ? 2 : 3>(p->y, p->ld, ry_late, 0, EARLY ? nlate : nch, cfetch, tile_b, wave, lane);
        SDT(8);  // late half requested
        tmj::dma_wait_all();
        __syncthreads();
        SDT(9);  // y tile landed
        SDPH("y_keys");
        const bool has = K * lane < m;
        const int bl = has ? lane : 0;
        const tmj::Col cz = tmj::col_of(tile_b, col, bl, nl);
        const unsigned colb = tile_b + 8u * (unsigned)col;
        unsigned ky[K];
        {
            double v[K];
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = cz.get(i);
            double s = 0.0;
            bool bad = false;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                s += v[i];
                bad |= nonfinite64(v[i]);
            }
            s = has ? s : 0.0;
            yc = wave_sum_f64(s) / (double)m;  // bcsd.py:223
            if (__any(bad && has) && lane == 0 && cell_ok) atomicOr(&p->status_fit[c], SDI_NONFINITE);
            tmj::keys_pos<K>(v, m, lane, tmj::tags_of(lane, nl), ky);
        }
        SDT(10);  // y_climo, keys
        SDPH("y_sort");
        if (!(abl & 2)) sdws::wave_sort<K>(ky, lane, (m + K - 1) / K);
        SDT(11);  // sort of y
        SDPH("y_fix");
        const auto cmp_y = [colb](unsigned ta, unsigned tb, bool* gt, bool* eq) {
            const double a = lds_f64(colb + tmj::tag_off(ta)), b = lds_f64(colb + tmj::tag_off(tb));
            *gt = a > b;
            *eq = a == b;
        };
        if (!(abl & 8)) redo_y = (tmj::fix_equal_q_by<K>(ky, lane, cmp_y) & kUnsorted) != 0 && cell_live;  // tied observations are interchangeable
        SDPH("y_gather");
        {
            const unsigned tm = has ? kTagMask : 0u;  // (pad keys carry tags of no sample: those lanes read position 0)
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = lds_f64(colb + tmj::tag_off(ky[i] & tm));
        }
        SDT(12);  // fix-up of y, sorted observations gathered
        // ---- identity map (equal group lengths): rank r of u takes the r-th smallest observation; scatter to the time slots ----
        SDPH("map_scatter");
        wave_fence();  // every lane has read what it needs of the column
        if (has) {
#pragma unroll
            for (int i = 0; i < K; ++i) lds_store_f64(colb + tmj::tag_off(ku[i] & kTagMask), t[i]);
        }
        wave_fence();
        SDPH("restore");
        double q[K];
#pragma unroll
        for (int i = 0; i < K; ++i) q[i] = cz.get(i);
        const double ycr = p->return_anoms ? yc : 0.0;  // bcsd.py:266-267
        if (has) {
#pragma unroll
            for (int i = 0; i < K; ++i) cz.put(i, (shift[i] + q[i]) - ycr);  // bcsd.py:253, 263
        }
    }
    SDPH("store");
    SDT(13);  // scatter, shift restored
    redo_flag[wave] = redo_y ? 1 : 0;
    tmj::RowIdx<3> ro = tmj::rows_of_wave<3>(p->ord_p + begp, m, 0, wave, tid_now() % kWave);  // (named ahead of the last barrier)
    __syncthreads();
    tmj::rows_ready(ro);
    SDT(14);  // last barrier
    any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {  // nothing has been written yet
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
        }
        return;
    }
    if (!(abl & 16)) {
        SD_LANE();
        // the tile is time-major already: a chunk is read lane-linear and leaves as 16 row fragments of 64 bytes
        const int pair = 2 * (lane & 3);                     // cells pair, pair + 1 of the fetched tile
        const bool keep = pair >= cshift;                    // (a tile shifted back over its predecessor stores its own cells only)
        char* colp = reinterpret_cast<char*>(p->out + cfetch) + 16 * (lane & 3);
#pragma unroll
        for (int k = 0; k < NKX; ++k) {
            const int qc = wave + kW * k;
            const int S = lane >> 2;
            // (a tail slot of a lane without data holds no row)
            const bool row_in = qc < nl ? K * qc + S < m : (((qc - nl) & 7) + 8 * (S & 3) + 32 * ((qc - nl) >> 3)) < nl;
            const int ti = tmj::row_of_request(ro, k, lane);
            if (qc < nch && row_in && keep) {
                typedef double __attribute__((ext_vector_type(2))) f64x2;
                const f64x2 v = *reinterpret_cast<__attribute__((address_space(3))) const f64x2*>((uintptr_t)(tile_b + (unsigned)qc * (unsigned)tmj::kChunkStride + 16u * (unsigned)lane));
                *reinterpret_cast<f64x2*>(colp + (uint64_t)(uint32_t)ti * (uint64_t)(uint32_t)((uint32_t)p->ld_out * 8u)) = v;
            }
        }
    }
#ifdef SD_DEV
    SDT(15);  // stores issued
    if (traced && threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kTraceSlots; ++i) sd_fx_trace[(blockIdx.x / kTraceStride) * kTraceSlots + i] = tstamp[i];
    }
#endif
#undef SD_LANE
}

// ---- BcsdPrecipitation (bcsd.py:115-185) --------------------------------------------------------------------------
// The same pass without a climate-trend shift: x_hist is only validated (bcsd.py:130-147), the raw x_fut series is ranked
// (bcsd.py:167), the result is the mapped value over y_climo (ratio anomalies, bcsd.py:170-185).  Zero-inflated series: the
// exact zeros of a segment are one tie -- np.interp gives every one of them the largest rank among them (quantile.py:488) --
// so they form key class 0 (sorted in front, in any order, never compared), n0 of them are counted, and sorted position r
// maps through rank max(r, n0 - 1).  Ties among the wet days, or negative values, send the (tile, group) to RANK / APPLY.
template <int K, bool IDENT, bool FULL>
__device__ __forceinline__ void fxp_segment(ParamsPtr p, int64_t tile_id, int g, char* smem_raw) {
    static_assert(!FULL || IDENT, "FULL launches serve groups of equal fit / predict length");
    constexpr int NR = K / 2;
    constexpr int CH = K >= 20 ? K / 4 : K >= 14 ? K / 2 : K;
    using L = Lay<K>;
    double* const scratch = reinterpret_cast<double*>(smem_raw);
    int* const bad_cell = reinterpret_cast<int*>(scratch + 64 + 16);
    int* const redo_flag = bad_cell + kW;
    double* const tile = scratch + kHeadDoubles;
    const int RS = p->RS;
    if (threadIdx.x >= 32 && threadIdx.x < 32 + kW) bad_cell[threadIdx.x - 32] = 0;

    const int64_t c0 = tile_id * kW;
    const int wave = __builtin_amdgcn_readfirstlane(tid_now() / kWave);
#define SD_LANE() const int lane = tid_now() % kWave
    const int64_t c = c0 + wave;
    const bool cell_ok = c < p->C;
    double* const row = tile + wave * RS;
    const unsigned rowb = lds_addr(row);
    const int64_t seg = c * p->G + g;
    const int begf = p->off_f[g];
    const int n = p->off_f[g + 1] - begf;
    const int begp = p->off_p[g];
    const int m = p->off_p[g + 1] - begp;
    if (m == 0) return;
    const bool vec_f = (p->ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->y) & 15) == 0) &&
                       (p->X == nullptr || (reinterpret_cast<uintptr_t>(p->X) & 15) == 0);
    const bool vec_p = (p->ld_p % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->Xp) & 15) == 0);
    const bool cell_live = cell_ok && p->status_fit[cell_ok ? c : 0] == 0;
    const unsigned spare = rowb + 8u * (unsigned)(RS - 1);
    __syncthreads();

    // ---- x side: validation of x_hist, the x_fut tile ------------------------------------------------------------------
    {
        TileRegs<NR> xf;
        int tp[NR];
        rows_load<NR, FULL>(p->ord_p + begp, m, tp);
        if (!p->from_state && p->X != nullptr && n > 0) {
            TileRegs<NR> xh;
            int tf[NR];
            rows_load<NR, FULL>(p->ord_f + begf, n, tf);
            tile_issue_ti<NR>(p->X, p->ld, tf, c0, p->C, vec_f, xh);
            tile_issue_ti<NR>(p->Xp, p->ld_p, tp, c0, p->C, vec_p, xf);
            tile_check_finite<NR>(xh, n, c0, p->C, p->status_fit, bad_cell);
        } else {
            tile_issue_ti<NR>(p->Xp, p->ld_p, tp, c0, p->C, vec_p, xf);
        }
        tile_commit_sw<NR, K, FULL>(xf, m, c0, p->C, tile, RS, p->status_p, bad_cell);
    }
    __syncthreads();

    unsigned ku[K];  // sorted keys of the x_fut segment: tag = time slot of the sample at that sorted position
    int n0 = 0;      // exact zeros of the x_fut segment
    bool redo = false;
    TileRegs<NR> yt;  // the y_obs tile is requested ahead of the sort: no shift to keep here, the registers are free
    {
        SD_LANE();
        const double* ob = row + L::own(K * lane < m ? lane : 0);
        double v[K];
#pragma unroll
        for (int i = 0; i < K; ++i) v[i] = ob[i];
        if constexpr (FULL) {  // (a lane is all data or all pad)
            const bool has = K * lane < m;
#pragma unroll
            for (int i = 0; i < K; ++i) n0 += __popcll(__ballot(has && v[i] == 0.0));
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) n0 += __popcll(__ballot(K * lane + i < m && v[i] == 0.0));
        }
        const double lo = make_keys<K, true, FULL>(v, m, lane, ku);
        if (!p->from_state && n > 0) {
            int ty[NR];
            rows_load<NR, FULL>(p->ord_f + begf, n, ty);
            tile_issue_ti<NR>(p->y, p->ld, ty, c0, p->C, vec_f, yt);
        }
        sdws::wave_sort<K>(ku, lane, (m + K - 1) / K);
        const bool tie = fix_equal_q<K, true>(ku, rowb, lane) != 0;
        redo = (tie || lo < 0.0) && cell_live && bad_cell[wave] == 0;
    }
    redo_flag[wave] = redo ? 1 : 0;
    __syncthreads();
    int any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
        }
        return;
    }

    // ---- y: climatology + sorted observations in the row (plain order) ----------------------------------------------
    double yc = 0.0;
    bool redo_y = false;
    if (!p->from_state) {
        if (n > 0) {
            SD_LANE();
            tile_commit_sw<NR, K, FULL>(yt, n, c0, p->C, tile, RS, p->status_fit, nullptr);
            __syncthreads();
            unsigned ky[K];
            {
                const double* ob = row + L::own(K * lane < n ? lane : 0);
                double v[K];
#pragma unroll
                for (int i = 0; i < K; ++i) v[i] = ob[i];
                double s = 0.0;
                if constexpr (FULL) {
#pragma unroll
                    for (int i = 0; i < K; ++i) s += v[i];
                    s = K * lane < n ? s : 0.0;
                } else {
#pragma unroll
                    for (int i = 0; i < K; ++i) s += K * lane + i < n ? v[i] : 0.0;
                }
                yc = wave_sum_f64(s) / (double)n;  // bcsd.py:138
                if (lane == 0 && cell_ok && p->return_anoms && yc <= 0.0) atomicOr(&p->status_fit[c], SDI_BAD_CLIMO);  // bcsd.py:140-141
                const double lo = make_keys<K, true, FULL>(v, n, lane, ky);
                redo_y = lo < 0.0 && cell_live;
            }
            sdws::wave_sort<K>(ky, lane, (n + K - 1) / K);
            redo_y = (redo_y || (fix_equal_q<K, true>(ky, rowb, lane) & kUnsorted) != 0) && cell_live;  // tied observations are interchangeable
            double t[K];
            if constexpr (FULL) {  // (the pad keys of the lanes past the segment carry slots beyond the row: those lanes read slot 0)
                const unsigned tm = K * lane < n ? kTagMask : 0u;
#pragma unroll
                for (int i = 0; i < K; ++i) t[i] = lds_f64(rowb + 8u * (ky[i] & tm));
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) t[i] = lds_f64(rowb + 8u * (K * lane + i < n ? (ky[i] & kTagMask) : (unsigned)(RS - 1)));
            }
            wave_fence();  // all reads by tag done: the row becomes the sorted segment (np.sort, quantile.py:462)
            if constexpr (FULL) {
                if (K * lane < n) {
#pragma unroll
                    for (int i = 0; i < K; ++i) lds_store_f64(rowb + 8u * (unsigned)(K * lane + i), t[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) lds_store_f64(K * lane + i < n ? rowb + 8u * (unsigned)(K * lane + i) : spare, t[i]);
            }
        }
    } else {
        SD_LANE();
        if (cell_ok) {
            yc = p->y_climo[seg];
            const double* src = p->ys + c * p->Tf + begf;
            for (int i = lane; i < n; i += kWave) row[i] = src[i];
        }
    }

    // ---- map sorted position r through rank max(r, n0 - 1) and the fitted inverse CDF (quantile.py:488, 523-545) --------
    int to[NR];  // time indices of this thread's rows of the output tile
    {
        SD_LANE();
        wave_fence();
        double t[K];
        const int rz = n0 - 1;
        if (IDENT) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int r = K * lane + i;
                const int re = r > rz ? r : rz;
                t[i] = row[r < m ? re : 0];
            }
        } else {
            double slo = 0.0, ilo = 0.0, shi = 0.0, ihi = 0.0;
            if (m > n && n > 0) {
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
                const int r = K * lane + i;
                const bool in = r < m;
                const int re = in ? (r > rz ? r : rz) : 0;
                const int idx = in ? qi[re] : -3;
                const double w = in ? qv[re] : 0.0;
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
                if ((i + 1) % CH == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
        wave_fence();  // every lane has read what it needs of the row
        const bool has = K * lane < m;
        if constexpr (FULL) {
            if (has) {
#pragma unroll
                for (int i = 0; i < K; ++i) lds_store_f64(rowb + 8u * (ku[i] & kTagMask), t[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) lds_store_f64(K * lane + i < m ? rowb + 8u * (ku[i] & kTagMask) : spare, t[i]);
        }
        wave_fence();
        // ---- ratio anomalies (bcsd.py:170-185), in place: q / y_climo by the reciprocal and one correction step ------------
        double* ob = row + L::own(has ? lane : 0);
        double q[K];
#pragma unroll
        for (int i = 0; i < K; ++i) q[i] = ob[i];
        const double rc = 1.0 / yc;
        if (!FULL || has) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                double res = q[i];
                if (p->return_anoms) {
                    const double a = q[i] * rc;
                    res = __builtin_fma(__builtin_fma(-yc, a, q[i]), rc, a);
                    res = __builtin_isfinite(res) ? res : q[i] / yc;  // (zero or denormal climatology: the plain quotient)
                }
                if constexpr (FULL) ob[i] = res;
                else lds_store_f64(K * lane + i < m ? rowb + 8u * (unsigned)(L::own(lane) + i) : spare, res);
            }
        }
        if (FULL) rows_load<NR, FULL>(p->ord_p + begp, m, to);  // (named ahead of the last barrier: the stores go out behind it at once)
    }
    redo_flag[wave] = redo_y ? 1 : 0;
    __syncthreads();
    any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {  // negative observations: nothing has been written, RANK / APPLY take the (tile, group)
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(p->work_count, 1);
            if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
        }
        return;
    }
    const bool vec_o = (p->ld_out % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->out) & 15) == 0);
    if (!FULL) rows_load<NR, FULL>(p->ord_p + begp, m, to);
    store_tile_ti<K, FULL>(p->out, p->ld_out, to, m, c0, p->C, vec_o, tile, RS);
#undef SD_LANE
}

template <int K, bool IDENT, bool FULL>
__global__ void __launch_bounds__(kThreads, 4) bcsd_fxp_kernel(const Params) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    if (p->use_worklist == 2) {  // the segments the compacting kernel handed over (too many wet days): a fixed grid walks the list
        int count = *p->work_count2;
        if (count > p->work_cap) count = p->work_cap;
#pragma unroll 1
        for (int i = (int)blockIdx.x; i < count; i += (int)gridDim.x) {
            const int64_t item = p->worklist2[i];
            fxp_segment<K, IDENT, FULL>(p, item / p->G, (int)(item % p->G), smem_raw);
            __syncthreads();  // the tile is reused by the next item
        }
        return;
    }
    int64_t tile_id;
    int g;
    xcd_tile_of_block(blockIdx.x, p->ntiles, &tile_id, &g);
    if (p->gmask != 0ull) g = nth_set_bit(p->gmask, g);
    if (tile_id >= p->ntiles || g < 0 || g >= p->G) return;
    fxp_segment<K, IDENT, FULL>(p, tile_id, g, smem_raw);
}

// ---- BcsdPrecipitation, fit + predict, whole lanes, equal group lengths: only the wet days are sorted ---------------------
// A zero-inflated segment (45 - 60 % exact zeros) needs no order among its zeros: every one of them takes the largest rank
// among them (quantile.py:488).  Once a lane holds its K samples the row is dead until results are scattered, so the wet
// values are compacted into its front (ballot-free: per-lane counts, one wave scan, K conditional stores), read back KC
// consecutive ones per lane and sorted with the KC-wide network (KC = 12 for K = 20: 800 instead of 1 520 instructions per
// sort); rank = zeros + rank among the wet days.  The same for y_obs, whose sorted wet values stay at the front of the row
// (the zeros in front of them are implied).  The mapped values go back through the compacted positions: the lane that owns a
// time slot re-walks its wet flags.  A wave whose segment has more than 64 * KC wet days -- or a tie among them, or a negative
// value -- hands the (tile, group) over: too many wet days to `worklist2` (bcsd_fxp_kernel<K, true, true> in list mode), the
// others to RANK / APPLY as before.
template <int K>
constexpr int compact_width() { return K == 20 ? 12 : K == 24 ? 16 : 0; }

template <int K, int KC, bool FULL>
__global__ void __launch_bounds__(kThreads, 4) bcsd_fxc_kernel(const Params) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ParamsPtr p = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NR = K / 2;
    using L = Lay<K>;
    double* const scratch = reinterpret_cast<double*>(smem_raw);
    int* const bad_cell = reinterpret_cast<int*>(scratch + 64 + 16);
    int* const redo_flag = bad_cell + kW;
    double* const tile = scratch + kHeadDoubles;
    const int RS = p->RS;
    if (threadIdx.x >= 32 && threadIdx.x < 32 + kW) bad_cell[threadIdx.x - 32] = 0;

    int64_t tile_id;
    int g;
    xcd_tile_of_block(blockIdx.x, p->ntiles, &tile_id, &g);
    g = nth_set_bit(p->gmask, g);
    if (tile_id >= p->ntiles || g < 0 || g >= p->G) return;

    const int64_t c0 = tile_id * kW;
    const int wave = __builtin_amdgcn_readfirstlane(tid_now() / kWave);
#define SD_LANE() const int lane = tid_now() % kWave
    const int64_t c = c0 + wave;
    const bool cell_ok = c < p->C;
    double* const row = tile + wave * RS;
    const unsigned rowb = lds_addr(row);
    const int begf = p->off_f[g];
    const int n = p->off_f[g + 1] - begf;  // == m (equal group lengths)
    const int begp = p->off_p[g];
    const int m = n;
    if (m == 0) return;  // an empty group (workgroup-uniform, like bcsd_fx_kernel / fxp_segment): its order-table entries belong to the next group
    const bool vec_f = (p->ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->y) & 15) == 0) &&
                       (p->X == nullptr || (reinterpret_cast<uintptr_t>(p->X) & 15) == 0);
    const bool vec_p = (p->ld_p % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->Xp) & 15) == 0);
    const bool cell_live = cell_ok && p->status_fit[cell_ok ? c : 0] == 0;
    const unsigned spare = rowb + 8u * (unsigned)(RS - 1);
    __syncthreads();

    // ---- x side: validation of x_hist, the x_fut tile ------------------------------------------------------------------
    {
        TileRegs<NR> xf;
        int tp[NR];
        rows_load<NR, FULL>(p->ord_p + begp, m, tp);
        if (p->X != nullptr) {
            TileRegs<NR> xh;
            int tf[NR];
            rows_load<NR, FULL>(p->ord_f + begf, n, tf);
            tile_issue_ti<NR>(p->X, p->ld, tf, c0, p->C, vec_f, xh);
            tile_issue_ti<NR>(p->Xp, p->ld_p, tp, c0, p->C, vec_p, xf);
            tile_check_finite<NR>(xh, n, c0, p->C, p->status_fit, bad_cell);
        } else {
            tile_issue_ti<NR>(p->Xp, p->ld_p, tp, c0, p->C, vec_p, xf);
        }
        tile_commit_sw<NR, K, FULL>(xf, m, c0, p->C, tile, RS, p->status_p, bad_cell);
    }
    __syncthreads();

    // Compaction of the lane's K samples v[] (all data or all pad) into row[0 .. wet): returns the wet count of the segment;
    // *zmask = the lane's zero flags (bit i: sample i is an exact zero), *base = wet samples in the lanes below
    auto compact = [&](const double (&v)[K], int len, int lane, unsigned* zmask, int* base) -> int {
        unsigned zm = 0u;
#pragma unroll
        for (int i = 0; i < K; ++i) zm |= v[i] == 0.0 ? 1u << i : 0u;
        // (positions past the segment count as dry: whole lanes, or -- not FULL -- the tail of the last lane)
        const int valid = len - K * lane;  // samples of this lane inside the segment (<= 0: none, >= K: all)
        zm |= valid >= K ? 0u : valid <= 0 ? (1u << K) - 1u : ((1u << K) - 1u) & ~((1u << valid) - 1u);
        const int cnt = K - __popc(zm);
        const int incl = wave_incl_scan_i32(cnt);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        *zmask = zm;
        *base = incl - cnt;
        if (total <= kWave * KC) {  // (wave-uniform) else the caller hands the segment over: nothing is moved
            wave_fence();  // every lane holds its samples: the row is free
            unsigned a = rowb + 8u * (unsigned)(incl - cnt);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const bool wet = ((zm >> i) & 1u) == 0u;
                lds_store_f64(wet ? a : spare, v[i]);
                a += wet ? 8u : 0u;
            }
            wave_fence();
        }
        return total;
    };

    unsigned kx[KC] = {};  // sorted keys of the wet x_fut samples: tag = compacted position of the sample with that rank
    unsigned zm_x = 0u;
    int base_x = 0, nw_x = 0;
    bool redo = false, many = false;
    TileRegs<NR> yt;  // the y_obs tile is requested ahead of the sort
    {
        SD_LANE();
        const bool has = K * lane < m;
        const double* ob = row + L::own(has ? lane : 0);
        double v[K];
#pragma unroll
        for (int i = 0; i < K; ++i) v[i] = ob[i];
        nw_x = compact(v, m, lane, &zm_x, &base_x);
        many = nw_x > kWave * KC;
        if (many && !(cell_live && bad_cell[wave] == 0)) {  // (a masked or non-finite cell: nothing of it is used; treat it as all dry)
            many = false;
            nw_x = 0;
            zm_x = (1u << K) - 1u;
        }
        {
            int ty[NR];
            rows_load<NR, FULL>(p->ord_f + begf, n, ty);
            tile_issue_ti<NR>(p->y, p->ld, ty, c0, p->C, vec_f, yt);
        }
        if (!many && nw_x > 0) {
            const double* cb = row + (KC * lane < nw_x ? KC * lane : 0);
            double vc[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) vc[i] = cb[i];
            const double lo = make_keys_plain<KC>(vc, nw_x, lane, kx);
            sdws::wave_sort<KC>(kx, lane, (nw_x + KC - 1) / KC);
            const bool tie = fix_equal_q<KC>(kx, rowb, lane) != 0;
            redo = (tie || lo < 0.0) && cell_live && bad_cell[wave] == 0;
        }
    }
    redo_flag[wave] = (redo ? 1 : 0) | (many ? 2 : 0);
    __syncthreads();
    int any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {
        if (threadIdx.x == 0) {
            if (any_redo & 1) {  // ties or negative values: RANK / APPLY
                const int slot = atomicAdd(p->work_count, 1);
                if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
            } else {  // too many wet days for the narrow sort: the K-wide kernel
                const int slot = atomicAdd(p->work_count2, 1);
                if (slot < p->work_cap) p->worklist2[slot] = tile_id * p->G + g;
            }
        }
        return;
    }

    // ---- y: climatology + sorted wet observations at the front of the row --------------------------------------------
    double yc = 0.0;
    int n0_y = 0;
    bool redo_y = false, many_y = false;
    {
        SD_LANE();
        tile_commit_sw<NR, K, FULL>(yt, n, c0, p->C, tile, RS, p->status_fit, nullptr);
        __syncthreads();
        const bool has = K * lane < n;
        const double* ob = row + L::own(has ? lane : 0);
        double v[K];
#pragma unroll
        for (int i = 0; i < K; ++i) v[i] = ob[i];
        double s = 0.0;
        if constexpr (FULL) {
#pragma unroll
            for (int i = 0; i < K; ++i) s += v[i];
            s = has ? s : 0.0;
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) s += K * lane + i < n ? v[i] : 0.0;
        }
        yc = wave_sum_f64(s) / (double)n;  // bcsd.py:138 (the same order of additions as bcsd_fxp_kernel)
        if (lane == 0 && cell_ok && p->return_anoms && yc <= 0.0) atomicOr(&p->status_fit[c], SDI_BAD_CLIMO);  // bcsd.py:140-141
        unsigned zm_y;
        int base_y;
        const int nw_y = compact(v, n, lane, &zm_y, &base_y);
        n0_y = n - nw_y;
        many_y = nw_y > kWave * KC && cell_live;
        if (nw_y <= kWave * KC) {
            unsigned ky[KC];
            const double* cb = row + (KC * lane < nw_y ? KC * lane : 0);
            double vc[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) vc[i] = cb[i];
            const double lo = make_keys_plain<KC>(vc, nw_y, lane, ky);
            sdws::wave_sort<KC>(ky, lane, (nw_y + KC - 1) / KC);
            redo_y = (lo < 0.0 || (fix_equal_q<KC>(ky, rowb, lane) & kUnsorted) != 0) && cell_live;  // tied observations are interchangeable
            double t[KC];
            const unsigned tm = KC * lane < nw_y ? kTagMask : 0u;  // (lanes past the wet days: pad keys, slot 0)
#pragma unroll
            for (int i = 0; i < KC; ++i) t[i] = lds_f64(rowb + 8u * (ky[i] & tm));
            wave_fence();  // all reads by tag done: row[r] = r-th smallest wet observation (np.sort, quantile.py:462)
            if (KC * lane < nw_y) {
#pragma unroll
                for (int i = 0; i < KC; ++i) lds_store_f64(KC * lane + i < nw_y ? rowb + 8u * (unsigned)(KC * lane + i) : spare, t[i]);
            }
            wave_fence();
        }
    }

    // ---- rank r of the predict sample -> r-th sorted observation (equal lengths: the inverse CDF at its own positions,
    //      quantile.py:523-545), zeros through rank n0 - 1 (quantile.py:488); ratio anomalies (bcsd.py:170-185) -------------
    int to[NR];
    {
        SD_LANE();
        const int n0_x = m - nw_x;
        // sorted observation of rank r: zero below n0_y, else the wet one at row[r - n0_y]
        auto ys = [&](int r) -> double {
            const double w = row[r >= n0_y ? r - n0_y : 0];
            return r >= n0_y ? w : 0.0;
        };
        double t[KC];
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            const int rw = KC * lane + i;  // rank among the wet days
            t[i] = ys(rw < nw_x ? n0_x + rw : 0);
        }
        const double val0 = ys(n0_x > 0 ? n0_x - 1 : 0);  // what every exact zero maps to
        wave_fence();  // every lane has read what it needs of the sorted observations
        if (KC * lane < nw_x) {
#pragma unroll
            for (int i = 0; i < KC; ++i) lds_store_f64(KC * lane + i < nw_x ? rowb + 8u * (kx[i] & kTagMask) : spare, t[i]);
        }
        wave_fence();
        // the lane that owns the time slots walks its wet flags through the compacted positions
        double q[K];
        {
            const double* cp = row + base_x;
            int pos = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const bool wet = ((zm_x >> i) & 1u) == 0u;
                const double wv = cp[wet ? pos : 0];
                q[i] = wet ? wv : val0;
                pos += wet ? 1 : 0;
            }
        }
        wave_fence();  // the compacted values are in registers: the time slots may be written
        const bool has = K * lane < m;
        rows_load<NR, FULL>(p->ord_p + begp, m, to);
        if (has) {
            double* ob = row + L::own(lane);
            const double rc = 1.0 / yc;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                double res = q[i];
                if (p->return_anoms) {
                    const double a = q[i] * rc;
                    res = __builtin_fma(__builtin_fma(-yc, a, q[i]), rc, a);
                    res = __builtin_isfinite(res) ? res : q[i] / yc;  // (zero or denormal climatology: the plain quotient)
                }
                ob[i] = res;
            }
        }
    }
    redo_flag[wave] = (redo_y ? 1 : 0) | (many_y ? 2 : 0);
    __syncthreads();
    any_redo = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) any_redo |= redo_flag[w];
    if (any_redo) {  // nothing has been written
        if (threadIdx.x == 0) {
            if (any_redo & 1) {
                const int slot = atomicAdd(p->work_count, 1);
                if (slot < p->work_cap) p->worklist[slot] = tile_id * p->G + g;
            } else {
                const int slot = atomicAdd(p->work_count2, 1);
                if (slot < p->work_cap) p->worklist2[slot] = tile_id * p->G + g;
            }
        }
        return;
    }
    const bool vec_o = (p->ld_out % 2 == 0) && ((reinterpret_cast<uintptr_t>(p->out) & 15) == 0);
    store_tile_ti<K, FULL>(p->out, p->ld_out, to, m, c0, p->C, vec_o, tile, RS);
#undef SD_LANE
}

template <int K, bool IDENT, bool FULL>
int launch_one(sd_ctx* ctx, const Params& p, size_t lds) {
    const int64_t tx = (p.ntiles + 7) / 8;
    const int64_t nblocks = 8 * tx * (p.gmask ? __builtin_popcountll(p.gmask) : p.G);
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "grid too large");
    if (p.kind == SD_BCSD_TAS) {
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fx_kernel<K, IDENT, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        SD_LAUNCH(ctx, FULL ? "bcsd_fx_kernel_full" : "bcsd_fx_kernel", (bcsd_fx_kernel<K, IDENT, FULL>), dim3((unsigned)nblocks),
                  dim3(kThreads), lds, p);
    } else {
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fxp_kernel<K, IDENT, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        SD_LAUNCH(ctx, FULL ? "bcsd_fxp_kernel_full" : "bcsd_fxp_kernel", (bcsd_fxp_kernel<K, IDENT, FULL>), dim3((unsigned)nblocks),
                  dim3(kThreads), lds, p);
    }
    return SD_OK;
}

template <int K, bool IDENT>
int launch_ki(sd_ctx* ctx, Params p, int nmax, const int* group_len) {
    p.RS = row_slots<K>(nmax);
    const size_t lds = ((size_t)kW * p.RS + kHeadDoubles) * sizeof(double);
    if (lds > ctx->lds_max) return sd_set_error(SD_ERR_UNSUPPORTED, "segment of %d samples needs %zu bytes of LDS", nmax, lds);
    // Groups whose segments are whole lanes of K samples (10 of the 12 months of a daily series at K = 20) take the FULL
    // instantiation -- no per-sample predicates --, the others a second launch of the general one.
    unsigned long long full = 0ull, rest = 0ull;
#ifndef SD_FX_NOFULL
    if (IDENT && group_len != nullptr && p.G <= 64 && sd_dev_env("SD_FX_NOFULL") == nullptr) {
        for (int g = 0; g < p.G; ++g) {
            const bool f = group_len[g] % K == 0 && group_len[g] >= full_min_len<K>();
            (f ? full : rest) |= 1ull << g;
        }
    }
#endif
#ifndef SD_FX_NOCOMPACT
    if constexpr (IDENT && compact_width<K>() != 0) {
        if (p.kind == SD_BCSD_PR && !p.from_state && p.work_count2 != nullptr && p.G <= 64 && group_len != nullptr &&
            sd_dev_env("SD_FX_NOCOMPACT") == nullptr) {
            // BcsdPrecipitation fit + predict: only the wet days are sorted; segments with too many of them for the narrow
            // network come back on the second list and take the K-wide kernel
            constexpr int KC = compact_width<K>();
            Params q = p;
            if ((full | rest) == 0ull) rest = p.G == 64 ? ~0ull : (1ull << p.G) - 1ull;  // (no split into whole-lane groups: all of them)
            if (full != 0ull) {
                q.gmask = full;
                const int64_t nb = 8 * ((p.ntiles + 7) / 8) * __builtin_popcountll(full);
                SD_CHECK_ARG(nb < ((int64_t)1 << 31), "grid too large");
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fxc_kernel<K, KC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                SD_LAUNCH(ctx, "bcsd_fxc_kernel_full", (bcsd_fxc_kernel<K, KC, true>), dim3((unsigned)nb), dim3(kThreads), lds, q);
            }
            if (rest != 0ull) {  // the months that are not whole lanes: the same kernel with its per-sample predicates
                q.gmask = rest;
                const int64_t nb = 8 * ((p.ntiles + 7) / 8) * __builtin_popcountll(rest);
                SD_CHECK_ARG(nb < ((int64_t)1 << 31), "grid too large");
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fxc_kernel<K, KC, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                SD_LAUNCH(ctx, "bcsd_fxc_kernel", (bcsd_fxc_kernel<K, KC, false>), dim3((unsigned)nb), dim3(kThreads), lds, q);
            }
            Params r = p;
            r.use_worklist = 2;  // (the general instantiation: segments of either kind may come back)
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fxp_kernel<K, IDENT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SD_LAUNCH(ctx, "bcsd_fxp_kernel_list", (bcsd_fxp_kernel<K, IDENT, false>), dim3((unsigned)(2 * (ctx->cu_count > 0 ? ctx->cu_count : 256))),
                      dim3(kThreads), lds, r);
            return SD_OK;
        }
    }
#endif
    if constexpr (IDENT && K == 20) {
        // Round 6: the whole-lane months of BcsdTemperature fit + predict take the kernel whose tiles land by LDS-DMA
        // (bcsd_fd_kernel) when the fields allow 16-byte requests of whole cell pairs and two workgroups still fit a CU
        int nfull = 0;
        for (int g = 0; g < p.G && group_len != nullptr && p.G <= 64; ++g)
            if ((full >> g) & 1ull) nfull = group_len[g] > nfull ? group_len[g] : nfull;
        const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const size_t lds_fd = fd_lds_bytes(nfull);
        const bool ok = full != 0ull && p.kind == SD_BCSD_TAS && !p.from_state && p.X != nullptr && p.C >= kW && p.C % 2 == 0 && p.ld % 2 == 0 &&
                        p.ld_p % 2 == 0 && p.ld_out % 2 == 0 && al16(p.X) && al16(p.y) && al16(p.Xp) && al16(p.out) && 2 * lds_fd <= ctx->lds_max &&
                        nfull % 20 == 0 && nfull > 640 && tmj::chunks_of(nfull) <= 80 && fd_late_chunks(fd_u2_stride(nfull)) <= 40 &&
                        tmj::chunks_of(nfull) - fd_late_chunks(fd_u2_stride(nfull)) <= 40 && sd_dev_env("SD_FX_NODMA") == nullptr;
        if (ok) {
            Params q = p;
            q.gmask = full;
            q.RS = fd_u2_stride(nfull);
            const int64_t nb = 8 * ((p.ntiles + 7) / 8) * __builtin_popcountll(full);
            SD_CHECK_ARG(nb < ((int64_t)1 << 31), "grid too large");
            if (sd_dev_env("SD_FD_LATE") != nullptr) {  // (development: the variant without the early half of the y tile)
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fd_kernel<K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fd));
                SD_LAUNCH(ctx, "bcsd_fd_kernel", (bcsd_fd_kernel<K, false>), dim3((unsigned)nb), dim3(kThreads), lds_fd, q);
            } else {
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcsd_fd_kernel<K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fd));
                SD_LAUNCH(ctx, "bcsd_fd_kernel", (bcsd_fd_kernel<K, true>), dim3((unsigned)nb), dim3(kThreads), lds_fd, q);
            }
            if (rest == 0ull) return SD_OK;
            q = p;
            q.gmask = rest;
            return launch_one<K, IDENT, false>(ctx, q, lds);
        }
    }
    if (IDENT && full != 0ull) {
        Params q = p;
        q.gmask = full;
        SD_TRY((launch_one<K, IDENT, IDENT>(ctx, q, lds)));  // (FULL exists for IDENT only)
        if (rest == 0ull) return SD_OK;
        q.gmask = rest;
        return launch_one<K, IDENT, false>(ctx, q, lds);
    }
    return launch_one<K, IDENT, false>(ctx, p, lds);
}

template <int K>
int launch_k(sd_ctx* ctx, const Params& p, int nmax, const int* gl) {
    return p.identity ? launch_ki<K, true>(ctx, p, nmax, gl) : launch_ki<K, false>(ctx, p, nmax, gl);
}

int launch_width(sd_ctx* ctx, const Params& p, int nmax, const int* gl) {
#ifdef SD_FX_ONLY_K  // development: one width only (fast compiles for ISA inspection)
    if (true) return launch_k<SD_FX_ONLY_K>(ctx, p, nmax, gl);
#else
    if (nmax <= 64 * 4) return launch_k<4>(ctx, p, nmax, gl);
    if (nmax <= 64 * 8) return launch_k<8>(ctx, p, nmax, gl);
    if (nmax <= 64 * 12) return launch_k<12>(ctx, p, nmax, gl);
    if (nmax <= 64 * 16) return launch_k<16>(ctx, p, nmax, gl);
    if (nmax <= 64 * 20) return launch_k<20>(ctx, p, nmax, gl);
    if (nmax <= 64 * 24) return launch_k<24>(ctx, p, nmax, gl);
#endif
    return sd_set_error(SD_ERR_UNSUPPORTED, "segment of %d samples exceeds the fused register-sort path", nmax);
}

}  // namespace sdfx

bool sd_bcsd_fx_supported(int nmax) { return nmax >= 1 && nmax <= 64 * 24; }

// One launch per call: the K = 20 kernel serves every month of a daily series (1 130 .. 1 240 samples), a narrower
// kernel only pays when the longest group allows it.
int sd_bcsd_fx_launch(sd_ctx* ctx, const sdrs::Params& p, int nmax, const int* group_len) {
    sdrs::Params q = p;
    q.gmask = 0ull;
    q.use_worklist = 0;
    if (q.n_endpoints <= 0) q.n_endpoints = 10;
    SD_TRY(sdfx::launch_width(ctx, q, nmax, group_len));
#ifdef SD_DEV
    if (const char* path = sd_dev_env("SD_FX_TRACE")) {  // raw phase clocks of the sampled workgroups -> file (tools/dev/trace_fx.py)
        SD_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<long long> h((size_t)sdfx::kTraceWgs * sdfx::kTraceSlots);
        SD_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(sdfx::sd_fx_trace), h.size() * sizeof(long long)));
        if (FILE* f = fopen(path, "wb")) {
            fwrite(h.data(), sizeof(long long), h.size(), f);
            fclose(f);
        }
    }
#endif
    return SD_OK;
}
