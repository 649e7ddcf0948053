// sd_analog_topk.h -- part of the translation unit csrc/sd_analog.hip (included there, inside its unnamed namespace, after
// sd_analog_fn.h; not a stand-alone header).  F > 1 predict for k <= 30: the slab scan of analog_slab_predict_kernel with the
// per-query top-k kept as a candidate list that is pruned by a register sorting network, instead of an LDS heap.
//
// Why: the heap costs one dependent chain of LDS round trips per inserted point (5 levels, 2 reads + 1 write each) and a wave
// spends as many rounds per chunk as its busiest lane has flagged points: 361 rounds of ~1 400 clocks per wave of 64 queries,
// two thirds of the kernel (F = 3, T = Tq = 14 600, k = 30).  Here a flagged point is *appended*: its exact reduced distance is
// recomputed from the staged coordinates (as before), the high word of that float64 -- a monotone 32-bit image of a
// non-negative double -- goes to the lane's column of new keys with its slot number in the 6 low bits, its position in the
// feature-0 order to a 16-bit column: two LDS writes, no read, no dependent chain.  When a lane's 32 new slots are full (or, at
// the end of a chunk, `prune_at` are in use) the wave *prunes*: the new keys are loaded into registers and sorted by a
// 191-comparator network (sd_wsort.h's Batcher generator), merged with the 32 kept keys -- which live in registers, sorted, from
// one prune to the next -- by the lower half of a bitonic merger (32 minima + 80 comparators), the positions of the 32 survivors
// are gathered through the slot bits and rewritten in sorted order, and the threshold tau becomes the upper edge of the k-th
// key's bucket.  ~700 vector instructions for 64 queries, every ~8 chunks.
//
// Exactness.  Keys are truncated (14 mantissa bits), so the order among the kept candidates is approximate -- but the *set* is
// exact: d1 <= d2 implies key(d1) <= key(d2), so a candidate whose key exceeds the k-th smallest key has k candidates strictly
// closer than itself and is in no k-nearest set, whatever the tie rule; tau = the largest double of the k-th key's bucket bounds
// every kept candidate from above, and a point is admitted with d <= tau.  The 32 kept slots hold every candidate of the k-th
// key's bucket unless the 32nd kept key is in that bucket too (three or more candidates within 6e-5 relative of the k-th
// distance, k <= 30): then the (cell, query batch) is handed back through a work list and analog_slab_predict_kernel answers it
// with the heap.  After the scan the exact reduced distances of the 32 survivors are recomputed (the arithmetic of
// chunk_mask), the pairs are put in exact (rdist, index) order by an insertion sort over the almost sorted list in LDS, and the
// first k go to the epilogue of the heap kernel (finish_query): same neighbours, same order, same statistics, bit for bit.

constexpr int kTopKeep = 32, kTopNew = 32;
constexpr int kTopMaxF = 6;                   // (7 and 8 features do not fit 256 registers with the matrix-core tiles: the heap kernel)
constexpr int kTopMaxK = 30;                  // two spare kept slots tell "every tie of the k-th bucket is here" from "maybe not"
constexpr unsigned kTopEmpty = 0xffffffc0u;   // keys of empty slots: above the high word of every finite double

__device__ __forceinline__ void topk_prune(unsigned (&kept)[kTopKeep], const unsigned* __restrict__ nk, uint16_t* __restrict__ bi,
                                           int& cnt, double& tau, int k, int lane, bool& overflow) {
    constexpr sdws::SortNet<kTopNew> snet{};
    constexpr sdws::BitonicNet<kTopKeep> bnet{};
    static_assert(kTopKeep == kTopNew, "the merge below pairs kept[i] with the reversed new keys");
    unsigned nw[kTopNew];
#pragma unroll
    for (int i = 0; i < kTopNew; ++i) nw[i] = i < cnt ? nk[i * 64 + lane] : (kTopEmpty | (unsigned)(kTopKeep + i));
    sdws::apply_net<kTopNew>(nw, snet);
    unsigned old[kTopKeep];
#pragma unroll
    for (int i = 0; i < kTopKeep; ++i) {
        old[i] = kept[i];
        kept[i] = sdws::umin(old[i], nw[kTopNew - 1 - i]);  // the 32 smallest of the 64, bitonic
    }
    sdws::apply_net<kTopKeep>(kept, bnet);
    uint16_t pos[kTopKeep];
#pragma unroll
    for (int i = 0; i < kTopKeep; ++i) pos[i] = bi[(kept[i] & 63u) * 64 + lane];
    unsigned kth = kTopEmpty;
#pragma unroll
    for (int i = 0; i < kTopKeep; ++i) kth = i == k - 1 ? kept[i] : kth;
    // (a lane's LDS operations execute in program order and touch its own column only: all reads above precede the writes)
#pragma unroll
    for (int i = 0; i < kTopKeep; ++i) {
        bi[i * 64 + lane] = pos[i];
        kept[i] = (kept[i] & ~63u) | (unsigned)i;
    }
    cnt = 0;
    if (kth < kTopEmpty) {  // k candidates so far
        tau = fmin(tau, __hiloint2double((int)(kth | 63u), (int)0xffffffffu));
        if ((kept[kTopKeep - 1] >> 6) == (kth >> 6)) {
            // the 32nd kept key shares the k-th key's bucket: was a key of that bucket dropped?  (rare: only now the smallest
            // dropped key -- the 33rd of the 64 -- is worked out)
            unsigned dropped = kTopEmpty;
#pragma unroll
            for (int i = 0; i < kTopKeep; ++i) dropped = sdws::umin(dropped, sdws::umax(old[i], nw[kTopNew - 1 - i]));
            overflow |= (dropped >> 6) == (kth >> 6);
        }
    }
}


// ---- float32 pre-filter of the scan -------------------------------------------------------------------------------------
// The scan needs one bit per (query, point): "can this point be within tau?".  The heap kernel answers it exactly with nine
// float64 instructions on wave-uniform coordinates fetched by scalar loads one group of points ahead -- and is bound by the
// latency of those loads (two waves per SIMD), not by the arithmetic.  Here the wave fetches the chunk after the current one with
// one coalesced vector load per feature (one point per lane, a whole chunk of latency to hide behind), rounds the current
// chunk's coordinates to float32 in registers and broadcasts point j with v_readlane_b32 -- no memory operation inside the
// mask loop -- into six float32 instructions (half the issue cost of float64 ones on gfx950), a subtraction and a v_alignbit
// that shifts the sign of (s - tauf) into the mask.  Only the flagged points -- a handful per chunk and lane -- see float64
// arithmetic (the append loop recomputes the exact distance).
// Conservative: with M >= every |coordinate| involved, |(q~ - x~) - (q - x)| <= 2^-23 M =: eta per feature, so the float32 sum s
// of squares satisfies  s <= d + 2 eta sqrt(F d) + F eta^2  up to its own rounding (< 2^-20 relative): every point with d <= tau
// has s < tauf := that bound at d = tau, inflated by 2^-19 and rounded up.  Magnitudes whose squares could leave the float32
// range (M >= 1e15) go to the heap kernel.
__device__ __forceinline__ float topk_tauf(double tau, double eta, int F) {
    if (!(tau >= 0.0)) return -1.0f;  // a lane without a query
    const double b = (tau + 2.0 * eta * sqrt((double)F * tau) + (double)F * eta * eta) * (1.0 + 0x1p-19) + 1e-37;
    const float t = (float)b;  // (round to nearest; +inf for tau = inf or an overflow: every finite sum is below it)
    return t < __builtin_inff() ? __uint_as_float(__float_as_uint(t) + 2u) : t;
}

// bit j of the result: the point lane j holds (mf) may be within tau of this lane's query
template <int F>
__device__ __forceinline__ unsigned long long lanes_mask_f32(const float (&mf)[F], const float (&qf)[F], float tauf) {
    unsigned half[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        float sacc = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float x = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(mf[f]), j));
            const float e = qf[f] - x;
            sacc = f == 0 ? e * e : __builtin_fmaf(e, e, sacc);
        }
        // sign of (s - tauf) shifted in: after 32 points the first one sits in bit 31
        half[j >> 5] = __builtin_amdgcn_alignbit(half[j >> 5], __float_as_uint(sacc - tauf), 31);
    }
    return (unsigned long long)__builtin_bitreverse32(half[0]) | ((unsigned long long)__builtin_bitreverse32(half[1]) << 32);
}

// pure_analog_stats (sd_analog_epilogue.h, gard.py:301-346) over register-resident analog values a[0..k) and the reduced
// distances sd[i * 64 + lane]: the same operations in the same order, loops unrolled over the N slots with i < k guards
template <int N>
__device__ __forceinline__ void pure_analog_stats_regs(const PredictArgs& pa, int k, int kind, int sample_i, const double (&a)[N],
                                                       const double* sd, int lane, double* pred, double* prob, double* err) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double sum = 0.0, wsum = 0.0, awsum = 0.0, asel = a[0];
    int nexc = 0;
    bool any_masked = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < k) {
            const bool exc = !pa.has_thresh || a[i] > pa.thresh;  // gard.py:307
            nexc += exc ? 1 : 0;
            any_masked |= !exc;
            sum += a[i];
            if (kind == SD_ANALOG_WEIGHT) {
                const double d = sqrt(sd[i * 64 + lane]);
                const double w = 1.0 / (d == 0.0 ? 1e-20 : d);  // gard.py:322-323
                wsum += w;
                awsum += a[i] * w;
            }
            asel = i == sample_i ? a[i] : asel;
        }
    }
    double p;
    if (kind == SD_ANALOG_BEST) p = a[0];                                   // gard.py:311
    else if (kind == SD_ANALOG_SAMPLE) p = asel;                            // gard.py:313-317
    else if (kind == SD_ANALOG_WEIGHT) p = any_masked ? nan : awsum / wsum;  // gard.py:319-327
    else p = any_masked ? nan : sum / (double)k;                            // gard.py:329-333
    if (pa.has_thresh) {
        p = nan_to_num(p);  // gard.py:341
        *prob = (double)nexc / (double)k;  // gard.py:343
    } else {
        *prob = 1.0;  // gard.py:346
    }
    if (any_masked) {
        *err = nan;
    } else {
        const double mean = sum / (double)k;
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i < k) {
                const double d = a[i] - mean;
                ss += d * d;
            }
        }
        *err = sqrt(ss / (double)k);  // ddof = 0 (gard.py:342,345)
    }
    *pred = p;
}

// cen[cl][f] = centre of the range of feature f over the training points of cell c_base + cl, pmax[cl] = the largest half-range:
// the float32 images are taken of (coordinate - centre), so a common offset of the data costs the pre-filter no precision
// (one workgroup per cell)
__global__ void __launch_bounds__(256) analog_slab_center_kernel(const double* __restrict__ ps, int64_t T, int F, int64_t cc,
                                                                 double* __restrict__ cen, double* __restrict__ pmax) {
    __shared__ double red[2][4];
    for (int64_t cl = blockIdx.x; cl < cc; cl += gridDim.x) {
        double half = 0.0;
        for (int f = 0; f < F; ++f) {
            const double* src = ps + (cl * F + f) * T;
            double lo = __longlong_as_double(0x7ff0000000000000ll), hi = -lo;
            for (int64_t i = threadIdx.x; i < T; i += blockDim.x) {
                const double v = src[i];
                lo = fmin(lo, v);
                hi = fmax(hi, v);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                lo = fmin(lo, __shfl_xor(lo, o, 64));
                hi = fmax(hi, __shfl_xor(hi, o, 64));
            }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) {
                red[0][threadIdx.x >> 6] = lo;
                red[1][threadIdx.x >> 6] = hi;
            }
            __syncthreads();
            lo = fmin(fmin(red[0][0], red[0][1]), fmin(red[0][2], red[0][3]));
            hi = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
            const double c = lo * 0.5 + hi * 0.5;  // (no overflow for finite data)
            if (threadIdx.x == 0) cen[cl * F + f] = c;
            half = fmax(half, fmax(hi - c, c - lo));
        }
        if (threadIdx.x == 0) pmax[cl] = half;
    }
}

// ---- the same bit from the matrix cores ------------------------------------------------------------------------------------
// 64 queries x 64 points x (F coordinates + 1) is a small matrix product: with x', q' the centred float32 images,
//     |q' - x'|^2 - T  =  sum_k X'[k][point] * Q'[k][query] + (|q'|^2 - T),   X' = (x'_0 .. x'_{F-1}, |x'|^2),  Q' = (-2 q'_0 .. -2 q'_{F-1}, 1)
// is what v_mfma_f32_32x32x2_f32 accumulates when the accumulator starts at (|q'|^2 - T): four 32 x 32 tiles (two point blocks x two
// query blocks), ceil((F + 1) / 2) instructions each, 64 cycles of the matrix pipe per instruction -- 512 cycles per chunk for
// F = 3 against ~2 100 cycles of vector issue for the v_readlane form -- and the sign bits of the 64 accumulators a lane ends up
// with are the mask bits (one v_alignbit each).  Operand layout (cdna_hip_programming.md): A[i = lane & 31][k = lane >> 5],
// B[k = lane >> 5][j = lane & 31], D: column j = lane & 31, row i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  Points are rows,
// queries columns: a lane holds, per tile, 16 of the 32 points of one query; the two lanes that share a query (l, l ^ 32)
// exchange their halves so that lane Q ends up with the 64 bits of query Q -- in the order the hardware produced them, see
// mfma_mask_point.  The matrix instruction is an fmaf chain, bit for bit, so its error obeys the usual dot-product bound:
//     computed <= |q - x|^2 - T + E_coord + E_mfma,
//     E_coord = 2 eta sqrt(F d) + F eta^2  (the rounding of the coordinates, eta = 2^-23 M),
//     E_mfma <= (K + 2) 2^-23 (4 F M^2 + T)  (K + 2 roundings of partial sums bounded by |x'|^2 + 2 |q'| |x'| + |q'|^2 + T), doubled here;
// T is tau plus both, inflated by 2^-19 and the accumulator start is rounded down: every point with d <= tau comes out negative.
typedef float topk_v16f __attribute__((ext_vector_type(16)));
template <int F>
struct TopkMfma {
    static constexpr int NKS = (F + 2) / 2;  // MFMA steps: F coordinates and the norm, two values of k per instruction
    static constexpr int KP = 2 * NKS;
};
__device__ __forceinline__ float float_down(float c, double exact) {  // the float32 at or below `exact` nearest to c = (float)exact
    if ((double)c <= exact) return c;
    const unsigned b = __float_as_uint(c);
    return c > 0.0f ? __uint_as_float(b - 1u) : (c < 0.0f ? __uint_as_float(b + 1u) : -1.4e-45f);
}
// accumulator start of the lane's own query: |q'|^2 - T(tau)
template <int F>
__device__ __forceinline__ float mfma_start(double tau, double nq, double eta, double m2) {
    if (!(tau >= 0.0)) return 1e30f;  // a lane without a query: never negative
    if (!(tau < 1e29)) return (float)(nq - 1e30);  // no bound yet: every point of the chunk (padding rows carry |x'|^2 = 3e38)
    const double t0 = tau + 2.0 * eta * sqrt((double)F * tau) + (double)F * eta * eta;
    const double em = (double)(TopkMfma<F>::KP + 2) * 0x1p-22 * (4.0 * (double)F * m2 + t0);
    const double t = fmin((t0 + em) * (1.0 + 0x1p-19) + 1e-37, 1e30);
    return float_down((float)(nq - t), nq - t);
}
// mask bit b of a lane -> point of the chunk (see the layout above): bits 0..31 come from the lane's own accumulators, 32..63 from
// lane ^ 32; inside a half: 16 bits per point block, register v = 15 - (b & 15) (v_alignbit shifts left)
__device__ __forceinline__ int mfma_mask_point(int b, int lane) {
    const int own_half = lane >> 5, src_half = (b >> 5) ? own_half ^ 1 : own_half;
    const int v = 15 - (b & 15);
    return (((b >> 4) & 1) << 5) | ((v >> 2) << 3) | (src_half << 2) | (v & 3);
}

// slab_pick with the two boundary values it tests held in the cursor: the value behind a side that moved is requested when the
// side moves and read a chunk later (slab_pick's own loads are an exposed round trip per chunk)
struct TopkCursor {
    int R, L, side;
    bool rdone, ldone;
    double pr, pl;  // P0[R] / P0[L - 1] while the side is open
};
__device__ __forceinline__ SlabChunk topk_pick(const double* __restrict__ P0, int T, double qlo, double qhi, double taumax,
                                               TopkCursor& cs) {
    if (!cs.rdone) {
        const double g = cs.pr - qhi;
        cs.rdone = g > 0.0 && g * g > taumax;
    }
    if (!cs.ldone) {
        const double g = qlo - cs.pl;
        cs.ldone = g > 0.0 && g * g > taumax;
    }
    SlabChunk ch{0, 0};
    if (cs.rdone && cs.ldone) return ch;
    const bool right = cs.rdone ? false : (cs.ldone ? true : cs.side == 0);
    cs.side ^= 1;
    if (right) {
        ch.j0 = cs.R;
        ch.nj = T - cs.R < 64 ? T - cs.R : 64;
        cs.R += ch.nj;
        cs.rdone = cs.R >= T;
        if (!cs.rdone) cs.pr = P0[cs.R];
    } else {
        ch.nj = cs.L < 64 ? cs.L : 64;
        ch.j0 = cs.L - ch.nj;
        cs.L = ch.j0;
        cs.ldone = cs.L <= 0;
        if (!cs.ldone) cs.pl = P0[cs.L - 1];
    }
    ch.j0 = __builtin_amdgcn_readfirstlane(ch.j0);
    ch.nj = __builtin_amdgcn_readfirstlane(ch.nj);
    return ch;
}

// (two waves per SIMD -- what the 20 KB of LDS per wave allow -- need the kernel inside 256 registers)
template <int F>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) analog_slab_topk_kernel(int mode, const double* __restrict__ qc /* [cc][F][Tq] */,
                                                              const int32_t* __restrict__ qi /* [cc][Tq] */, int64_t c_base,
                                                              int64_t Tq, int64_t T, int nbatch, const double* __restrict__ Xc,
                                                              const double* __restrict__ yc, const double* __restrict__ ps,
                                                              const int32_t* __restrict__ pi, const int32_t* __restrict__ fit_status,
                                                              int32_t* status, PredictArgs pa, int prune_at,
                                                              const double* __restrict__ cen /* [cc][F] */,
                                                              const double* __restrict__ pmax /* [cc] */, int use_mfma,
                                                              int32_t* __restrict__ worklist, int32_t* __restrict__ nwork,
                                                              unsigned long long* dbg) {
    typedef uint16_t IT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int k = pa.k, lane = threadIdx.x;
    // during the scan: new keys [32][64] u32, positions [64][64] u16, chunk coordinates [F][64]; afterwards the exact pairs
    unsigned* nk = reinterpret_cast<unsigned*>(smem_raw);
    IT* bi = reinterpret_cast<IT*>(nk + kTopNew * 64);
    double* stage = reinterpret_cast<double*>(bi + (kTopKeep + kTopNew) * 64);
    const int64_t cl = blockIdx.x / nbatch, c = c_base + cl;
    const int64_t slot = (int64_t)(blockIdx.x % nbatch) * 64 + lane;
    const bool active = fit_status[c] == 0, has_q = slot < Tq;
    const int64_t tq = has_q ? qi[cl * Tq + slot] : 0;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double q[F];
    float qf[F];
    bool ok = active && has_q;
    double mq = 0.0;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        q[f] = has_q ? qc[(cl * F + f) * Tq + tq] : 0.0;
        if (active && has_q && !sd_finite(q[f])) {
            atomicOr(&status[c], SDI_NONFINITE);
            ok = false;
        }
    }
    double cf[F];  // centre of the cell's training points, per feature (wave-uniform)
    double nq = 0.0;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        cf[f] = uniform_f64(cen[cl * F + f]);
        qf[f] = ok ? (float)(q[f] - cf[f]) : 0.0f;
        nq += (double)qf[f] * (double)qf[f];
        mq = fmax(mq, ok ? __builtin_fabs(q[f] - cf[f]) : 0.0);
    }
    const double mx = uniform_f64(wave_max_f64(fmax(mq, pmax[cl]))) * (1.0 + 0x1p-20);  // >= every |centred coordinate| of the wave's work
    if (!(mx < 1e12)) {  // squares beyond the float32 range: the heap kernel's batch
        if (lane == 0) worklist[atomicAdd(nwork, 1)] = (int32_t)blockIdx.x;
        return;
    }
    const double eta = mx * (0x1p-23 * (1.0 + 0x1p-10));
    const double m2 = mx * mx;
    constexpr int NKS = TopkMfma<F>::NKS, KP = TopkMfma<F>::KP;
    // matrix-core operands of the wave's queries: B[k = lane >> 5][j = lane & 31] of query block qb = Q'[2 ks + (lane >> 5)] of
    // query 32 qb + (lane & 31), fetched from the lane that owns that query
    float bop[2][NKS];
    {
        float qp[KP];
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) qp[kk] = kk < F ? -2.0f * qf[kk] : (kk == F ? 1.0f : 0.0f);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int src = 32 * qb + (lane & 31);
                const float t0 = __shfl(qp[2 * ks], src, 64), t1 = __shfl(qp[2 * ks + 1], src, 64);
                bop[qb][ks] = lane < 32 ? t0 : t1;
            }
    }
    topk_v16f cstart[2];  // accumulator starts of the two query blocks (every register: the start of the lane's column query)
    auto set_starts = [&](double tau_now) {
        const float own = mfma_start<F>(tau_now, nq, eta, m2);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float cs = __shfl(own, 32 * qb + (lane & 31), 64);
#pragma unroll
            for (int v = 0; v < 16; ++v) cstart[qb][v] = cs;
        }
    };
    const double* __restrict__ P = ps + c * F * T;  // [F][T], ascending in feature 0
    const int32_t* __restrict__ PI = pi + c * T;
    unsigned kept[kTopKeep];
#pragma unroll
    for (int i = 0; i < kTopKeep; ++i) kept[i] = kTopEmpty | (unsigned)i;
    int cnt = 0;
    bool overflow = false;
    double tau = ok ? inf : -1.0;  // bound of the k-th best distance; a lane without a query never flags a point
    float tauf = topk_tauf(tau, eta, F);
    if (use_mfma) set_starts(tau);
    long long tclk[3] = {0, 0, 0};  // (dbg: clocks in mask building, appends, prunes)
    // The queries of a wave ascend in feature 0 -- except where the wave straddles two classes of the query order
    // (analog_slab_s2_kernel): the second class starts over at its smallest q0, and one slab around both ranges would be the
    // whole training set.  Each ascending run of lanes is scanned on its own (one run in all but ~7 of a cell's 229 waves).
    unsigned long long breaks;
    {
        const double qprev = __shfl_up(q[0], 1, 64);
        const bool okprev = __shfl_up(ok ? 1 : 0, 1, 64) != 0;
        breaks = __builtin_amdgcn_ballot_w64(lane > 0 && ok && okprev && q[0] < qprev);
    }
    const int nruns = __builtin_popcountll(breaks) + 1;
    const int runid = __builtin_popcountll(breaks & ((2ull << lane) - 1ull));
    const bool ok_any = ok;
    for (int run = 0; run < nruns; ++run) {
    ok = ok_any && runid == run;
    const double tau_kept = tau;  // (lanes of other runs: no flags, no say in the slab)
    if (!ok) tau = -1.0;
    tauf = topk_tauf(tau, eta, F);
    if (use_mfma) set_starts(tau);
    if (__any(ok)) {
        const double qlo = uniform_f64(wave_min_f64(ok ? q[0] : inf));
        const double qhi = uniform_f64(wave_max_f64(ok ? q[0] : -inf));
        // start between the wave's queries: first sorted point >= the middle of their range, rounded down to 8 points
        const double qmid = qlo + (qhi - qlo) * 0.5;
        int lo = 0, hi = (int)T;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (P[mid] < qmid) lo = mid + 1; else hi = mid;
        }
        const int p8 = __builtin_amdgcn_readfirstlane(lo) & ~7;
        TopkCursor cs{p8, p8, 0, p8 >= (int)T, p8 <= 0, 0.0, 0.0};
        if (!cs.rdone) cs.pr = P[p8];
        if (!cs.ldone) cs.pl = P[p8 - 1];
        SlabChunk ch = topk_pick(P, (int)T, qlo, qhi, uniform_f64(wave_max_f64(tau)), cs);
        double mine[F];  // the chunk's points, one per lane
#pragma unroll
        for (int f = 0; f < F; ++f) mine[f] = lane < ch.nj ? P[(int64_t)f * T + ch.j0 + lane] : 0.0;
        while (ch.nj > 0) {
            // the chunk after this one is chosen with the thresholds known now (one chunk stale: it can only scan more than
            // necessary) and its points are requested before this chunk is looked at
            const SlabChunk nx = topk_pick(P, (int)T, qlo, qhi, uniform_f64(wave_max_f64(tau)), cs);
            double mnext[F];
#pragma unroll
            for (int f = 0; f < F; ++f) mnext[f] = lane < nx.nj ? P[(int64_t)f * T + nx.j0 + lane] : 0.0;
            const int j0 = ch.j0, nj = ch.nj;
            const long long t0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
            float mf[F];
#pragma unroll
            for (int f = 0; f < F; ++f) mf[f] = (float)(mine[f] - cf[f]);
            unsigned long long mask;
            if (use_mfma) {
                // A[i = lane & 31][k = lane >> 5] of point block pb = X'[2 ks + (lane >> 5)] of point 32 pb + (lane & 31): the lane's
                // own point or the one lane ^ 32 holds
                float xp[KP], sw[KP];
                double nx = 0.0;
#pragma unroll
                for (int f = 0; f < F; ++f) nx += (double)mf[f] * (double)mf[f];
#pragma unroll
                for (int kk = 0; kk < KP; ++kk) {
                    xp[kk] = kk < F ? (lane < nj ? mf[kk] : 0.0f) : (kk == F ? (lane < nj ? (float)nx : 3.0e38f) : 0.0f);
                    sw[kk] = __shfl_xor(xp[kk], 32, 64);
                }
                topk_v16f acc[2][2];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const float a0 = lane < 32 ? xp[2 * ks] : sw[2 * ks + 1];  // point block 0
                    const float a1 = lane < 32 ? sw[2 * ks] : xp[2 * ks + 1];  // point block 1
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        acc[0][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bop[qb][ks], ks == 0 ? cstart[qb] : acc[0][qb], 0, 0, 0);
                        acc[1][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bop[qb][ks], ks == 0 ? cstart[qb] : acc[1][qb], 0, 0, 0);
                    }
                }
                unsigned m16[2][2];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        unsigned m = 0u;
#pragma unroll
                        for (int v = 0; v < 16; ++v) m = __builtin_amdgcn_alignbit(m, __float_as_uint(acc[pb][qb][v]), 31);
                        m16[pb][qb] = m;
                    }
                const unsigned q0 = m16[0][0] | (m16[1][0] << 16), q1 = m16[0][1] | (m16[1][1] << 16);
                const unsigned keep = lane < 32 ? q0 : q1, send = lane < 32 ? q1 : q0;
                const unsigned recv = (unsigned)__shfl_xor((int)send, 32, 64);
                mask = (unsigned long long)keep | ((unsigned long long)recv << 32);
            } else {
                mask = lanes_mask_f32<F>(mf, qf, tauf);
                if (nj < 64) mask &= (1ull << nj) - 1ull;
            }
            const long long t1 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
            tclk[0] += t1 - t0;
            if (__builtin_amdgcn_ballot_w64(mask != 0ull) != 0ull) {
                __syncthreads();  // (one wave per workgroup: orders the LDS traffic)
#pragma unroll
                for (int f = 0; f < F; ++f) stage[f * 64 + lane] = mine[f];
                __syncthreads();
                if (dbg) {
                    int pc = __builtin_popcountll(mask);
                    for (int o = 32; o >= 1; o >>= 1) pc = max(pc, __shfl_xor(pc, o, 64));
                    if (lane == 0) atomicAdd(&dbg[1], (unsigned long long)pc);
                }
                for (;;) {
                    // flagged points two at a time: the staged coordinates of both are requested before either distance is
                    // formed (one chain of LDS round trip + nine dependent float64 operations per point otherwise)
                    while (mask != 0ull && cnt < kTopNew - 1) {
                        const int b1 = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const bool two = mask != 0ull;
                        const int b2 = two ? __builtin_ctzll(mask) : b1;
                        mask = two ? (mask & (mask - 1)) : mask;
                        const int j1 = use_mfma ? mfma_mask_point(b1, lane) : b1, j2 = use_mfma ? mfma_mask_point(b2, lane) : b2;
                        double x1[F], x2[F];
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            x1[f] = stage[f * 64 + j1];
                            x2[f] = stage[f * 64 + j2];
                        }
                        double d1 = 0.0, d2 = 0.0;
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            const double e1 = q[f] - x1[f], e2 = q[f] - x2[f];
                            d1 += e1 * e1;
                            d2 += e2 * e2;
                        }
                        if (d1 <= tau) {  // (tau may have tightened since the mask was built; the mask itself is conservative)
                            nk[cnt * 64 + lane] = ((unsigned)__double2hiint(d1) & ~63u) | (unsigned)(kTopKeep + cnt);
                            bi[(kTopKeep + cnt) * 64 + lane] = (IT)(j0 + j1);
                            ++cnt;
                        }
                        if (two && d2 <= tau) {
                            nk[cnt * 64 + lane] = ((unsigned)__double2hiint(d2) & ~63u) | (unsigned)(kTopKeep + cnt);
                            bi[(kTopKeep + cnt) * 64 + lane] = (IT)(j0 + j2);
                            ++cnt;
                        }
                    }
                    if (mask != 0ull && cnt < kTopNew) {  // one free slot
                        const int b = __builtin_ctzll(mask);
                        const int j = use_mfma ? mfma_mask_point(b, lane) : b;
                        mask &= mask - 1;
                        double d = 0.0;
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            const double e = q[f] - stage[f * 64 + j];
                            d += e * e;
                        }
                        if (d <= tau) {
                            nk[cnt * 64 + lane] = ((unsigned)__double2hiint(d) & ~63u) | (unsigned)(kTopKeep + cnt);
                            bi[(kTopKeep + cnt) * 64 + lane] = (IT)(j0 + j);
                            ++cnt;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(mask != 0ull) == 0ull) break;
                    const long long t2 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    topk_prune(kept, nk, bi, cnt, tau, k, lane, overflow);  // a lane ran out of slots with points left
                    tauf = topk_tauf(tau, eta, F);
                    if (use_mfma) set_starts(tau);
                    if (dbg) {
                        const long long dt = (long long)__builtin_amdgcn_s_memtime() - t2;
                        tclk[2] += dt;
                        tclk[1] -= dt;
                        if (lane == 0) atomicAdd(&dbg[2], 1ull);
                    }
                }
            }
            const long long t3 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
            tclk[1] += t3 - t1;
            if (__builtin_amdgcn_ballot_w64(cnt >= prune_at) != 0ull) {
                topk_prune(kept, nk, bi, cnt, tau, k, lane, overflow);
                tauf = topk_tauf(tau, eta, F);
                if (use_mfma) set_starts(tau);
                if (dbg) {
                    tclk[2] += (long long)__builtin_amdgcn_s_memtime() - t3;
                    if (lane == 0) atomicAdd(&dbg[2], 1ull);
                }
            }
            if (dbg && lane == 0) atomicAdd(&dbg[0], 1ull);
#pragma unroll
            for (int f = 0; f < F; ++f) mine[f] = mnext[f];
            ch = nx;
        }
        if (__builtin_amdgcn_ballot_w64(cnt > 0) != 0ull) topk_prune(kept, nk, bi, cnt, tau, k, lane, overflow);
    }
    if (!ok) tau = tau_kept;
    }
    ok = ok_any;
    if (__builtin_amdgcn_ballot_w64(overflow) != 0ull) {  // ties beyond the kept slots: the heap kernel answers this batch
        if (lane == 0) worklist[atomicAdd(nwork, 1)] = (int32_t)blockIdx.x;
        return;
    }
    // ---- exact pairs of the survivors, exact order, epilogue ----
    const long long t4 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    IT pos[kTopKeep];
#pragma unroll
    for (int i = 0; i < kTopKeep; ++i) pos[i] = bi[i * 64 + lane];
    __syncthreads();
    double* sd = reinterpret_cast<double*>(smem_raw);              // [32][64]
    IT* si = reinterpret_cast<IT*>(sd + (size_t)kTopKeep * 64);    // [32][64]
#pragma unroll
    for (int i = 0; i < kTopKeep; ++i) {
        const bool real = kept[i] < kTopEmpty;
        const int j = real ? (int)pos[i] : 0;
        double d = 0.0;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const double df = q[f] - P[(int64_t)f * T + j];
            d += df * df;
        }
        const int32_t idx = PI[j];
        sd[i * 64 + lane] = real ? d : inf;
        si[i * 64 + lane] = real ? (IT)idx : (IT)0xffffu;
    }
    {
        double pd = sd[lane];
        IT pidx = si[lane];
        for (int i = 1; i < kTopKeep; ++i) {
            const double d = sd[i * 64 + lane];
            const IT ix = si[i * 64 + lane];
            if (pair_gt<IT>(pd, pidx, d, ix)) {  // out of order (a truncated-key tie): sink it; the largest so far stays at i
                int p = i;
                while (p > 0) {
                    const double dq = sd[(p - 1) * 64 + lane];
                    const IT iq = si[(p - 1) * 64 + lane];
                    if (!pair_gt<IT>(dq, iq, d, ix)) break;
                    sd[p * 64 + lane] = dq;
                    si[p * 64 + lane] = iq;
                    --p;
                }
                sd[p * 64 + lane] = d;
                si[p * 64 + lane] = ix;
            } else {
                pd = d;
                pidx = ix;
            }
        }
    }
    const long long t5 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    if (mode == 0) {
        // PureAnalog: the analog values of the k neighbours are fetched at once (finish_query reads them one dependent round
        // trip after the other, twice) and the statistics run over registers: pure_analog_stats's operations in its order
        const double* __restrict__ ycell = yc + c * T;
        double a[kTopKeep];
#pragma unroll
        for (int i = 0; i < kTopKeep; ++i) a[i] = (ok && i < k) ? ycell[si[i * 64 + lane]] : 0.0;
        if (has_q) {
            double pred, prob = 1.0, err;
            if (!ok) {
                pred = prob = err = __longlong_as_double(0x7ff8000000000000ll);
            } else {
                const int smp = (pa.kind == SD_ANALOG_SAMPLE && pa.sample) ? pa.sample[tq * pa.ld_s + c] : 0;
                pure_analog_stats_regs<kTopKeep>(pa, k, pa.kind, smp < 0 ? 0 : (smp >= k ? k - 1 : smp), a, sd, lane, &pred, &prob, &err);
            }
            put_out(pa, tq, c, pred, prob, err);
            if (ok && pa.inds)
                for (int i = 0; i < k; ++i) pa.inds[(tq * k + i) * pa.ld_out + c] = si[i * 64 + lane];
            if (ok && pa.dist)
                for (int i = 0; i < k; ++i) pa.dist[(tq * k + i) * pa.ld_out + c] = sqrt(sd[i * 64 + lane]);
        }
    } else if (has_q) {
        finish_query(mode, pa, F, T, c, tq, q, Xc + c * F * T, yc + c * T, sd, si, 64, ok);
    }
    if (dbg && lane == 0) {
        atomicAdd(&dbg[4], (unsigned long long)tclk[0]);
        atomicAdd(&dbg[5], (unsigned long long)tclk[1]);
        atomicAdd(&dbg[6], (unsigned long long)tclk[2]);
        atomicAdd(&dbg[7], (unsigned long long)(t5 - t4));
        atomicAdd(&dbg[8], (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - t5));
    }
}

size_t topk_lds_bytes(int F) {
    const size_t scan = sizeof(unsigned) * kTopNew * 64 + sizeof(uint16_t) * (kTopKeep + kTopNew) * 64 + sizeof(double) * (size_t)F * 64;
    const size_t fin = (sizeof(double) + sizeof(uint16_t)) * (size_t)kTopKeep * 64;
    return scan > fin ? scan : fin;
}

// the fast kernel over the chunk's cells, then the heap kernel over what it handed back
template <int F>
int launch_slab_topk(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* qc, const int32_t* qi, int64_t cb, int64_t cc,
                     int64_t Tq, int32_t* status_p, const PredictArgs& pa, int32_t* worklist /* [cc * nbatch + 1] */,
                     double* cen /* [cc][F] */, double* pmax /* [cc] */) {
    const size_t lds = topk_lds_bytes(F);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_slab_topk_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t nbatch = (Tq + 63) / 64, nblocks = cc * nbatch;
    const char* epa = sd_dev_env("SD_TOPK_PRUNE_AT");
    int prune_at = epa ? atoi(epa) : 16;
    prune_at = prune_at < 1 ? 1 : (prune_at > kTopNew ? kTopNew : prune_at);
    const char* eab = sd_dev_env("SD_ANALOG_ABLATE");  // 4: count chunks, append rounds, prunes
    const bool count = eab && (atoi(eab) & 4);
    sd_scratch dbg;
    if (count) {
        SD_HIP(dbg.alloc(ctx, 128));
        SD_HIP(hipMemsetAsync(dbg.p, 0, 128, ctx->stream));
    }
    SD_LAUNCH(ctx, "analog_slab_center_kernel", analog_slab_center_kernel, dim3((unsigned)std::min<int64_t>(cc, (int64_t)ctx->cu_count * 8)),
              dim3(256), 0, (const double*)st->ps + cb * F * st->T, st->T, F, cc, cen, pmax);
    const int use_mfma = sd_dev_env("SD_TOPK_READLANE") == nullptr ? 1 : 0;  // (A/B: the v_readlane form of the pre-filter)
    int32_t* nwork = worklist + nblocks;
    SD_HIP(hipMemsetAsync(nwork, 0, sizeof(int32_t), ctx->stream));
    SD_LAUNCH(ctx, "analog_slab_topk_kernel", (analog_slab_topk_kernel<F>), dim3((unsigned)nblocks), dim3(64), lds, mode, qc, qi, cb,
              Tq, st->T, (int)nbatch, (const double*)st->X, (const double*)st->y, (const double*)st->ps, (const int32_t*)st->xi,
              (const int32_t*)st->status, status_p, pa, prune_at, (const double*)cen, (const double*)pmax, use_mfma, worklist, nwork,
              count ? dbg.as<unsigned long long>() : nullptr);
    int32_t nw = 0;
    SD_HIP(hipMemcpyAsync(&nw, nwork, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    if (count) {
        unsigned long long h[16];
        SD_HIP(hipMemcpy(h, dbg.p, 128, hipMemcpyDeviceToHost));
        fprintf(stderr, "[topk] waves %lld: chunks/wave %.1f append rounds/wave %.1f prunes/wave %.1f handed back %d\n", (long long)nblocks,
                (double)h[0] / (double)nblocks, (double)h[1] / (double)nblocks, (double)h[2] / (double)nblocks, nw);
        fprintf(stderr, "[topk] clocks/wave (s_memtime): masks %.0f appends %.0f prunes %.0f exact finish %.0f epilogue %.0f\n",
                (double)h[4] / (double)nblocks, (double)h[5] / (double)nblocks, (double)h[6] / (double)nblocks,
                (double)h[7] / (double)nblocks, (double)h[8] / (double)nblocks);
    }
    if (nw > 0) SD_TRY((launch_slab<F>(ctx, mode, st, qc, qi, cb, cc, Tq, status_p, pa, worklist, nw)));
    return SD_OK;
}

// F > 1 with the feature-0 sorted copy: queries go cell-major, are sorted by feature 0 per cell, and every wave scans
// only the slab of training points its 64 neighbouring queries can reach (analog_slab_predict_kernel)
int predict_slab(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int32_t* status_p,
                 const PredictArgs& pa) {
    const int F = st->F;
    const int64_t C = st->C, nbatch = (Tq + 63) / 64;
    int64_t chunk = 4096;
    while (chunk > 1 && chunk * nbatch >= ((int64_t)1 << 31)) chunk >>= 1;
    const int64_t cc_max = C < chunk ? C : chunk;
    const int Kq = sort2_width(Tq, ctx->lds_max);
    // classes of the query order (analog_slab_s2_kernel); a short series would only get waves that straddle classes
    const char* ecl = sd_dev_env("SD_ANALOG_SLAB_CLASSES");
    int nclass = ecl ? atoi(ecl) : (int)std::min<int64_t>(8, Tq / 512);
    nclass = nclass < 1 ? 1 : (nclass > 8 ? 8 : nclass);
    // k <= 30: candidate lists pruned by a register sorting network (analog_slab_topk_kernel); the heap kernel takes larger k
    // and the batches the fast kernel hands back
    const bool topk = pa.k <= kTopMaxK && F <= kTopMaxF && sd_dev_env("SD_ANALOG_HEAP") == nullptr;
    sd_scratch qc, qs, qi, key, wl;
    sd_scratch pmax, cen;
    if (topk) {
        SD_HIP(wl.alloc(ctx, sizeof(int32_t) * (size_t)(cc_max * nbatch + 1)));
        SD_HIP(pmax.alloc(ctx, sizeof(double) * (size_t)cc_max));
        SD_HIP(cen.alloc(ctx, sizeof(double) * (size_t)cc_max * F));
    }
    if (nclass > 1) SD_HIP(key.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
    SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tq * F * cc_max));
    SD_HIP(qs.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
    SD_HIP(qi.alloc(ctx, sizeof(int32_t) * (size_t)Tq * cc_max));
    for (int64_t cb = 0; cb < C; cb += chunk) {
        const int64_t cc = C - cb < chunk ? C - cb : chunk;
        dim3 tgrid((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32));
        for (int f = 0; f < F; ++f)
            SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, tgrid, dim3(256), 0, Xq + cb, ld, Tq, F, f, cc,
                      qc.as<double>(), status_p + cb, 0);
        const int nbk = (int)std::min<int64_t>(cc, (int64_t)ctx->cu_count * 8);
        Sort2Args a{qc.as<double>(), (int64_t)F * Tq, 1, nullptr, Tq, cc, qs.as<double>(), qi.as<int32_t>(),
                    nullptr, nullptr, nullptr};
        if (nclass > 1) {
            SD_LAUNCH(ctx, "analog_slab_s2_kernel", analog_slab_s2_kernel, dim3(nbk), dim3(256), 0, (const double*)qc.p, Tq, F, cc,
                      key.as<double>());
            a.X = key.as<double>();
            a.x_stride = Tq;
            SD_TRY(launch_sort2_width(ctx, Kq, a));  // qs = sorted s2 (class thresholds)
            SD_LAUNCH(ctx, "analog_slab_key_kernel", analog_slab_key_kernel, dim3(nbk), dim3(256), 0, (const double*)qc.p,
                      (const double*)qs.p, Tq, F, cc, nclass, key.as<double>());
        }
        SD_TRY(launch_sort2_width(ctx, Kq, a));  // qi = query order
        const double* q = qc.as<double>();
        const int32_t* qix = qi.as<int32_t>();
        if (topk) {
            int32_t* w = wl.as<int32_t>();
            switch (F) {
                case 2: SD_TRY(launch_slab_topk<2>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa, w, cen.as<double>(), pmax.as<double>())); break;
                case 3: SD_TRY(launch_slab_topk<3>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa, w, cen.as<double>(), pmax.as<double>())); break;
                case 4: SD_TRY(launch_slab_topk<4>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa, w, cen.as<double>(), pmax.as<double>())); break;
                case 5: SD_TRY(launch_slab_topk<5>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa, w, cen.as<double>(), pmax.as<double>())); break;
                default: SD_TRY(launch_slab_topk<6>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa, w, cen.as<double>(), pmax.as<double>())); break;
            }
            continue;
        }
        switch (F) {
            case 2: SD_TRY(launch_slab<2>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 3: SD_TRY(launch_slab<3>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 4: SD_TRY(launch_slab<4>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 5: SD_TRY(launch_slab<5>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 6: SD_TRY(launch_slab<6>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            case 7: SD_TRY(launch_slab<7>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
            default: SD_TRY(launch_slab<8>(ctx, mode, st, q, qix, cb, cc, Tq, status_p, pa)); break;
        }
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));  // the staging buffers go back to the block cache at scope exit
    return SD_OK;
}
