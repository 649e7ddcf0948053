// sd_analog_fn.h -- part of the translation unit csrc/sd_analog.hip (included there, inside its unnamed namespace; not a
// stand-alone header).  general F predict kernels: LDS-staged brute force, wave scanner, feature-0 slab search, and their launchers.

// ------------------------------------------------------------------------------------------------
// general F predict: brute force, training rows staged through LDS, per-thread top-k in scratch
// ------------------------------------------------------------------------------------------------
constexpr int kBfThreads = 256;
constexpr int kBfChunk = 1024;

__global__ void __launch_bounds__(kBfThreads) analog_bf_predict_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                      int64_t Tq, int64_t T, int F, int64_t C,
                                                                      const double* __restrict__ Xc,
                                                                      const double* __restrict__ yc,
                                                                      const int32_t* __restrict__ fit_status,
                                                                      int32_t* status, double* scratch_d,
                                                                      int32_t* scratch_i, PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xt = reinterpret_cast<double*>(smem_raw);  // [F][kBfChunk]
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int k = pa.k;
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    int64_t step, end;
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const double* Xcell = Xc + c * F * T;
        for (int64_t tq0 = 0; tq0 < Tq; tq0 += nthr) {
            const int64_t tq = tq0 + tid;
            const bool has_q = tq < Tq;
            double q[kMaxF];
            bool ok = active && has_q;
            for (int f = 0; f < F; ++f) {
                q[f] = has_q ? Xq[(tq * F + f) * ld + c] : 0.0;
                if (active && has_q && !sd_finite(q[f])) {
                    atomicOr(&status[c], SDI_NONFINITE);
                    ok = false;
                }
            }
            // unsorted top-k with tracked worst element
            for (int i = 0; i < k; ++i) {
                sd[(int64_t)i * nthr + tid] = inf;
                si[(int64_t)i * nthr + tid] = 0x7fffffff;
            }
            double worst_d = inf;
            int worst_i = 0x7fffffff, worst_slot = 0;
            for (int64_t j0 = 0; j0 < T; j0 += kBfChunk) {
                const int nj = (int)((T - j0) < kBfChunk ? (T - j0) : kBfChunk);
                __syncthreads();
                if (active)
                    for (int i = tid; i < nj * F; i += nthr) {
                        const int f = i / nj, j = i - f * nj;
                        xt[f * kBfChunk + j] = Xcell[(int64_t)f * T + j0 + j];
                    }
                __syncthreads();
                if (!ok) continue;
                for (int j = 0; j < nj; ++j) {
                    double d = 0.0;
                    for (int f = 0; f < F; ++f) {
                        const double df = q[f] - xt[f * kBfChunk + j];
                        d += df * df;
                    }
                    // ascending j: an equal distance with a larger index never displaces
                    if (d < worst_d) {
                        sd[(int64_t)worst_slot * nthr + tid] = d;
                        si[(int64_t)worst_slot * nthr + tid] = (int32_t)(j0 + j);
                        worst_d = -1.0;
                        worst_i = -1;
                        for (int i = 0; i < k; ++i) {
                            const double di = sd[(int64_t)i * nthr + tid];
                            const int ii = si[(int64_t)i * nthr + tid];
                            if (di > worst_d || (di == worst_d && ii > worst_i)) {
                                worst_d = di;
                                worst_i = ii;
                                worst_slot = i;
                            }
                        }
                    }
                }
            }
            if (ok) {
                // selection sort into ascending (rdist, index)
                for (int i = 0; i < k - 1; ++i) {
                    int best = i;
                    double bd = sd[(int64_t)i * nthr + tid];
                    int bi = si[(int64_t)i * nthr + tid];
                    for (int j = i + 1; j < k; ++j) {
                        const double dj = sd[(int64_t)j * nthr + tid];
                        const int ij = si[(int64_t)j * nthr + tid];
                        if (dj < bd || (dj == bd && ij < bi)) {
                            best = j;
                            bd = dj;
                            bi = ij;
                        }
                    }
                    if (best != i) {
                        sd[(int64_t)best * nthr + tid] = sd[(int64_t)i * nthr + tid];
                        si[(int64_t)best * nthr + tid] = si[(int64_t)i * nthr + tid];
                        sd[(int64_t)i * nthr + tid] = bd;
                        si[(int64_t)i * nthr + tid] = bi;
                    }
                }
            }
            if (has_q) finish_query(mode, pa, F, T, c, tq, q, Xcell, yc + c * T, sd, si, nthr, ok);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// general F predict, second form: one wave per (cell, 64 queries).  Every lane owns one query and keeps its
// k best (rdist, index) pairs sorted in LDS ([k][64]: lane-contiguous, conflict-free).  The training set is
// scanned in chunks of 64 points whose coordinates are wave-uniform (scalar loads, no LDS staging, no
// barriers); a chunk first yields a 64-bit mask of points closer than the lane's current k-th distance,
// then only the flagged points are inserted.  After the first few chunks the mask is almost always empty,
// so the steady state is 3F+3 vector instructions per (query, training point).
// ------------------------------------------------------------------------------------------------
// (d, idx) pairs of one lane as a binary max-heap in LDS ([k][64]): the root is the worst of the k best.
template <typename IT>
__device__ __forceinline__ bool pair_gt(double da, IT ia, double db, IT ib) { return da > db || (da == db && ia > ib); }

// place (d, idx) at the root and sift it down within the first `n` entries
template <typename IT>
__device__ __forceinline__ void heap_replace_root(double* sd, IT* si, int lane, int n, double d, IT idx) {
    int pos = 0;
    for (;;) {
        const int l = 2 * pos + 1;
        if (l >= n) break;
        const int r = l + 1;
        double dc = sd[l * 64 + lane];
        IT ic = si[l * 64 + lane];
        int ch = l;
        if (r < n) {
            const double dr = sd[r * 64 + lane];
            const IT ir = si[r * 64 + lane];
            if (pair_gt<IT>(dr, ir, dc, ic)) {
                dc = dr;
                ic = ir;
                ch = r;
            }
        }
        if (!pair_gt<IT>(dc, ic, d, idx)) break;
        sd[pos * 64 + lane] = dc;
        si[pos * 64 + lane] = ic;
        pos = ch;
    }
    sd[pos * 64 + lane] = d;
    si[pos * 64 + lane] = idx;
}

// One 64-point chunk [j0, j0 + nj) of a cell's training set (P: [F][T]) against the 64 queries of the wave.
//  * the coordinates (and, for the sorted copy, the original indices PI) of the chunk are also fetched one point per
//    lane at the start and parked in LDS (stage / stage_i) once the mask is built: the insertion loop, where every
//    lane looks at a different point, then reads LDS instead of paying a global-memory round trip per candidate
//    (that latency, not the 3F+3 instructions per pair, used to dominate these kernels);
//  * full chunks build the mask from wave-uniform coordinates (scalar loads) in groups of G points, the next group
//    requested before the current one is used; the last request fetches the first group of the chunk the caller will
//    scan next (next_j0), so that `cur` is ready on entry.
// SORTED: points arrive in feature-0 order, equal distances are decided by the index comparison with the heap root;
// otherwise they arrive in index order and an equal distance never displaces.
// points per group of wave-uniform coordinates: two groups (current + requested) of F x G doubles must fit the ~100
// scalar registers, or they spill into vector-register lanes inside the scan loop
template <int F>
constexpr int kScanG = F <= 2 ? 8 : 4;
// The chunk's points closer to the lane's query than tau (INCL: or as close), bit j = point j0 + j.  Full chunks take their
// coordinates from `cur` (wave-uniform groups, see scan_chunk) and leave the first group of the chunk at next_j0 there.
template <int F, bool INCL>
__device__ __forceinline__ unsigned long long chunk_mask(const double* __restrict__ P, int64_t T, int j0, int nj, int next_j0,
                                                         double (&cur)[F][kScanG<F>], const double (&q)[F], double tau) {
    constexpr int G = kScanG<F>;
    unsigned long long mask = 0ull;
    if (nj == 64) {
        double nxt[F][G];
#pragma unroll
        for (int jg = 0; jg < 64; jg += G) {
            const int jn = jg + G < 64 ? j0 + jg + G : next_j0;
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int g = 0; g < G; ++g) nxt[f][g] = P[(int64_t)f * T + jn + g];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                double d = 0.0;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const double df = q[f] - cur[f][g];
                    d += df * df;
                }
                const bool hit = INCL ? d <= tau : d < tau;
                mask |= hit ? (1ull << (jg + g)) : 0ull;
            }
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int g = 0; g < G; ++g) cur[f][g] = nxt[f][g];
        }
    } else {
        for (int j = 0; j < nj; ++j) {
            double d = 0.0;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const double df = q[f] - P[(int64_t)f * T + j0 + j];
                d += df * df;
            }
            const bool hit = INCL ? d <= tau : d < tau;
            mask |= hit ? (1ull << j) : 0ull;
        }
    }
    return mask;
}

template <int F, typename IT, bool SORTED>
__device__ __forceinline__ void scan_chunk(const double* __restrict__ P, int64_t T, const int32_t* __restrict__ PI, int j0, int nj,
                                           int next_j0, double (&cur)[F][kScanG<F>], const double (&q)[F], double& tau, double* sd,
                                           IT* si, int k, int lane, double* stage /* [F][64] */, int32_t* stage_i /* [64] */,
                                           int ablate = 0, unsigned long long* dbg = nullptr) {
    double mine[F];
    int32_t mine_i = 0;
#pragma unroll
    for (int f = 0; f < F; ++f) mine[f] = lane < nj ? P[(int64_t)f * T + j0 + lane] : 0.0;
    if (SORTED) mine_i = lane < nj ? PI[j0 + lane] : 0;
    unsigned long long mask = chunk_mask<F, SORTED>(P, T, j0, nj, next_j0, cur, q, tau);
    if (ablate & 1) mask = 0ull;  // (timing experiments: no insertions)
    if (__builtin_amdgcn_ballot_w64(mask != 0ull) == 0ull) return;  // (the usual case far from the queries)
    __syncthreads();  // one wave per workgroup: orders the LDS traffic of the previous chunk's insertions
#pragma unroll
    for (int f = 0; f < F; ++f) stage[f * 64 + lane] = mine[f];
    if (SORTED) stage_i[lane] = mine_i;
    __syncthreads();
    if (dbg) {  // (experiments: insertion rounds = largest number of flagged points of a lane)
        int pc = __builtin_popcountll(mask);
        for (int o = 32; o >= 1; o >>= 1) pc = max(pc, __shfl_xor(pc, o, 64));
        if (lane == 0) atomicAdd(&dbg[1], (unsigned long long)pc);
    }
    while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        double d = 0.0;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const double df = q[f] - stage[f * 64 + j];
            d += df * df;
        }
        if (SORTED) {
            if (d <= tau) {  // tau may have tightened since the mask was built
                const IT idx = (IT)stage_i[j];
                if (d < tau || idx < si[lane]) {  // equal distance: the smaller training index is the better pair
                    heap_replace_root<IT>(sd, si, lane, k, d, idx);
                    tau = sd[lane];
                }
            }
        } else if (d < tau) {  // ascending index: an equal distance with a larger index never displaces
            heap_replace_root<IT>(sd, si, lane, k, d, (IT)(j0 + j));
            tau = sd[lane];
        }
    }
}

template <int F, typename IT>
__global__ void __launch_bounds__(64) analog_bf2_predict_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                int64_t Tq, int64_t T, int64_t C, int nbatch,
                                                                const double* __restrict__ Xc, const double* __restrict__ yc,
                                                                const int32_t* __restrict__ fit_status, int32_t* status,
                                                                PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int k = pa.k, lane = threadIdx.x;
    double* sd = reinterpret_cast<double*>(smem_raw);  // [k][64]
    IT* si = reinterpret_cast<IT*>(sd + (size_t)k * 64);  // [k][64]; 16-bit indices when T <= 65535 (more waves per CU)
    const int64_t c = blockIdx.x / nbatch;
    const int64_t tq = (int64_t)(blockIdx.x % nbatch) * 64 + lane;
    const bool active = fit_status[c] == 0, has_q = tq < Tq;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double q[F];
    bool ok = active && has_q;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        q[f] = has_q ? Xq[(tq * F + f) * ld + c] : 0.0;
        if (active && has_q && !sd_finite(q[f])) {
            atomicOr(&status[c], SDI_NONFINITE);
            ok = false;
        }
    }
    for (int i = 0; i < k; ++i) {
        sd[i * 64 + lane] = inf;
        si[i * 64 + lane] = (IT)~(IT)0 >> 1;  // larger than any training index
    }
    const double* __restrict__ Xcell = Xc + c * F * T;  // [F][T]
    double tau = ok ? inf : -1.0;  // k-th best distance so far; a lane without a query never flags a point
    double* stage = reinterpret_cast<double*>(si + (size_t)k * 64);  // [F][64] chunk coordinates for the insertion loop
    double cur[F][kScanG<F>];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int g = 0; g < kScanG<F>; ++g) cur[f][g] = T >= 64 ? Xcell[(int64_t)f * T + g] : 0.0;
    for (int64_t j0 = 0; j0 < T; j0 += 64) {
        const int nj = (int)(T - j0 < 64 ? T - j0 : 64);
        const int next_j0 = j0 + 128 <= T ? (int)j0 + 64 : 0;  // one uniform form: the last full chunk re-reads the first group
        scan_chunk<F, IT, false>(Xcell, T, nullptr, (int)j0, nj, next_j0, cur, q, tau, sd, si, k, lane, stage, nullptr);
    }
    // heap -> ascending (rdist, index): move the root behind the shrinking heap, k - 1 times
    for (int n = k - 1; n > 0; --n) {
        const double dl = sd[n * 64 + lane];
        const IT il = si[n * 64 + lane];
        sd[n * 64 + lane] = sd[lane];
        si[n * 64 + lane] = si[lane];
        heap_replace_root<IT>(sd, si, lane, n, dl, il);
    }
    if (has_q) finish_query(mode, pa, F, T, c, tq, q, Xcell, yc + c * T, sd, si, 64, ok);
}

// ------------------------------------------------------------------------------------------------
// general F predict, third form: the bf2 scan restricted to a slab of feature 0.
// fit keeps a copy of the training points sorted by feature 0 (ps, original indices in pi); predict sorts the
// queries of a cell by feature 0 as well, so the 64 queries of a wave are neighbours on that axis.  The wave scans
// the sorted training set outwards from its queries in 64-point chunks, alternating right and left, and a side is
// finished when its next point is farther from every query of the wave *along feature 0 alone* than the largest
// current k-th distance:  rdist >= fl((x0 - q0)^2) >= fl((x0 - qmax0)^2)  (floating-point subtraction, squaring and
// the accumulation of non-negative terms are monotone), so a skipped point is strictly worse than every kept one and
// the (rdist, index) selection stays exact.  Points now arrive out of index order: equal distances are decided by
// the explicit index comparison against the heap root.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}

// ps[c][f][j] = X[c][f][pi[c][j]]: the training points of a cell in feature-0 order
__global__ void __launch_bounds__(256) analog_gather_sorted_kernel(const double* __restrict__ Xc, const int32_t* __restrict__ pi,
                                                                   int64_t T, int F, int64_t C, double* __restrict__ ps) {
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x)
        for (int64_t j = threadIdx.x; j < T; j += blockDim.x) {
            const int idx = pi[c * T + j];
            for (int f = 0; f < F; ++f) ps[(c * F + f) * T + j] = Xc[(c * F + f) * T + idx];
        }
}

// scan state of a wave: [R, T) and [0, L) are still to be scanned, sides alternate
struct SlabCursor {
    int R, L, side;
    bool rdone, ldone;
};
struct SlabChunk {
    int j0, nj;  // nj = 0: both sides are finished
};
// next chunk (all values wave-uniform).  A side is finished when its next point is farther along feature 0 from every
// query of the wave (qlo..qhi) than the largest k-th distance taumax.
__device__ __forceinline__ SlabChunk slab_pick(const double* __restrict__ P0, int T, double qlo, double qhi, double taumax,
                                               SlabCursor& cs) {
    if (!cs.rdone) {
        const double g = P0[cs.R] - qhi;
        cs.rdone = g > 0.0 && g * g > taumax;
    }
    if (!cs.ldone) {
        const double g = qlo - P0[cs.L - 1];
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
    } else {
        ch.nj = cs.L < 64 ? cs.L : 64;
        ch.j0 = cs.L - ch.nj;
        cs.L = ch.j0;
        cs.ldone = cs.L <= 0;
    }
    ch.j0 = __builtin_amdgcn_readfirstlane(ch.j0);
    ch.nj = __builtin_amdgcn_readfirstlane(ch.nj);
    return ch;
}

template <int F>
__global__ void __launch_bounds__(64) analog_slab_predict_kernel(int mode, const double* __restrict__ qc /* [cc][F][Tq] */,
                                                                 const int32_t* __restrict__ qi /* [cc][Tq] */, int64_t c_base,
                                                                 int64_t Tq, int64_t T, int nbatch,
                                                                 const double* __restrict__ Xc, const double* __restrict__ yc,
                                                                 const double* __restrict__ ps, const int32_t* __restrict__ pi,
                                                                 const int32_t* __restrict__ fit_status, int32_t* status,
                                                                 PredictArgs pa, int ablate, unsigned long long* dbg,
                                                                 const int32_t* __restrict__ worklist) {
    typedef uint16_t IT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int k = pa.k, lane = threadIdx.x;
    double* sd = reinterpret_cast<double*>(smem_raw);   // [k][64]
    IT* si = reinterpret_cast<IT*>(sd + (size_t)k * 64);  // [k][64]
    // worklist: the (cell, query batch) pairs analog_slab_topk_kernel handed back (sd_analog_topk.h), one per workgroup
    const unsigned bid = worklist != nullptr ? (unsigned)worklist[blockIdx.x] : blockIdx.x;
    const int64_t cl = bid / nbatch, c = c_base + cl;
    const int64_t slot = (int64_t)(bid % nbatch) * 64 + lane;
    const bool active = fit_status[c] == 0, has_q = slot < Tq;
    const int64_t tq = has_q ? qi[cl * Tq + slot] : 0;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double q[F];
    bool ok = active && has_q;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        q[f] = has_q ? qc[(cl * F + f) * Tq + tq] : 0.0;
        if (active && has_q && !sd_finite(q[f])) {
            atomicOr(&status[c], SDI_NONFINITE);
            ok = false;
        }
    }
    for (int i = 0; i < k; ++i) {
        sd[i * 64 + lane] = inf;
        si[i * 64 + lane] = (IT)0xffffu;  // (never compared: a real distance is finite)
    }
    const double* __restrict__ P = ps + c * F * T;  // [F][T], ascending in feature 0
    const int32_t* __restrict__ PI = pi + c * T;
    double tau = ok ? inf : -1.0;  // k-th best distance so far; a lane without a query never flags a point
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
        const int R = p8, L = p8;  // [R, T) and [0, L) are still to be scanned
        const bool rdone = R >= (int)T, ldone = L <= 0;
        double* stage = reinterpret_cast<double*>(si + (size_t)k * 64);   // [F][64]
        int32_t* stage_i = reinterpret_cast<int32_t*>(stage + F * 64);     // [64]
        double cur[F][kScanG<F>];
        SlabCursor cs{R, L, 0, rdone, ldone};
        int cur_j = -1;
        SlabChunk ch = slab_pick(P, (int)T, qlo, qhi, uniform_f64(wave_max_f64(tau)), cs);
        while (ch.nj > 0) {
            // the chunk after this one is chosen with the k-th distances known now (one chunk stale: it can only scan
            // more than necessary), so that its first group is requested while this chunk is still being scanned
            const SlabChunk nx = slab_pick(P, (int)T, qlo, qhi, uniform_f64(wave_max_f64(tau)), cs);
            const int j0 = ch.j0, nj = ch.nj, j0n = nx.j0, njn = nx.nj;
            if (nj == 64 && cur_j != j0) {
#pragma unroll
                for (int f = 0; f < F; ++f)
#pragma unroll
                    for (int g = 0; g < kScanG<F>; ++g) cur[f][g] = P[(int64_t)f * T + j0 + g];
            }
            const int next_j0 = njn == 64 ? j0n : j0;
            scan_chunk<F, IT, true>(P, T, PI, j0, nj, next_j0, cur, q, tau, sd, si, k, lane, stage, stage_i, ablate, dbg);
            if (dbg && lane == 0) atomicAdd(&dbg[0], 1ull);
            if (nj == 64) cur_j = next_j0;
            ch = nx;
        }
    }
    if (ablate & 2) return;  // (timing experiments: no epilogue)
    // heap -> ascending (rdist, index): move the root behind the shrinking heap, k - 1 times
    for (int n = k - 1; n > 0; --n) {
        const double dl = sd[n * 64 + lane];
        const IT il = si[n * 64 + lane];
        sd[n * 64 + lane] = sd[lane];
        si[n * 64 + lane] = si[lane];
        heap_replace_root<IT>(sd, si, lane, n, dl, il);
    }
    if (has_q) finish_query(mode, pa, F, T, c, tq, q, Xc + c * F * T, yc + c * T, sd, si, 64, ok);
}

// heap [k][64] of (double, IT) + the chunk staging area [F][64] doubles + [64] indices (16-byte aligned pieces)
size_t bf2_lds_bytes(int k, int F, size_t it_bytes) {
    const size_t heap = ((size_t)k * 64 * (sizeof(double) + it_bytes) + 15) / 16 * 16;
    return heap + (size_t)F * 64 * sizeof(double) + 64 * sizeof(int32_t);
}

template <int F, typename IT>
int launch_bf2i(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int32_t* status_p,
                const PredictArgs& pa) {
    const size_t lds = bf2_lds_bytes(pa.k, F, sizeof(IT));
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_bf2_predict_kernel<F, IT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nbatch = (Tq + 63) / 64, nblocks = st->C * nbatch;
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "sd_analog_predict: too many (cell, query batch) pairs for one launch");
    SD_LAUNCH(ctx, "analog_bf2_predict_kernel", (analog_bf2_predict_kernel<F, IT>), dim3((unsigned)nblocks), dim3(64), lds, mode,
              Xq, ld, Tq, st->T, st->C, (int)nbatch, (const double*)st->X, (const double*)st->y, (const int32_t*)st->status,
              status_p, pa);
    return SD_OK;
}

template <int F>
int launch_bf2(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int32_t* status_p,
               const PredictArgs& pa) {
    if (st->T <= 65535) return launch_bf2i<F, uint16_t>(ctx, mode, st, Xq, ld, Tq, status_p, pa);
    return launch_bf2i<F, int32_t>(ctx, mode, st, Xq, ld, Tq, status_p, pa);
}

// Query order for the slab search.  A wave stops scanning when the axis distance exceeds the *largest* k-th distance of
// its 64 queries, so one query in a sparse region of the other features widens the slab for all of them.  The queries of
// a cell are therefore first classed by s2 = sum_{f>=1} ((q_f - mean_f) / std_f)^2 (moments of the query series itself)
// into nb classes holding 1/2, 1/4, 1/8, ... of the queries (thresholds = order statistics of s2, from a sort), then
// sorted by feature 0 inside a class: key = 4 * class + q0 / (1 + |q0|).  The order only groups the work; any order is
// exact.  (F=3, T=Tq=14 600, k=30, normal data: 64 % of the training points scanned per wave without classes, 29 % with 8.)
__global__ void __launch_bounds__(256) analog_slab_s2_kernel(const double* __restrict__ qc /* [C][F][Tq] */, int64_t Tq, int F,
                                                             int64_t C, double* __restrict__ s2 /* [C][Tq] */) {
    __shared__ double red[2][4];
    __shared__ double mom[2][kMaxF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double* q = qc + c * F * Tq;
        for (int f = 1; f < F; ++f) {
            double m = 0.0;
            for (int pass = 0; pass < 2; ++pass) {  // mean, then the sum of squared deviations, over the finite entries
                double a = 0.0, cnt = 0.0;
                for (int64_t j = tid; j < Tq; j += blockDim.x) {
                    const double v = q[f * Tq + j];
                    if (sd_finite(v)) {
                        a += pass == 0 ? v : (v - m) * (v - m);
                        cnt += 1.0;
                    }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    a += __shfl_xor(a, o, 64);
                    cnt += __shfl_xor(cnt, o, 64);
                }
                __syncthreads();
                if (lane == 0) {
                    red[0][wave] = a;
                    red[1][wave] = cnt;
                }
                __syncthreads();
                const double ta = red[0][0] + red[0][1] + red[0][2] + red[0][3];
                const double tc = red[1][0] + red[1][1] + red[1][2] + red[1][3];
                if (pass == 0) {
                    m = tc > 0.0 ? ta / tc : 0.0;
                } else if (tid == 0) {
                    mom[0][f] = m;
                    mom[1][f] = ta > 0.0 ? tc / ta : 0.0;  // 1 / variance
                }
            }
        }
        __syncthreads();
        for (int64_t j = tid; j < Tq; j += blockDim.x) {
            double s = 0.0;
            bool fin = sd_finite(q[j]);
            for (int f = 1; f < F; ++f) {
                const double v = q[f * Tq + j];
                fin = fin && sd_finite(v);
                s += (v - mom[0][f]) * (v - mom[0][f]) * mom[1][f];
            }
            s2[c * Tq + j] = fin && s < 1e300 ? s : 1e300;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) analog_slab_key_kernel(const double* __restrict__ qc, const double* __restrict__ s2_sorted,
                                                              int64_t Tq, int F, int64_t C, int nb,
                                                              double* __restrict__ key /* in: s2, out: sort key */) {
    __shared__ double th[8];
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < nb - 1) {
            const int64_t pos = Tq - (Tq >> (threadIdx.x + 1));  // 1/2, 3/4, 7/8, ... of the queries lie below
            th[threadIdx.x] = s2_sorted[c * Tq + (pos < Tq ? pos : Tq - 1)];
        }
        __syncthreads();
        for (int64_t j = threadIdx.x; j < Tq; j += blockDim.x) {
            const double s = key[c * Tq + j];
            int b = 0;
            for (int i = 0; i < nb - 1; ++i) b += s >= th[i] ? 1 : 0;
            const double q0 = qc[c * F * Tq + j];
            const double t = sd_finite(q0) ? q0 / (1.0 + __builtin_fabs(q0)) : 0.0;  // monotone map into (-1, 1)
            key[c * Tq + j] = 4.0 * (double)b + t;
        }
    }
}

// worklist (device, nwork entries): only the listed (cell, query batch) pairs (the hand-backs of analog_slab_topk_kernel)
template <int F>
int launch_slab(sd_ctx* ctx, int mode, const sd_analog_state* st, const double* qc, const int32_t* qi, int64_t cb, int64_t cc,
                int64_t Tq, int32_t* status_p, const PredictArgs& pa, const int32_t* worklist = nullptr, int64_t nwork = 0) {
    const size_t lds = bf2_lds_bytes(pa.k, F, sizeof(uint16_t));
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_slab_predict_kernel<F>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nbatch = (Tq + 63) / 64, nblocks = worklist != nullptr ? nwork : cc * nbatch;
    const char* eab = sd_dev_env("SD_ANALOG_ABLATE");  // timing experiments only (results are wrong): 1 no insertions, 2 no epilogue
    const int ablate = eab ? atoi(eab) : 0;
    sd_scratch dbg;  // 4: count scanned chunks and insertion rounds, printed per launch
    if (ablate & 4) {
        SD_HIP(dbg.alloc(ctx, 16));
        SD_HIP(hipMemsetAsync(dbg.p, 0, 16, ctx->stream));
    }
    SD_LAUNCH(ctx, "analog_slab_predict_kernel", (analog_slab_predict_kernel<F>), dim3((unsigned)nblocks), dim3(64), lds, mode, qc,
              qi, cb, Tq, st->T, (int)nbatch, (const double*)st->X, (const double*)st->y, (const double*)st->ps,
              (const int32_t*)st->xi, (const int32_t*)st->status, status_p, pa, ablate, dbg.as<unsigned long long>(), worklist);
    if (ablate & 4) {
        unsigned long long h[2];
        SD_HIP(hipMemcpyAsync(h, dbg.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[slab] waves %lld: chunks/wave %.1f insertion rounds/wave %.1f\n", (long long)nblocks,
                (double)h[0] / (double)nblocks, (double)h[1] / (double)nblocks);
    }
    return SD_OK;
}
