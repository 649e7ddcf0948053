// sd_analog_f1.h -- part of the translation unit csrc/sd_analog.hip (included there, inside its unnamed namespace; not a
// stand-alone header).  F == 1 predict kernels (exact walk, window form, prefix-sum forms) and the fused fit + predict kernel.

// ------------------------------------------------------------------------------------------------
// F == 1 predict: sorted training values in LDS, binary search + two-pointer walk
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) analog_f1_predict_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                 int64_t Tq, int64_t T, int64_t C,
                                                                 const double* __restrict__ xs_all,
                                                                 const int32_t* __restrict__ xi_all,
                                                                 const double* __restrict__ Xc,
                                                                 const double* __restrict__ yc,
                                                                 const int32_t* __restrict__ fit_status, int32_t* status,
                                                                 double* scratch_d, int32_t* scratch_i, PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int n = (int)T, k = pa.k;
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    int64_t step, end;
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const int32_t* xi = xi_all + c * T;
        __syncthreads();
        if (active)
            for (int i = tid; i < n; i += nthr) xs[i] = xs_all[c * T + i];
        __syncthreads();
        for (int64_t tq = tid; tq < Tq; tq += nthr) {
            const double q = Xq[tq * ld + c];
            bool ok = active;
            if (active && !sd_finite(q)) {
                atomicOr(&status[c], SDI_NONFINITE);
                ok = false;
            }
            if (ok) {
                // r = first sorted position with x > q ; left part ends at r - 1
                int lo = 0, hi = n;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (xs[mid] <= q) lo = mid + 1; else hi = mid;
                }
                int r = lo;          // next right candidate
                int le = lo - 1;     // last element of the current left run (-1: exhausted)
                int rs = 0, cur = 0; // current left run [rs, le], next to take = cur (ascending index order)
                if (le >= 0) {
                    rs = le;
                    while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
                    cur = rs;
                }
                for (int i = 0; i < k; ++i) {
                    double dl = 0.0, dr = 0.0;
                    const bool hl = le >= 0, hr = r < n;
                    if (hl) { const double d = q - xs[le]; dl = d * d; }
                    if (hr) { const double d = q - xs[r]; dr = d * d; }
                    bool take_left;
                    if (hl && hr) take_left = dl < dr || (dl == dr && xi[cur] < xi[r]);
                    else take_left = hl;
                    if (take_left) {
                        sd[(int64_t)i * nthr + tid] = dl;
                        si[(int64_t)i * nthr + tid] = xi[cur];
                        if (++cur > le) {
                            le = rs - 1;
                            if (le >= 0) {
                                rs = le;
                                while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
                                cur = rs;
                            }
                        }
                    } else {
                        sd[(int64_t)i * nthr + tid] = dr;
                        si[(int64_t)i * nthr + tid] = xi[r];
                        ++r;
                    }
                }
            }
            finish_query(mode, pa, 1, T, c, tq, &q, Xc + c * T, yc + c * T, sd, si, nthr, ok);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// F == 1 predict, window form.  In one dimension the k nearest training values are k consecutive
// entries of the sorted view unless a tie sits on the boundary, so a query costs one binary search
// for the window start (2 LDS reads per step) and one pass over yx[L .. L+k) -- the analog values
// in sorted-x order, k consecutive doubles.  No neighbour lists, no gathers.  Used for PureAnalog
// kinds best / weight / mean when neither indices nor distances are requested; a query whose window
// is not strictly separated from its outside neighbours (exact distance ties, tie runs cut by the
// left boundary: KDTree order then depends on the training index) is answered by the exact
// (rdist, index)-ordered walk below, as are 'sample_analogs' and AnalogRegression.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sq_dist(double q, double x) {
    const double d = q - x;
    return d * d;
}

__device__ void f1_walk_query(int mode, const PredictArgs& pa, int n, int64_t T, int64_t c, int64_t tq, double q,
                              const double* xs /* LDS */, const int32_t* __restrict__ xi, const double* __restrict__ Xc_cell,
                              const double* __restrict__ yc_cell, double* sd, int32_t* si, int nthr) {
    const int tid = threadIdx.x, k = pa.k;
    // r = first sorted position with x > q ; left part ends at r - 1
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] <= q) lo = mid + 1; else hi = mid;
    }
    int r = lo;          // next right candidate
    int le = lo - 1;     // last element of the current left run (-1: exhausted)
    int rs = 0, cur = 0; // current left run [rs, le], next to take = cur (ascending index order)
    if (le >= 0) {
        rs = le;
        while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
        cur = rs;
    }
    for (int i = 0; i < k; ++i) {
        double dl = 0.0, dr = 0.0;
        const bool hl = le >= 0, hr = r < n;
        if (hl) dl = sq_dist(q, xs[le]);
        if (hr) dr = sq_dist(q, xs[r]);
        bool take_left;
        if (hl && hr) take_left = dl < dr || (dl == dr && xi[cur] < xi[r]);
        else take_left = hl;
        if (take_left) {
            sd[(int64_t)i * nthr + tid] = dl;
            si[(int64_t)i * nthr + tid] = xi[cur];
            if (++cur > le) {
                le = rs - 1;
                if (le >= 0) {
                    rs = le;
                    while (rs > 0 && xs[rs - 1] == xs[le]) --rs;
                    cur = rs;
                }
            }
        } else {
            sd[(int64_t)i * nthr + tid] = dr;
            si[(int64_t)i * nthr + tid] = xi[r];
            ++r;
        }
    }
    finish_query(mode, pa, 1, T, c, tq, &q, Xc_cell, yc_cell, sd, si, nthr, true);
}

typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));  // a pair of doubles at any 8-byte boundary
constexpr int kWinQ = 2;      // queries a thread answers together (independent dependency chains)
constexpr int kWinBatch = 8;  // analog values read together per query

// The sorted view of a cell is processed in `npass` value ranges so that both xs and yx of a range (plus k
// entries of margin on either side) sit in LDS: the window search and the k analog values of a query are LDS
// reads, HBM/L2 only see the query and the three outputs.  A query belongs to the range that holds its value;
// its k nearest neighbours are at most k positions away from there.
__global__ void __launch_bounds__(1024) analog_f1_window_kernel(int mode, const double* __restrict__ Xq, int64_t ld,
                                                                int64_t Tq, int64_t T, int64_t C, int npass,
                                                                const double* __restrict__ xs_all,
                                                                const int32_t* __restrict__ xi_all,
                                                                const double* __restrict__ yx_all,
                                                                const double* __restrict__ Xc, const double* __restrict__ yc,
                                                                const int32_t* __restrict__ fit_status, int32_t* status,
                                                                double* scratch_d, int32_t* scratch_i, PredictArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int n = (int)T, k = pa.k;
    const int seg = (n + npass - 1) / npass;
    const int cap = seg + 2 * k + 1;                        // local entries per pass (upper bound)
    double* xs = reinterpret_cast<double*>(smem_raw);       // cap + 1 doubles (sentinel)
    double* yl = xs + cap + 1;                              // cap doubles
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    int64_t step, end;
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const int32_t* xi = xi_all + c * T;
        const double* xg = xs_all + c * T;
        const double* yx = yx_all + c * T;
        for (int p = 0; p < npass; ++p) {
            const int b0 = p * seg < n ? p * seg : n, b1 = (p + 1) * seg < n ? (p + 1) * seg : n;
            if (b0 >= b1) break;
            const int g0 = b0 - k > 0 ? b0 - k : 0, g1 = b1 + k < n ? b1 + k : n;  // global range held in LDS
            const int nl = g1 - g0;
            // value range of this pass: [vlo, vhi), open-ended at the ends of the sorted view
            const double vlo = (p == 0 || !active) ? -inf : xg[b0];
            const double vhi = (b1 >= n || !active) ? inf : xg[b1];
            __syncthreads();
            if (active)
                for (int i = tid; i < nl; i += nthr) {
                    xs[i] = xg[g0 + i];
                    yl[i] = yx[g0 + i];
                }
            if (tid == 0) xs[nl] = inf;
            __syncthreads();
            int nsteps = 0;  // fixed trip count of the window search: every lane and query runs the same loop
            while ((1 << nsteps) < nl - k + 1) ++nsteps;
            for (int64_t tq0 = tid; tq0 < Tq; tq0 += (int64_t)nthr * kWinQ) {
                double q[kWinQ];
                bool has[kWinQ], ok[kWinQ], mine[kWinQ];
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    const int64_t tq = tq0 + (int64_t)j * nthr;
                    has[j] = tq < Tq;
                    q[j] = has[j] ? Xq[c * ld + tq] : 0.0;  // cell-major copy of the queries (ld = Tq)
                }
                bool any = false;
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    ok[j] = active && has[j] && sd_finite(q[j]);
                    // inactive cells and non-finite queries are reported (NaN outputs) in the first pass
                    mine[j] = has[j] && (ok[j] ? (q[j] >= vlo && q[j] < vhi) || (q[j] == inf) : p == 0);
                    if (mine[j] && active && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                    if (!ok[j]) q[j] = 0.0;
                    any |= mine[j];
                }
                if (!any) continue;
                // window start: smallest L with rdist(L) <= rdist(L + k) (rdist is unimodal along the sorted view)
                int lo[kWinQ], hi[kWinQ];
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    lo[j] = 0;
                    hi[j] = nl - k;
                }
#pragma unroll 1
                for (int s = 0; s < nsteps; ++s) {
#pragma unroll
                    for (int j = 0; j < kWinQ; ++j) {
                        const int mid = (lo[j] + hi[j]) >> 1;
                        const bool act = lo[j] < hi[j];
                        const bool right = sq_dist(q[j], xs[mid]) > sq_dist(q[j], xs[mid + k]);
                        lo[j] = (act && right) ? mid + 1 : lo[j];
                        hi[j] = (act && !right) ? mid : hi[j];
                    }
                }
                bool unique[kWinQ];
                // sums over the window, shifted by its first element (x0, a0) so that no cancellation occurs;
                // PureAnalog: s1 = sum(a), s2 = sum(a^2), weights; AnalogRegression: wsum/awsum/sxx hold sum(x), sum(x*a), sum(x^2)
                double x0[kWinQ], a0[kWinQ], s1[kWinQ], s2[kWinQ], wsum[kWinQ], awsum[kWinQ], sxx[kWinQ];
                int nexc[kWinQ];
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    const int L = lo[j];
                    const double dL = sq_dist(q[j], xs[L]), dR = sq_dist(q[j], xs[L + k - 1]);
                    const double worst = dL > dR ? dL : dR;
                    // the outside neighbours must be strictly farther; at an edge of the LDS range that is not an
                    // edge of the sorted view the neighbour is unknown -> exact walk
                    const bool sep_l = L == 0 ? g0 == 0 : sq_dist(q[j], xs[L - 1]) > worst;
                    const bool sep_r = L + k == nl ? g1 == n : sq_dist(q[j], xs[L + k]) > worst;
                    unique[j] = sep_l && sep_r;
                    s1[j] = s2[j] = wsum[j] = awsum[j] = sxx[j] = 0.0;
                    nexc[j] = 0;
                    x0[j] = xs[L];
                    a0[j] = yl[L];
                }
                const bool need_x = mode == 1 || pa.kind == SD_ANALOG_WEIGHT;
                for (int i0 = 0; i0 < k; i0 += kWinBatch) {
#pragma unroll
                    for (int j = 0; j < kWinQ; ++j)
#pragma unroll
                        for (int b = 0; b < kWinBatch; ++b) {
                            const int i = i0 + b;
                            if (i < k) {
                                const double ai = yl[lo[j] + i];
                                const double e = ai - a0[j];
                                s1[j] += e;
                                s2[j] += e * e;
                                nexc[j] += (!pa.has_thresh || ai > pa.thresh) ? 1 : 0;  // gard.py:307
                                if (need_x) {
                                    const double xv = xs[lo[j] + i];
                                    if (mode == 1) {
                                        const double dx = xv - x0[j];
                                        wsum[j] += dx;
                                        awsum[j] += dx * e;
                                        sxx[j] += dx * dx;
                                    } else {
                                        // w = 1 / distance (gard.py:322-323); sqrt((q-x)^2) == |q-x| in IEEE arithmetic.
                                        // Reciprocal by v_rcp_f64 + two Newton steps (< 1 ulp; the tolerance is 1e-6).
                                        double d = __builtin_fabs(q[j] - xv);
                                        d = d == 0.0 ? 1e-20 : d;
                                        double r = __builtin_amdgcn_rcp(d);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                                        wsum[j] += r;
                                        awsum[j] += ai * r;
                                    }
                                }
                            }
                        }
                }
#pragma unroll
                for (int j = 0; j < kWinQ; ++j) {
                    if (!mine[j]) continue;
                    const int64_t tq = tq0 + (int64_t)j * nthr;
                    double pred = nan, prob = nan, err = nan;
                    if (ok[j]) {
                        const int L = lo[j];
                        double best_a = a0[j];
                        if (mode == 0 && pa.kind == SD_ANALOG_BEST && unique[j]) {
                            // nearest element = one of the two around the insertion point of q inside the window;
                            // equal distances or equal values there leave the choice to the training index -> walk
                            int m = 0;  // first window entry with x >= q
                            for (int len = k; len > 0;) {
                                const int half = len >> 1;
                                if (xs[L + m + half] < q[j]) { m += half + 1; len -= half + 1; } else len = half;
                            }
                            const double dl = m > 0 ? sq_dist(q[j], xs[L + m - 1]) : inf;
                            const double dr = m < k ? sq_dist(q[j], xs[L + m]) : inf;
                            if (dl == dr) unique[j] = false;
                            const int b = dl < dr ? m - 1 : m;
                            if (dl < dr ? (b > 0 && xs[L + b - 1] == xs[L + b]) : (b + 1 < k && xs[L + b + 1] == xs[L + b]))
                                unique[j] = false;
                            best_a = yl[L + (b < k ? b : k - 1)];
                        }
                        if (!unique[j]) {
                            f1_walk_query(mode, pa, n, T, c, tq, q[j], xg, xi, Xc + c * T, yc + c * T, sd, si, nthr);
                            continue;
                        }
                        const bool any_masked = nexc[j] != k;
                        const double kk = (double)k;
                        const double m1 = s1[j] / kk;
                        if (mode == 1) {
                            // one-feature OLS on the k analogs (gard.py:194-224): centred sums, slope 0 when all x are equal
                            const double mx = wsum[j] / kk;
                            const double vxx = sxx[j] - kk * mx * mx, vxy = awsum[j] - kk * mx * m1;
                            const double slope = vxx > 0.0 ? vxy / vxx : 0.0;
                            const double xm = x0[j] + mx, ym = a0[j] + m1;
                            const double icpt = ym - xm * slope;
                            pred = icpt + q[j] * slope;
                            double ss = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double r = yl[L + i] - (icpt + xs[L + i] * slope);
                                ss += r * r;
                            }
                            prob = 1.0;
                            err = sqrt(ss / kk);  // root_mean_squared_error (gard.py:218-219)
                        } else {
                            if (pa.kind == SD_ANALOG_BEST) pred = best_a;                                         // gard.py:311
                            else if (pa.kind == SD_ANALOG_WEIGHT) pred = any_masked ? nan : awsum[j] / wsum[j];   // gard.py:319-327
                            else pred = any_masked ? nan : a0[j] + m1;                                            // gard.py:329-333
                            if (pa.has_thresh) {
                                pred = nan_to_num(pred);      // gard.py:341
                                prob = (double)nexc[j] / kk;   // gard.py:343
                            } else {
                                prob = 1.0;  // gard.py:346
                            }
                            if (!any_masked) {
                                const double var = s2[j] / kk - m1 * m1;
                                err = sqrt(var > 0.0 ? var : 0.0);  // ddof = 0 (gard.py:342,345)
                            }
                        }
                    }
                    put_out(pa, tq, c, pred, prob, err);
                }
            }
        }
    }
}

// Window start of two queries in ONE bisection each (round 6).  The window of the k nearest values of q starts at the first i
// whose left end is not farther from q than the value k places on: (q - xs[i])^2 <= (xs[i + k] - q)^2, i.e. xs[i] + xs[i + k] >= 2 q
// -- a predicate that is monotone in i (a rounded sum of two sorted sequences is sorted), false before the start and true from it
// on, with xs[n] = +inf closing the range at i = M = n - k.  Two LDS reads, an addition and a compare per step over the M + 1
// candidates replace the position search among the values (14 steps) AND the refinement among the k + 1 candidates around the
// position (5 steps of two squared distances each): a third fewer vector instructions in the loop that bounds these kernels.
// The rounding of the sum can misplace the start by one only when q sits within an ulp of a midpoint; the callers' test that
// the window is STRICTLY separated from its outside neighbours -- exact, on the values -- then fails as it does for a real tie,
// and the query takes the exact walk (the unique strictly separated window is the same whichever search found it).
template <int NQ>
__device__ __forceinline__ void window_starts_n(const double* buf, int k, int M, const double (&q)[NQ], int (&L)[NQ]) {
    int pos[NQ];  // index of the last candidate known to lie before the start
    double q2[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        pos[j] = -1;
        q2[j] = q[j] + q[j];
    }
#pragma unroll 1
    for (int len = M + 1; len > 1;) {
        int half = len >> 1;
        if ((half & 15) == 0) --half;  // (strides that are multiples of 16 doubles pile the probes on two banks)
        len -= half;
        // The reads of ALL the queries are issued before the first is consumed (sched_barrier): left alone, the compiler -- short
        // of registers in these kernels -- reuses one pair of destination registers and serialises the queries inside a step,
        // read, wait, compare, next query: one chain of LDS round trips in flight per thread, whatever NQ is.  That, not LDS
        // bandwidth, bank conflicts or instruction count, is what bounded the search.
        double a[NQ], b[NQ];
        int t[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            t[j] = pos[j] + half;
            a[j] = buf[t[j]];
            b[j] = buf[t[j] + k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NQ; ++j) pos[j] = a[j] + b[j] < q2[j] ? t[j] : pos[j];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int t = pos[j] + 1;  // <= M; the candidate M is a start for every q (xs[n] = +inf)
        L[j] = t + (t < M && buf[t] + buf[t + k] < q2[j] ? 1 : 0);
    }
}
// the search these kernels had until round 5: position of q among the values (one read per step), then the start of the window
// among the k + 1 candidates around it (two reads per step).  Fewer reads than the single bisection while k + 1 candidates are
// few: a single analog (best_analog, the reference's default) refines in one step.
template <int NQ>
__device__ __forceinline__ void window_starts_two_level(const double* buf, int k, int n, int M, const double (&q)[NQ], int (&L)[NQ]) {
    int pos[NQ], hi[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) pos[j] = -1;  // index of the last value known to be < q
#pragma unroll 1
    for (int len = n; len > 1;) {
        int half = len >> 1;
        if ((half & 15) == 0) --half;
        len -= half;
        double a[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) a[j] = buf[pos[j] + half];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NQ; ++j) pos[j] += a[j] < q[j] ? half : 0;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int p = pos[j] + 1 + (buf[pos[j] + 1] < q[j] ? 1 : 0);
        L[j] = p - k > 0 ? p - k : 0;
        hi[j] = p < M ? p : M;
    }
    int nsteps = 0;
    while ((1 << nsteps) < (k + 1 < M + 1 ? k + 1 : M + 1)) ++nsteps;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        double a[NQ], b[NQ];
        int mid[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {  // (all reads first: see window_starts_n)
            mid[j] = (L[j] + hi[j]) >> 1;
            a[j] = buf[mid[j]];
            b[j] = buf[mid[j] + k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const bool act = L[j] < hi[j];
            const bool right = sq_dist(q[j], a[j]) > sq_dist(q[j], b[j]);
            L[j] = (act && right) ? mid[j] + 1 : L[j];
            hi[j] = (act && !right) ? mid[j] : hi[j];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
constexpr int kSingleBisectionFromK = 8;  // (k + 1 candidates: the refinement of the two-level search costs 2 log2(k + 1) reads)
template <int NQ>
__device__ __forceinline__ void window_starts_any(const double* buf, int k, int n, int M, const double (&q)[NQ], int (&L)[NQ]) {
    if (k >= kSingleBisectionFromK) window_starts_n<NQ>(buf, k, M, q, L);  // (k is uniform over the launch)
    else window_starts_two_level<NQ>(buf, k, n, M, q, L);
}

// B analogs of a 'weight_analogs' window (no threshold) as straight-line code: the distances, zero guards and refined reciprocals
// of the batch are independent of one another and are requested / computed side by side, only the four running sums are chains.
// (The generic loop of analog_f1_mean_kernel takes an analog at a time behind a scalar branch -- LDS read, wait, 14 dependent
// float64 instructions --: with four waves per SIMD the vector ALUs sat at 67 %, profiles/r06/weight_mean_kernel_sq_counters.log.)
// Same operations on the same operands in the same order of accumulation as that loop: bit-identical.
template <int B>
__device__ __forceinline__ void weight_batch(const double* __restrict__ yl, const double* xw, double qj, double a0, double& s1, double& s2,
                                             double& wsum, double& awsum) {
    static_assert(B % 2 == 0, "pairs of analog values per load");
    double ab[B], xv[B], rw[B];
#pragma unroll
    for (int b = 0; b < B; b += 2) {
        const f64x2_a8 v = *reinterpret_cast<const f64x2_a8*>(yl + b);
        ab[b] = v.x;
        ab[b + 1] = v.y;
    }
#pragma unroll
    for (int b = 0; b < B; ++b) xv[b] = xw[b];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        // w = 1 / distance (gard.py:322-323): v_rcp_f64 + two Newton steps (< 1 ulp)
        double d = __builtin_fabs(qj - xv[b]);
        d = d == 0.0 ? 1e-20 : d;
        double r = __builtin_amdgcn_rcp(d);
        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
        rw[b] = r;
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const double e = ab[b] - a0;
        s1 += e;
        s2 += e * e;
        wsum += rw[b];
        awsum += ab[b] * rw[b];
    }
}
// the same with a threshold (gard.py:307: the analogs above it are counted; the sums run over all of them, a masked window answers NaN
// further down), with or without the weights
template <int B, bool WEIGHT>
__device__ __forceinline__ void thresh_batch(const double* __restrict__ yl, const double* xw, double qj, double a0, double thresh, double& s1,
                                             double& s2, double& wsum, double& awsum, int& nexc) {
    static_assert(B % 2 == 0, "pairs of analog values per load");
    double ab[B], xv[B], rw[B];
#pragma unroll
    for (int b = 0; b < B; b += 2) {
        const f64x2_a8 v = *reinterpret_cast<const f64x2_a8*>(yl + b);
        ab[b] = v.x;
        ab[b + 1] = v.y;
    }
    if constexpr (WEIGHT) {
#pragma unroll
        for (int b = 0; b < B; ++b) xv[b] = xw[b];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            double d = __builtin_fabs(qj - xv[b]);
            d = d == 0.0 ? 1e-20 : d;
            double r = __builtin_amdgcn_rcp(d);
            r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
            r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
            rw[b] = r;
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const double e = ab[b] - a0;
        s1 += e;
        s2 += e * e;
        nexc += ab[b] > thresh ? 1 : 0;
        if constexpr (WEIGHT) {
            wsum += rw[b];
            awsum += ab[b] * rw[b];
        }
    }
}
// B analogs of a regression window, everything the one-feature OLS needs in one pass: sums of dx = x - x0, dx^2, e = y - a0, e^2, dx e
// (centred on the window's own first pair: no cancellation against the cell's spread), loads of the batch issued together
template <int B>
__device__ __forceinline__ void reg_batch(const double* __restrict__ yl, const double* xw, double x0, double a0, double& sx, double& sxx,
                                          double& t1, double& t2, double& txy) {
    static_assert(B % 2 == 0, "pairs of analog values per load");
    double ab[B], xv[B];
#pragma unroll
    for (int b = 0; b < B; b += 2) {
        const f64x2_a8 v = *reinterpret_cast<const f64x2_a8*>(yl + b);
        ab[b] = v.x;
        ab[b + 1] = v.y;
    }
#pragma unroll
    for (int b = 0; b < B; ++b) xv[b] = xw[b];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const double dx = xv[b] - x0, e = ab[b] - a0;
        sx += dx;
        sxx += dx * dx;
        t1 += e;
        t2 += e * e;
        txy += dx * e;
    }
}
// B training values of a regression window: sums of x - x0 and of its square, the LDS reads of the batch issued together
template <int B>
__device__ __forceinline__ void xsum_batch(const double* xw, double x0, double& sx, double& sxx) {
    double xv[B];
#pragma unroll
    for (int b = 0; b < B; ++b) xv[b] = xw[b];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const double dx = xv[b] - x0;
        sx += dx;
        sxx += dx * dx;
    }
}

#ifndef SD_MEANQ
#define SD_MEANQ 2
#endif
constexpr int kRegDirectK = 64;  // AnalogRegression windows up to this long are summed directly (reg_batch); longer ones take the prefix sums
constexpr int kMeanQ = SD_MEANQ;  // queries a thread of analog_f1_mean_kernel searches together (independent bisection chains)
// F == 1, single pass over the queries with only the sorted training values LDS-resident.  'mean_analogs' without a
// threshold, a single analog and AnalogRegression (mode 1, k >= 3) take the window statistics from the prefix sums
// pq / rx (analog_prefix_kernel): the window search plus two (regression: three) pairs of prefix loads per query.
// 'weight_analogs' and the thresholded kinds read the k consecutive analog values of the window from memory.
// Tie handling as in analog_f1_window_kernel.
__global__ void __launch_bounds__(1024) analog_f1_mean_kernel(int mode, const double* __restrict__ Xq /* [C][Tq] */, int64_t Tq,
                                                              int64_t T, int64_t C, const double* __restrict__ xs_all,
                                                              const int32_t* __restrict__ xi_all,
                                                              const double* __restrict__ pq_all,
                                                              const double* __restrict__ ybar_all,
                                                              const double* __restrict__ rx_all,
                                                              const double* __restrict__ xbar_all,
                                                              const double* __restrict__ yx_all, const double* __restrict__ Xc,
                                                              const double* __restrict__ yc,
                                                              const int32_t* __restrict__ fit_status, int32_t* status,
                                                              double* scratch_d, int32_t* scratch_i, PredictArgs pa, int qsplit,
                                                              int reg_direct /* mode 1: window sums by direct summation (pq / rx unused) */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);  // n sorted values + one +inf sentinel
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int n = (int)T, k = pa.k;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    int nsteps = 0;  // window refinement: the range p - k .. p holds at most k + 1 candidates
    while ((1 << nsteps) < (k + 1 < n - k + 1 ? k + 1 : n - k + 1)) ++nsteps;
    const double kk = (double)k;
    // qsplit workgroups of one XCD share a cell (each answers 1/qsplit of its queries), so that the XCD works on
    // fewer cells at a time and the prefix sums of those cells (read at two random places per query) stay in its L2.
    // Pays for the regression (three prefix arrays: pq, rx), not for the plain mean (the extra LDS fills cost more)
    int64_t step, end, c0;
    int part = 0;
    if (qsplit > 1) {  // (the launcher guarantees gridDim.x % (8 * qsplit) == 0)
        const int64_t cx = (C + 7) / 8;
        const int x = blockIdx.x % 8, j = blockIdx.x / 8;
        part = j % qsplit;
        step = gridDim.x / 8 / qsplit;
        end = (x + 1) * cx < C ? (x + 1) * cx : C;
        c0 = x * cx + j / qsplit;
    } else {
        c0 = first_cell(C, &step, &end);
    }
    const int64_t qchunk = (Tq + qsplit - 1) / qsplit, q_beg = part * qchunk, q_end = q_beg + qchunk < Tq ? q_beg + qchunk : Tq;
    for (int64_t c = c0; c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const double* xg = xs_all + c * T;
        const double2* pq = reinterpret_cast<const double2*>(pq_all) + c * (T + 1);
        const double ybar = ybar_all[c];
        const double* rx = rx_all + c * (T + 1);
        const double xbar = mode == 1 && !reg_direct ? xbar_all[c] : 0.0;
        // AnalogRegression: residual sums below this are left to direct summation (the prefix differences carry an
        // absolute error of ~1e-16 of the cell total)
        const double ss_floor = mode == 1 && !reg_direct ? 1e-4 * kk * (pq[n].y / (double)n) : 0.0;
        __syncthreads();
        if (active)
            for (int i = tid; i < n; i += nthr) xs[i] = xg[i];
        if (tid == 0) xs[n] = inf;
        __syncthreads();
        for (int64_t tq0 = q_beg + tid; tq0 < q_end; tq0 += (int64_t)nthr * kMeanQ) {
            double q[kMeanQ];
            bool has[kMeanQ], ok[kMeanQ];
#pragma unroll
            for (int j = 0; j < kMeanQ; ++j) {
                const int64_t tq = tq0 + (int64_t)j * nthr;
                has[j] = tq < q_end;
                q[j] = has[j] ? Xq[c * Tq + tq] : 0.0;
                ok[j] = active && has[j] && sd_finite(q[j]);
                if (active && has[j] && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                if (!ok[j]) q[j] = 0.0;
            }
            // p = number of training values < q (one LDS read per step; strides that are multiples of 16 doubles are
            // shortened by one, see the rank search in sd_bcsd_rs.hip).  Without ties the k nearest values are a window
            // [L, L + k) with p - k <= L <= p: the smallest L of that range with rdist(L) <= rdist(L + k), log2(k + 1)
            // more steps of two reads (rdist is unimodal along xs).  With ties the separation test below sends the query
            // to the exact walk.
            int lo[kMeanQ];
#ifdef SD_MEAN_NOSEARCH
#pragma unroll
            for (int j = 0; j < kMeanQ; ++j) lo[j] = (int)(((unsigned)(tq0 + j) * 2654435761u) % (unsigned)(n - k));
#else
            window_starts_any<kMeanQ>(xs, k, n, n - k > 0 ? n - k : 0, q, lo);  // (round 6: one bisection on xs[i] + xs[i + k] >= 2 q, see window_starts)
#endif
            // the statistics of one query at a time, as a rolled loop (its body is long: unrolled over the queries of the
            // thread it no longer fits the instruction cache); the query in turn sits in slot 0, the others move down
            unsigned hasm = 0u, okm = 0u;
#pragma unroll
            for (int j = 0; j < kMeanQ; ++j) {
                hasm |= has[j] ? 1u << j : 0u;
                okm |= ok[j] ? 1u << j : 0u;
            }
#pragma unroll 1
            for (int jr = 0; jr < kMeanQ; ++jr) {
                const double qj = q[0];
                const int L = lo[0];
                const bool has_j = (hasm >> jr) & 1u, ok_j = (okm >> jr) & 1u;
#pragma unroll
                for (int i = 0; i + 1 < kMeanQ; ++i) {
                    q[i] = q[i + 1];
                    lo[i] = lo[i + 1];
                }
                if (!has_j) continue;
                const int64_t tq = tq0 + (int64_t)jr * nthr;
                double pred = nan, prob = nan, err = nan;
                if (ok_j) {
                    const double dL = sq_dist(qj, xs[L]), dR = sq_dist(qj, xs[L + k - 1]);
                    const double worst = dL > dR ? dL : dR;
                    const bool sep_l = L == 0 || sq_dist(qj, xs[L - 1]) > worst;
                    const bool sep_r = L + k == n || sq_dist(qj, xs[L + k]) > worst;
                    if (!(sep_l && sep_r)) {
                        f1_walk_query(mode, pa, n, T, c, tq, qj, xg, xi_all + c * T, Xc + c * T, yc + c * T, sd, si, nthr);
                        continue;
                    }
                    if (mode == 1 && reg_direct) {
                        // one-feature OLS on the k analogs (gard.py:194-224) from ONE pass over the window (short windows: k <= kRegDirectK):
                        // every sum centred on the window's first pair, so the residual sum ss = vyy - slope vxy carries a relative error
                        // of ~ eps vyy / ss -- far below the 1e-6 of the parity contract unless the fit is (nearly) exact, and then the
                        // residuals are summed directly in a second pass, as in the prefix form.  No prefix arrays: analog_prefix_kernel
                        // and analog_rx_kernel (21 of 110 ms per 100 000 cells at k = 30) and their 24 bytes per sample are not needed.
                        const double x0 = xs[L];
                        const double* yl = yx_all + c * T + L;
                        const double a0 = yl[0];
                        double sx = 0.0, sxx = 0.0, t1 = 0.0, t2 = 0.0, txy = 0.0;
                        {
                            int i = 0;
                            for (; i + 8 <= k; i += 8) reg_batch<8>(yl + i, xs + L + i, x0, a0, sx, sxx, t1, t2, txy);
                            if (i + 4 <= k) {
                                reg_batch<4>(yl + i, xs + L + i, x0, a0, sx, sxx, t1, t2, txy);
                                i += 4;
                            }
                            if (i + 2 <= k) {
                                reg_batch<2>(yl + i, xs + L + i, x0, a0, sx, sxx, t1, t2, txy);
                                i += 2;
                            }
                            if (i < k) {
                                const double dx = xs[L + i] - x0, e = yl[i] - a0;
                                sx += dx;
                                sxx += dx * dx;
                                t1 += e;
                                t2 += e * e;
                                txy += dx * e;
                            }
                        }
                        const double mx = sx / kk, xm = x0 + mx, n1 = t1 / kk;
                        const double vxx = sxx - kk * mx * mx, vyy = t2 - kk * n1 * n1, wxy = txy - kk * mx * n1;
                        const double slope = vxx > 0.0 ? wxy / vxx : 0.0;
                        double ss = vyy - slope * wxy;
                        pred = (a0 + n1) + (qj - xm) * slope;
                        if (!(ss > 1e-7 * vyy)) {
                            // (nearly) exact fit or constant analogs: the residuals themselves
                            const double icpt = (a0 + n1) - xm * slope;
                            pred = icpt + qj * slope;
                            ss = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double r = yl[i] - (icpt + xs[L + i] * slope);
                                ss += r * r;
                            }
                        }
                        prob = 1.0;
                        err = sqrt(ss / kk);  // root_mean_squared_error (gard.py:218-219)
                    } else if (mode == 1) {
                        // one-feature OLS on the k analogs (gard.py:194-224), slope 0 when all x are equal.  The x sums
                        // come from the LDS window, the y and cross sums from the prefix differences:
                        //   sum (x - xm)(y - ym) = [rx] + (xbar - xm) [p],  sum (y - ym)^2 = [q] - k m1^2
                        const double x0 = xs[L];
                        double sx = 0.0, sxx = 0.0;
                        {
                            int i = 0;
                            for (; i + 8 <= k; i += 8) xsum_batch<8>(xs + L + i, x0, sx, sxx);
                            if (i + 4 <= k) {
                                xsum_batch<4>(xs + L + i, x0, sx, sxx);
                                i += 4;
                            }
                            for (; i < k; ++i) {
                                const double dx = xs[L + i] - x0;
                                sx += dx;
                                sxx += dx * dx;
                            }
                        }
                        const double2 a = pq[L], b = pq[L + k];
                        const double s1 = b.x - a.x, m1 = s1 / kk, mx = sx / kk, xm = x0 + mx;
                        const double vxx = sxx - kk * mx * mx, vyy = (b.y - a.y) - kk * m1 * m1;
                        const double vxy = (rx[L + k] - rx[L]) + (xbar - xm) * s1;
                        const double slope = vxx > 0.0 ? vxy / vxx : 0.0;
                        double ss = vyy - slope * vxy;
                        pred = (ybar + m1) + (qj - xm) * slope;
                        if (!(ss > ss_floor)) {
                            // (nearly) exact fit or constant analogs: the sums directly, as analog_f1_window_kernel
                            const double* yl = yx_all + c * T + L;
                            const double a0 = yl[0];
                            double t1 = 0.0, txy = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double e = yl[i] - a0;
                                t1 += e;
                                txy += (xs[L + i] - x0) * e;
                            }
                            const double n1 = t1 / kk;
                            const double wxy = txy - kk * mx * n1;
                            const double sl = vxx > 0.0 ? wxy / vxx : 0.0;
                            const double icpt = (a0 + n1) - xm * sl;
                            pred = icpt + qj * sl;
                            ss = 0.0;
                            for (int i = 0; i < k; ++i) {
                                const double r = yl[i] - (icpt + xs[L + i] * sl);
                                ss += r * r;
                            }
                        }
                        prob = 1.0;
                        err = sqrt(ss / kk);  // root_mean_squared_error (gard.py:218-219)
                    } else if (k == 1) {
                        // a single analog (best_analog, or n_analogs = 1: gard.py:291-296): the value itself, no spread
                        const double a1 = yx_all[c * T + L];
                        const bool exc = !pa.has_thresh || a1 > pa.thresh;  // gard.py:307
                        pred = (pa.kind == SD_ANALOG_BEST || exc) ? a1 : 0.0;  // mean / weight of a masked analog: NaN -> 0 (gard.py:341)
                        prob = pa.has_thresh ? (exc ? 1.0 : 0.0) : 1.0;       // gard.py:343, 346
                        err = exc ? 0.0 : nan;                                // gard.py:342, 345
                    } else if (pa.kind == SD_ANALOG_MEAN && !pa.has_thresh) {
                        const double2 a = pq[L], b = pq[L + k];
                        const double m1 = (b.x - a.x) / kk;           // mean of the centred analogs
                        const double var = (b.y - a.y) / kk - m1 * m1;
                        pred = ybar + m1;                            // gard.py:329-333
                        prob = 1.0;                                  // gard.py:346
                        err = sqrt(var > 0.0 ? var : 0.0);           // ddof = 0 (gard.py:345)
                    } else {
                        // weights and / or a threshold need every analog: the window of yx is read from memory (k
                        // consecutive values, cache-resident), the training values come from LDS
                        const double* yl = yx_all + c * T + L;
                        const double a0 = yl[0];
                        double s1 = 0.0, s2 = 0.0, wsum = 0.0, awsum = 0.0;
                        int nexc = 0;
                        int i0 = 0;
                        if (pa.kind == SD_ANALOG_WEIGHT && !pa.has_thresh) {  // (weight_batch: straight-line batches of 8, 4, 2)
                            const double* xw = xs + L;
                            for (; i0 + 8 <= k; i0 += 8) weight_batch<8>(yl + i0, xw + i0, qj, a0, s1, s2, wsum, awsum);
                            if (i0 + 4 <= k) {
                                weight_batch<4>(yl + i0, xw + i0, qj, a0, s1, s2, wsum, awsum);
                                i0 += 4;
                            }
                            if (i0 + 2 <= k) {
                                weight_batch<2>(yl + i0, xw + i0, qj, a0, s1, s2, wsum, awsum);
                                i0 += 2;
                            }
                            nexc = i0;  // (no threshold: every analog counts, gard.py:307; an odd last one takes the generic loop)
                        } else if (pa.has_thresh) {  // (thresh_batch: the same batches with the exceedance count)
                            const double* xw = xs + L;
                            const double th = pa.thresh;
                            if (pa.kind == SD_ANALOG_WEIGHT) {
                                for (; i0 + 8 <= k; i0 += 8) thresh_batch<8, true>(yl + i0, xw + i0, qj, a0, th, s1, s2, wsum, awsum, nexc);
                                if (i0 + 4 <= k) {
                                    thresh_batch<4, true>(yl + i0, xw + i0, qj, a0, th, s1, s2, wsum, awsum, nexc);
                                    i0 += 4;
                                }
                            } else {
                                for (; i0 + 8 <= k; i0 += 8) thresh_batch<8, false>(yl + i0, xw + i0, qj, a0, th, s1, s2, wsum, awsum, nexc);
                                if (i0 + 4 <= k) {
                                    thresh_batch<4, false>(yl + i0, xw + i0, qj, a0, th, s1, s2, wsum, awsum, nexc);
                                    i0 += 4;
                                }
                            }
                        }
                        for (; i0 < k; i0 += kWinBatch) {
                            // two analog values per load (a lane's window is contiguous; the texture path is the limit of
                            // this branch: half the load instructions, half the line requests)
                            double ab[kWinBatch];
#pragma unroll
                            for (int b = 0; b < kWinBatch; b += 2) {
                                if (i0 + b + 1 < k) {
#ifdef SD_MEAN_NOWIN
                                    ab[b] = a0 + (double)(i0 + b);
                                    ab[b + 1] = a0 - (double)(i0 + b);
#else
                                    const f64x2_a8 v = *reinterpret_cast<const f64x2_a8*>(yl + i0 + b);
                                    ab[b] = v.x;
                                    ab[b + 1] = v.y;
#endif
                                } else {
                                    ab[b] = i0 + b < k ? yl[i0 + b] : 0.0;
                                    ab[b + 1] = 0.0;
                                }
                            }
#pragma unroll
                            for (int b = 0; b < kWinBatch; ++b) {
                                const int i = i0 + b;
                                if (i < k) {
                                    const double ai = ab[b], e = ai - a0;
                                    s1 += e;
                                    s2 += e * e;
                                    nexc += (!pa.has_thresh || ai > pa.thresh) ? 1 : 0;  // gard.py:307
                                    if (pa.kind == SD_ANALOG_WEIGHT) {
                                        // w = 1 / distance (gard.py:322-323): v_rcp_f64 + two Newton steps (< 1 ulp)
#ifdef SD_MEAN_NOXS
                                        double d = __builtin_fabs(qj - (a0 + (double)i));
#else
                                        double d = __builtin_fabs(qj - xs[L + i]);
#endif
                                        d = d == 0.0 ? 1e-20 : d;
#ifdef SD_MEAN_NORCP
                                        double r = d;
#else
                                        double r = __builtin_amdgcn_rcp(d);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                                        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
#endif
                                        wsum += r;
                                        awsum += ai * r;
                                    }
                                }
                            }
                        }
                        const bool any_masked = nexc != k;
                        const double m1 = s1 / kk;
                        if (pa.kind == SD_ANALOG_WEIGHT) pred = any_masked ? nan : awsum / wsum;  // gard.py:319-327
                        else pred = any_masked ? nan : a0 + m1;                                    // gard.py:329-333
                        if (pa.has_thresh) {
                            pred = nan_to_num(pred);     // gard.py:341
                            prob = (double)nexc / kk;    // gard.py:343
                        } else {
                            prob = 1.0;  // gard.py:346
                        }
                        if (!any_masked) {
                            const double var = s2 / kk - m1 * m1;
                            err = sqrt(var > 0.0 ? var : 0.0);  // ddof = 0 (gard.py:342,345)
                        }
                    }
                }
                put_out(pa, tq, c, pred, prob, err);
            }
        }
    }
}

// F == 1, PureAnalog 'mean_analogs' without a threshold (the BASELINE configuration) or a single analog.  One 1024-thread
// workgroup per cell; a thread keeps the window starts of its (up to kPhQ) queries in registers and the LDS array
// (n + 1 doubles: all the LDS a workgroup can have at the BASELINE length) is filled three times per cell:
//   1. sorted training values: every query finds its window of k nearest values (two branch-free bisections: position
//      among the values, then window start among the k + 1 candidates), windows with a tie on their boundary take the
//      exact walk;
//   2. exclusive prefix sums of the centred analog values d = yx - mean(y), computed here from yx (blocked partial sums,
//      wave scans) -> window means;
//   3. exclusive prefix sums of d^2 -> spreads.
// Every byte of the state is fetched once, coalesced: xs and yx (8 + 8 bytes per training sample; the fitted state holds
// no prefix sums for this path); every fill keeps all of a thread's loads in flight together (register staging).  The
// workgroup is alone on its CU (LDS), so its memory phases and its LDS phases do not overlap; warming L2 for the next
// phase with early one-word-per-line loads was tried and made the kernel 13 % slower (the lines are gone again before the
// fill: 32 workgroups per XCD stream ~11 MB through a 4 MB L2) and doubled its counted fetch traffic.
// With skip_prob the exceedance-probability column is not written: it is 1 wherever the prediction is not NaN
// (gard.py:346) and the staging transpose fills it in.
constexpr int kPhQ = 16;  // queries per thread and LDS generation (1024 threads: series up to 16 384 queries per pass)
#ifndef SD_SEARCHQ
#define SD_SEARCHQ 2
#endif
// queries a thread searches together: a query is a chain of dependent LDS round trips, and the phase is bound by the number of
// chains in flight, not by LDS bandwidth or instruction issue
constexpr int kSearchQ = SD_SEARCHQ;
static_assert(kSearchQ % 2 == 0 && kPhQ % kSearchQ == 0, "queries are searched in groups that fill the 16-bit window-start pairs");

// a wave-uniform double, pinned to scalar registers (the allocator otherwise keeps such values in vector registers and,
// in this kernel, spills them)
__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
typedef __attribute__((address_space(1))) double global_f64;  // (pointers that travel inside PredictArgs are generic otherwise)

// x / k for an integer-valued k with rk = RN(1 / k): quotient estimate, exact remainder, one correction (correctly rounded;
// the hardware-assisted IEEE division costs ~10x as many instructions and this kernel needs two per query)
__device__ __forceinline__ double div_by(double x, double kk, double rk) {
    const double q = x * rk;
    const double r = __builtin_fma(-q, kk, x);
    return __builtin_fma(r, rk, q);
}

template <int PER>
__global__ void __launch_bounds__(1024) analog_f1_mean3_kernel(const double* __restrict__ Xq /* [C][Tq] */, int64_t Tq, int64_t T,
                                                               int64_t C, const double* __restrict__ xs_all,
                                                               const int32_t* __restrict__ xi_all,
                                                               const double* __restrict__ ybar_all,
                                                               const double* __restrict__ yx_all, const double* __restrict__ Xc,
                                                               const double* __restrict__ yc,
                                                               const int32_t* __restrict__ fit_status, int32_t* status,
                                                               double* scratch_d, int32_t* scratch_i, PredictArgs pa, int skip_prob,
                                                               long long* trace /* development library: phase clocks of block 0 */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ double wsum[16];
    double* buf = reinterpret_cast<double*>(smem_raw);  // n + 1 doubles
    const int nthr = blockDim.x;
    const int n = (int)T, k = pa.k;
    const int per = (n + nthr - 1) / nthr;  // consecutive samples per thread in the prefix sums (<= PER)
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double* sd = scratch_d + (int64_t)blockIdx.x * k * nthr;
    int32_t* si = scratch_i + (int64_t)blockIdx.x * k * nthr;
    // The thread id is re-read behind an opaque barrier in every phase: otherwise the compiler computes the dozens of
    // per-sample indices, predicates and LDS addresses of all phases once, ahead of the cell loop, and spills them.
#define SD_TID()                                        \
    int tid = (int)threadIdx.x;                         \
    asm volatile("" : "+v"(tid));                       \
    const int lane = tid & 63, wave = tid >> 6;         \
    (void)lane;                                         \
    (void)wave
    const double kk = uniform_f64((double)k), rk = uniform_f64(1.0 / (double)k);
    const int M = n - k > 0 ? n - k : 0;  // window starts 0 .. M
    int nsteps = 0;  // window refinement: the range p - k .. p holds at most k + 1 candidates
    while ((1 << nsteps) < (k + 1 < M + 1 ? k + 1 : M + 1)) ++nsteps;
    int64_t step, end;
#ifdef SD_DEV
    int traced = 0;
#define SD_STAMP(slot)                                                                                                              \
    do {                                                                                                                            \
        if (trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && traced < 8) trace[traced * 16 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SD_STAMP(slot) do { } while (0)
#endif
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        const bool active = fit_status[c] == 0;
        const double* xg = xs_all + c * T;
        const double* yx = yx_all + c * T;
        const double ybar = uniform_f64(ybar_all[c]);
        for (int64_t q0 = 0; q0 < Tq; q0 += (int64_t)kPhQ * nthr) {
            // ---- generation 1: sorted training values -> window start of every query
            __syncthreads();
            SD_STAMP(0);
            {
                SD_TID();
                double xv[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    xv[i] = (active && j < n) ? xg[j] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    if (j < n) buf[j] = xv[i];
                }
                if (tid == 0) buf[n] = inf;
            }
            __syncthreads();
            SD_STAMP(1);
            SD_TID();
            // the queries
            const double* qrow = Xq + c * Tq + q0;  // (uniform) queries of this pass
            const int nq = (int)(Tq - q0 < (int64_t)kPhQ * nthr ? Tq - q0 : (int64_t)kPhQ * nthr);
            double qv[kPhQ];
            unsigned hasmask = 0u;
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) {
                const int j = tid + i * nthr;
                qv[i] = 0.0;
                if (j < nq) {
                    qv[i] = qrow[j];
                    hasmask |= 1u << i;
                }
            }
            SD_STAMP(2);
            unsigned Lw2[kPhQ / 2];  // window starts, two 16-bit values per word
#define SD_LW(i) ((int)(((i) & 1) ? (Lw2[(i) >> 1] >> 16) : (Lw2[(i) >> 1] & 0xffffu)))
            unsigned okmask = 0u, nanmask = 0u, walkmask = 0u;  // bit i: prefix-sum statistics / NaN output / exact walk
#pragma unroll
            for (int i0 = 0; i0 < kPhQ; i0 += kSearchQ) {
                double q[kSearchQ];
                bool has[kSearchQ], ok[kSearchQ];
                int lo[kSearchQ];
#pragma unroll
                for (int j = 0; j < kSearchQ; ++j) {
                    has[j] = (hasmask >> (i0 + j)) & 1u;
                    q[j] = qv[i0 + j];
                    ok[j] = active && has[j] && sd_finite(q[j]);
                    if (active && has[j] && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                    if (!ok[j]) q[j] = 0.0;
                }
                window_starts_any<kSearchQ>(buf, k, n, M, q, lo);
#pragma unroll
                for (int j = 0; j < kSearchQ; j += 2) Lw2[(i0 + j) >> 1] = (unsigned)lo[j] | ((unsigned)lo[j + 1] << 16);
#pragma unroll
                for (int j = 0; j < kSearchQ; ++j) {
                    const int i = i0 + j;
                    if (!has[j]) continue;
                    if (!ok[j]) {
                        nanmask |= 1u << i;
                        continue;
                    }
                    const int L = lo[j];
                    const double dL = sq_dist(q[j], buf[L]), dR = sq_dist(q[j], buf[L + k - 1]);
                    const double worst = dL > dR ? dL : dR;
                    const bool sep_l = L == 0 || sq_dist(q[j], buf[L - 1]) > worst;
                    const bool sep_r = L + k >= n || sq_dist(q[j], buf[L + k]) > worst;
                    if (sep_l && sep_r) okmask |= 1u << i;
                    else walkmask |= 1u << i;  // a tie on the window boundary (or a bracket that missed)
                }
            }
            SD_STAMP(3);
#pragma unroll 1
            for (int i = 0; walkmask >> i; ++i)  // exact (rdist, index)-ordered walk; writes its own output
                if ((walkmask >> i) & 1u) {
                    const int64_t tq = q0 + tid + (int64_t)i * nthr;
                    f1_walk_query(0, pa, n, T, c, tq, Xq[c * Tq + tq], xg, xi_all + c * T, Xc + c * T, yc + c * T, sd, si, nthr);
                }
            // ---- y in sorted-x order
            __syncthreads();
            SD_STAMP(4);
            {
                SD_TID();
                double yv[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    yv[i] = j < n ? yx[j] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = tid + i * nthr;
                    if (j < n) buf[j] = yv[i];
                }
            }
            __syncthreads();
            SD_STAMP(5);
            // (uniform) staging rows of this cell and pass: predictions, probabilities, spreads
            global_f64* const orow = (global_f64*)(pa.out + c * 3 * pa.oc_Tq + q0);
            global_f64* const prow = orow + pa.oc_Tq;
            global_f64* const erow = prow + pa.oc_Tq;
            if (k == 1) {
                // a single analog (best_analog, or n_analogs = 1: gard.py:291-296)
#pragma unroll
                for (int i = 0; i < kPhQ; ++i) {
                    const int idx = tid + i * nthr;
                    const bool okq = (okmask >> i) & 1u;
                    if (okq || ((nanmask >> i) & 1u)) {
                        const double a1 = buf[okq ? SD_LW(i) : 0];
                        const bool exc = !pa.has_thresh || a1 > pa.thresh;  // gard.py:307
                        orow[idx] = !okq ? nan : (pa.kind == SD_ANALOG_BEST || exc) ? a1 : 0.0;  // masked mean / weight: NaN -> 0 (gard.py:341)
                        prow[idx] = !okq ? nan : pa.has_thresh ? (exc ? 1.0 : 0.0) : 1.0;          // gard.py:343, 346
                        erow[idx] = !okq ? nan : exc ? 0.0 : nan;                                  // gard.py:342, 345
                    }
                }
                continue;
            }
            // ---- generation 2: exclusive prefix sums of d = yx - mean(y) -> window means
            const int beg = per * tid;
            double d[PER];
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = beg + i;
                d[i] = (i < per && j < n) ? buf[j] - ybar : 0.0;
                a += d[i];
                b += d[i] * d[i];
            }
            double ia = a, ib = b;  // inclusive scans inside the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
                if (lane >= o) {
                    ia += ta;
                    ib += tb;
                }
            }
            __syncthreads();  // every thread has read its block of yx
            if (lane == 63) wsum[wave] = ia;
            __syncthreads();
            double ra = ia - a;  // exclusive prefix at this thread's first sample
            for (int w = 0; w < wave; ++w) ra += wsum[w];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = beg + i;
                if (i < per && j <= n) buf[j] = ra;
                ra += d[i];
            }
            if (tid == nthr - 1 && beg + per == n) buf[n] = ra;  // n = nthr * per: no thread starts at position n
            __syncthreads();
            SD_STAMP(6);
            double m1[kPhQ];
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) m1[i] = (okmask >> i) & 1u ? div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) : 0.0;
            // ---- generation 3: exclusive prefix sums of d^2 -> spreads, outputs
            __syncthreads();
            if (lane == 63) wsum[wave] = ib;
            __syncthreads();
            double rb = ib - b;
            for (int w = 0; w < wave; ++w) rb += wsum[w];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = beg + i;
                if (i < per && j <= n) buf[j] = rb;
                rb += d[i] * d[i];
            }
            if (tid == nthr - 1 && beg + per == n) buf[n] = rb;
            __syncthreads();
            SD_STAMP(7);
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) {
                const int idx = tid + i * nthr;
                double pred, err;
                if ((okmask >> i) & 1u) {
                    const double var = div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) - m1[i] * m1[i];
                    pred = ybar + m1[i];                 // gard.py:329-333
                    err = sqrt(var > 0.0 ? var : 0.0);   // gard.py:345
                } else if ((nanmask >> i) & 1u) {
                    pred = err = nan;
                } else {
                    continue;
                }
                orow[idx] = pred;
                erow[idx] = err;
                if (!skip_prob) prow[idx] = pred != pred ? nan : 1.0;  // gard.py:346
            }
            SD_STAMP(8);
            SD_STAMP(9);
#ifdef SD_DEV
            ++traced;
#endif
        }
    }
#undef SD_STAMP
#undef SD_LW
#undef SD_TID
}

// ------------------------------------------------------------------------------------------------
// fit + predict of the BASELINE case in one kernel (sd_analog_fit_predict_dev: gard.py:58-87 with 273-364 on the same call)
// ------------------------------------------------------------------------------------------------
// The per-cell workgroup that has merged the sorted runs of analog_tile_sort_kernel answers the cell's queries before it
// leaves: the tail of analog_sort2_kernel<K, true> followed by analog_f1_mean3_kernel, with the sorted view handed over on
// chip.  No fitted state exists: xs / xi / yx (20 bytes written and 16 read back per training sample) never travel.  The
// tags of the sorted keys (= xi) are parked in LDS behind the key array; x and y are gathered through them twice, once into
// the sorted order of each.  Arithmetic, summation orders and the window search are those of the two kernels, so the result
// is bit-identical to fit -> predict.  Cells the fast paths cannot decide -- the tag pass fails (equal or nearly equal
// training values), or a query's window is not strictly separated (the exact walk: inlined here it costs the kernel 60 more
// spilled registers and 3 ms per 16 384 cells for a case continuous data never produces) -- are appended to `worklist`; the
// host answers them with the split path.  Pointers are relative to the chunk of cells of this
// launch, `cell0` is the grid index of its first cell.
#ifdef SD_DEV
__device__ long long sd_fused_trace[8 * 16];  // development library: phase clocks of the first 8 cells of workgroup 0 (SD_FUSED_TRACE)
#define SD_FSTAMP(slot)                                                                                                     \
    do {                                                                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0 && traced < 8) sd_fused_trace[traced * 16 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SD_FSTAMP(slot) do { } while (0)
#endif
template <int K>
__global__ void __launch_bounds__(1024) analog_f1_fused_kernel(const double* __restrict__ runs, int np,
                                                               const int32_t* __restrict__ odd_flags,
                                                               const double* __restrict__ Xc, const double* __restrict__ yc,
                                                               const double* __restrict__ Xq /* [C][Tq] */, int64_t Tq, int64_t T,
                                                               int64_t C, const int32_t* __restrict__ fit_status, int32_t* status,
                                                               int32_t* worklist, int32_t* work_count, int64_t cell0,
                                                               PredictArgs pa, int skip_prob) {
    constexpr int PER = K;  // consecutive samples per thread in the prefix sums: ceil(n / 1024) <= K
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ double wsum[16];
    double* buf = reinterpret_cast<double*>(smem_raw);          // np + 1 doubles
    int* xch = reinterpret_cast<int*>(buf + np + 1);            // 1025 ints (co-ranks of the merge rounds; reduction scratch)
    double* red = reinterpret_cast<double*>(xch);
    unsigned short* tagl = reinterpret_cast<unsigned short*>(xch + 1026);  // n tags: training index of the sorted position
    const int nthr = blockDim.x;
    const int n = (int)T, k = pa.k;
    const int per = (n + nthr - 1) / nthr;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
#define SD_TID()                                        \
    int tid = (int)threadIdx.x;                         \
    asm volatile("" : "+v"(tid));                       \
    const int lane = tid & 63, wave = tid >> 6;         \
    (void)lane;                                         \
    (void)wave
#define SD_LW(i) ((int)(((i) & 1) ? (Lw2[(i) >> 1] >> 16) : (Lw2[(i) >> 1] & 0xffffu)))
    const double kk = uniform_f64((double)k), rk = uniform_f64(1.0 / (double)k);
    const int M = n - k > 0 ? n - k : 0;
    int nsteps = 0;
    while ((1 << nsteps) < (k + 1 < M + 1 ? k + 1 : M + 1)) ++nsteps;
    int64_t step, end;
#ifdef SD_DEV
    int traced = 0;
#endif
    for (int64_t c = first_cell(C, &step, &end); c < end; c += step) {
        global_f64* const orow = (global_f64*)(pa.out + c * 3 * pa.oc_Tq);  // (uniform) staging rows: predictions, probabilities, spreads
        global_f64* const prow = orow + pa.oc_Tq;
        global_f64* const erow = prow + pa.oc_Tq;
        if (fit_status[c] != 0) {
            // masked / non-finite training series: every query answers NaN (what the query phase below does for such a cell)
            for (int j = (int)threadIdx.x; j < (int)Tq; j += nthr) {
                orow[j] = nan;
                erow[j] = nan;
                if (!skip_prob || k == 1) prow[j] = nan;
            }
            continue;
        }
        // ---- the sorted runs of 64 * K tagged keys -> merge rounds 6 .. (analog_sort2_kernel<K, true>)
        __syncthreads();
        SD_FSTAMP(0);
        {
            SD_TID();
            const double* rc = runs + c * (int64_t)np;
            double kv[K + 1];
#pragma unroll
            for (int t2 = 0; t2 <= K; ++t2) {
                const int i = tid + t2 * nthr;
                kv[t2] = i < np ? rc[i] : inf;
            }
#pragma unroll
            for (int t2 = 0; t2 <= K; ++t2) {
                const int i = tid + t2 * nthr;
                if (i <= np) buf[i] = kv[t2];
            }
        }
        __syncthreads();
        SD_FSTAMP(1);
        {
            SD_TID();
            sdsort::block_merge_rounds<K>(buf, np, xch, tid, nthr, 6);
            SD_FSTAMP(2);
            bool odd = odd_flags[c] != 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = K * tid + i;
                if (j + 1 < n) odd |= ((__double_as_longlong(buf[j]) ^ __double_as_longlong(buf[j + 1])) >> 14) == 0;
            }
            if (__syncthreads_or(odd) != 0) {
                if (tid == 0) worklist[atomicAdd(work_count, 1)] = (int32_t)(cell0 + c);
                continue;
            }
            // tags -> LDS; x in training order -> buf; gathered through the tags -> registers -> buf = xs
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) tagl[pos] = (unsigned short)(__double_as_longlong(buf[pos]) & kTagMask);
            }
            __syncthreads();
            const double* x = Xc + c * T;
            double xv[K];
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                xv[s2] = pos < n ? x[pos] : 0.0;
            }
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = sd_finite(xv[s2]) ? xv[s2] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                xv[s2] = pos < n ? buf[tagl[pos]] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = xv[s2];
            }
            if (tid == 0) buf[n] = inf;
        }
        __syncthreads();
        SD_FSTAMP(3);
        // ---- generation 1 (analog_f1_mean3_kernel): sorted training values -> window start of every query
        SD_TID();
        const double* qrow = Xq + c * Tq;
        const int nq = (int)Tq;  // (one pass: the host sends Tq <= kPhQ * 1024 here)
        // (the queries of a thread are fetched one group ahead of their use instead of all at once)
        double qn[2 * kSearchQ];
#pragma unroll
        for (int i = 0; i < 2 * kSearchQ; ++i) {
            const int j = tid + i * nthr;
            qn[i] = j < nq ? qrow[j] : 0.0;
        }
        unsigned Lw2[kPhQ / 2];
        unsigned okmask = 0u, nanmask = 0u, walkmask = 0u;
#pragma unroll
        for (int i0 = 0; i0 < kPhQ; i0 += kSearchQ) {
            double q[kSearchQ];
            bool has[kSearchQ], ok[kSearchQ];
            int lo[kSearchQ];
#pragma unroll
            for (int j = 0; j < kSearchQ; ++j) {
                has[j] = tid + (i0 + j) * nthr < nq;
                q[j] = qn[j];
                qn[j] = qn[j + kSearchQ];
                const int jn = tid + (i0 + j + 2 * kSearchQ) * nthr;
                qn[j + kSearchQ] = (i0 + j + 2 * kSearchQ < kPhQ && jn < nq) ? qrow[jn] : 0.0;
                ok[j] = has[j] && sd_finite(q[j]);
                if (has[j] && !ok[j]) atomicOr(&status[c], SDI_NONFINITE);
                if (!ok[j]) q[j] = 0.0;
            }
            window_starts_any<kSearchQ>(buf, k, n, M, q, lo);
#pragma unroll
            for (int j = 0; j < kSearchQ; j += 2) Lw2[(i0 + j) >> 1] = (unsigned)lo[j] | ((unsigned)lo[j + 1] << 16);
#pragma unroll
            for (int j = 0; j < kSearchQ; ++j) {
                const int i = i0 + j;
                if (!has[j]) continue;
                if (!ok[j]) {
                    nanmask |= 1u << i;
                    continue;
                }
                const int L = lo[j];
                const double dL = sq_dist(q[j], buf[L]), dR = sq_dist(q[j], buf[L + k - 1]);
                const double worst = dL > dR ? dL : dR;
                const bool sep_l = L == 0 || sq_dist(q[j], buf[L - 1]) > worst;
                const bool sep_r = L + k >= n || sq_dist(q[j], buf[L + k]) > worst;
                if (sep_l && sep_r) okmask |= 1u << i;
                else walkmask |= 1u << i;
            }
        }
        if (__syncthreads_or(walkmask != 0u) != 0) {  // some window needs the exact walk: the whole cell goes to the split path
            if (tid == 0) worklist[atomicAdd(work_count, 1)] = (int32_t)(cell0 + c);
            continue;
        }
        SD_FSTAMP(4);
        // ---- y in training order -> buf (and its mean, summed as analog_sort2_kernel does); gathered -> buf = yx
        double ybar;
        {
            const double* yy = yc + c * T;
            double yv[K];
            double ysum = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                yv[s2] = pos < n ? yy[pos] : 0.0;
            }
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = yv[s2];
                ysum += yv[s2];
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ysum += __shfl_xor(ysum, o, 64);
            if (lane == 0) red[wave] = ysum;
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                yv[s2] = pos < n ? buf[tagl[pos]] : 0.0;
            }
            double tot = 0.0;
            for (int w = 0; w < 16; ++w) tot += red[w];
            ybar = uniform_f64(tot / (double)n);
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < K; ++s2) {
                const int pos = tid + s2 * nthr;
                if (pos < n) buf[pos] = yv[s2];
            }
        }
        __syncthreads();
        SD_FSTAMP(5);
        if (k == 1) {
            // a single analog (best_analog, or n_analogs = 1: gard.py:291-296)
#pragma unroll
            for (int i = 0; i < kPhQ; ++i) {
                const int idx = tid + i * nthr;
                const bool okq = (okmask >> i) & 1u;
                if (okq || ((nanmask >> i) & 1u)) {
                    const double a1 = buf[okq ? SD_LW(i) : 0];
                    const bool exc = !pa.has_thresh || a1 > pa.thresh;  // gard.py:307
                    orow[idx] = !okq ? nan : (pa.kind == SD_ANALOG_BEST || exc) ? a1 : 0.0;
                    prow[idx] = !okq ? nan : pa.has_thresh ? (exc ? 1.0 : 0.0) : 1.0;
                    erow[idx] = !okq ? nan : exc ? 0.0 : nan;
                }
            }
            continue;
        }
        // ---- generation 2: exclusive prefix sums of d = yx - mean(y) -> window means
        const int beg = per * tid;
        double d[PER];
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = beg + i;
            d[i] = (i < per && j < n) ? buf[j] - ybar : 0.0;
            a += d[i];
            b += d[i] * d[i];
        }
        double ia = a, ib = b;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
            if (lane >= o) {
                ia += ta;
                ib += tb;
            }
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = ia;
        __syncthreads();
        double ra = ia - a;
        for (int w = 0; w < wave; ++w) ra += wsum[w];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = beg + i;
            if (i < per && j <= n) buf[j] = ra;
            ra += d[i];
        }
        if (tid == nthr - 1 && beg + per == n) buf[n] = ra;
        __syncthreads();
        SD_FSTAMP(6);
        double m1[kPhQ];
#pragma unroll
        for (int i = 0; i < kPhQ; ++i) m1[i] = (okmask >> i) & 1u ? div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) : 0.0;
        // ---- generation 3: exclusive prefix sums of d^2 -> spreads, outputs
        SD_FSTAMP(8);
        __syncthreads();
        if (lane == 63) wsum[wave] = ib;
        __syncthreads();
        double rb = ib - b;
        for (int w = 0; w < wave; ++w) rb += wsum[w];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = beg + i;
            if (i < per && j <= n) buf[j] = rb;
            rb += d[i] * d[i];
        }
        if (tid == nthr - 1 && beg + per == n) buf[n] = rb;
        SD_FSTAMP(9);
        __syncthreads();
        SD_FSTAMP(10);
#pragma unroll
        for (int i = 0; i < kPhQ; ++i) {
            const int idx = tid + i * nthr;
            double pred, err;
            if ((okmask >> i) & 1u) {
                const double var = div_by(buf[SD_LW(i) + k] - buf[SD_LW(i)], kk, rk) - m1[i] * m1[i];
                pred = ybar + m1[i];                 // gard.py:329-333
                err = sqrt(var > 0.0 ? var : 0.0);   // gard.py:345
            } else if ((nanmask >> i) & 1u) {
                pred = err = nan;
            } else {
                continue;
            }
            orow[idx] = pred;
            erow[idx] = err;
            if (!skip_prob) prow[idx] = pred != pred ? nan : 1.0;  // gard.py:346
        }
        SD_FSTAMP(7);
#ifdef SD_DEV
        ++traced;
#endif
    }
#undef SD_LW
#undef SD_TID
}
