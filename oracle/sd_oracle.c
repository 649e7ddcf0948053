/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's BCSD hot path.
 *
 * Checker and CPU baseline ("port") for the HIP engine: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this.  It follows, step by step and per cell, what
 * skdownscale/pointwise_models does (citations file:line under /root/reference/skdownscale/
 * pointwise_models), with NumPy's np.sort / np.interp spelled out in C.  Pinned: checked against
 * the golden vectors generated from the real reference (tests/test_oracle_c.py) and against the
 * NumPy restatement oracle/bcsd_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KIND_TAS 0
#define KIND_PR 1
#define ST_OK 0
#define ST_MASKED 1
#define ST_NONFINITE 2
#define ST_BAD_CLIMO 3
#define ALPHA 0.4 /* quantile.py:423 */
#define BETA 0.4  /* quantile.py:424 */
#define N_ENDPOINTS 10 /* quantile.py:426 */

/* np.sort (quantile.py:462) for finite data.  One counting pass over value buckets (a monotone map of the value range onto
 * ~n buckets: order between buckets is exact), then an insertion sort that only ever moves an element inside its bucket.
 * The bucket starts are kept (bucket_index) so that the rank queries of the same segment are a bucket lookup and a short scan
 * instead of a bisection.  ~3x faster than the merge sort it replaces (55 -> 17 us per 1 240 samples). */
typedef struct {
    int nb;         /* buckets (a power of two) */
    double lo, sc;  /* bucket of v = min((int)((v - lo) * sc), nb - 1) */
    int* start;     /* [nb + 1] first sorted position of every bucket */
} bucket_index;
static __thread int* tl_ints = NULL; /* per-thread integer workspace: start[nb + 1], pos[nb], bucket of every sample [n] */
static __thread int tl_cap = 0;
static int* tl_workspace(int need) {
    if (need > tl_cap) {
        free(tl_ints);
        tl_ints = (int*)malloc(sizeof(int) * (size_t)need);
        tl_cap = need;
    }
    return tl_ints;
}
static inline int bucket_of(const bucket_index* bi, double v) {
    const double x = (v - bi->lo) * bi->sc; /* compared as a double: an overflowing or NaN product never reaches the cast */
    if (!(x > 0.0)) return 0;
    return x < (double)(bi->nb - 1) ? (int)x : bi->nb - 1;
}
static void sort_doubles_idx(double* a, int n, double* tmp, bucket_index* bi) {
    int nb = 64;
    while (nb < n) nb <<= 1;
    int* ws = tl_workspace(2 * nb + 1 + n);
    int *start = ws, *pos = ws + nb + 1, *bk = ws + 2 * nb + 1;
    double lo = a[0], hi = a[0];
    for (int i = 1; i < n; ++i) {
        lo = a[i] < lo ? a[i] : lo;
        hi = a[i] > hi ? a[i] : hi;
    }
    bi->nb = nb;
    bi->lo = lo;
    bi->sc = hi > lo ? (double)(nb - 1) / (hi - lo) : 0.0;
    if (!(bi->sc <= 1.7e308)) bi->sc = 0.0; /* a denormal range overflows the quotient: one bucket, the insertion pass sorts it */
    bi->start = start;
    memset(start, 0, sizeof(int) * (size_t)(nb + 1));
    for (int i = 0; i < n; ++i) {
        const int b = bucket_of(bi, a[i]);
        bk[i] = b;
        ++start[b + 1];
    }
    for (int b = 0; b < nb; ++b) {
        start[b + 1] += start[b];
        pos[b] = start[b];
    }
    for (int i = 0; i < n; ++i) tmp[pos[bk[i]]++] = a[i];
    for (int i = 1; i < n; ++i) { /* inversions exist inside buckets only */
        const double v = tmp[i];
        int j = i - 1;
        while (j >= 0 && tmp[j] > v) { tmp[j + 1] = tmp[j]; --j; }
        tmp[j + 1] = v;
    }
    memcpy(a, tmp, sizeof(double) * (size_t)n);
}
static void sort_doubles(double* a, int n, double* tmp) {
    bucket_index bi;
    if (n > 0) sort_doubles_idx(a, n, tmp, &bi);
}

/* quantile.py:23-43 plotting_positions, same operation order */
static inline double pp_denom(int n) { return ((double)n + 1.0 - ALPHA) - BETA; }
static inline double pp_at(int i, double denom) { return ((double)(i + 1) - ALPHA) / denom; }

/* sklearn LinearRegression on one feature = centred least squares (quantile.py:535-543) */
static void ols_line(const double* ys, int first, int e, double denom, double* slope, double* icpt) {
    double xm = 0.0, ym = 0.0, sxx = 0.0, sxy = 0.0;
    for (int i = 0; i < e; ++i) { xm += pp_at(first + i, denom); ym += ys[first + i]; }
    xm /= e; ym /= e;
    for (int i = 0; i < e; ++i) {
        const double dx = pp_at(first + i, denom) - xm;
        sxx += dx * dx;
        sxy += dx * (ys[first + i] - ym);
    }
    *slope = sxx > 0.0 ? sxy / sxx : 0.0;
    *icpt = ym - *slope * xm;
}

/* CunnaneTransformer.inverse_transform (quantile.py:523-545): np.interp(p, pp, ys, -inf, inf) + OLS tails */
static double inverse_cdf(double p, const double* ys, const double* pp /* plotting positions of the fit, [n] */, int n,
                          const double* tails) {
    if (p < pp[0]) return p * tails[0] + tails[1];
    if (p > pp[n - 1]) return p * tails[2] + tails[3];
    /* the last i with pp[i] <= p (numpy compiled_base.c arr_interp finds it by bisection): pp is the affine grid
     * (i + 1 - ALPHA) / denom, so i = floor(p * denom + ALPHA) - 1 up to the rounding of that product -- two guarded steps */
    const double denom = ((double)n + 1.0 - ALPHA) - BETA;
    int i = (int)floor(p * denom + ALPHA) - 1;
    i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    while (i + 1 < n && pp[i + 1] <= p) ++i;
    while (i > 0 && pp[i] > p) --i;
    const double pi = pp[i];
    if (i == n - 1 || pi == p) return ys[i];
    const double slope = (ys[i + 1] - ys[i]) / (pp[i + 1] - pi);
    return slope * (p - pi) + ys[i];
}

/* QuantileMapper.transform for one segment (quantile.py:109-147): u[m] -> q[m]; work: 3 * max(m, n) doubles */
static void qm_segment(const double* u, int m, const double* ys, int n, double* q, double* work, int nmax) {
    double* tmp = work + nmax;
    double* ppn = work + 2 * nmax; /* plotting positions of the fitted CDF (quantile.py:457-463) */
    memcpy(work, u, sizeof(double) * m);
    bucket_index bi;
    sort_doubles_idx(work, m, tmp, &bi); /* quantile.py:462 via fit_transform 505-521 */
    const double dm = pp_denom(m), dn = pp_denom(n);
    for (int i = 0; i < n; ++i) ppn[i] = pp_at(i, dn);
    double tails[4] = {0, 0, 0, 0};
    const int e = n < N_ENDPOINTS ? n : N_ENDPOINTS;
    if (m > n) {
        ols_line(ys, 0, e, dn, &tails[0], &tails[1]);
        ols_line(ys, n - e, e, dn, &tails[2], &tails[3]);
    }
    for (int j = 0; j < m; ++j) {
        /* quantile.py:488 np.interp on own sorted data: the last sorted position holding a value <= u[j] -- it lies in u[j]'s bucket */
        const int bq = bucket_of(&bi, u[j]);
        int r = bi.start[bq + 1] - 1;
        while (work[r] > u[j]) --r;
        q[j] = inverse_cdf(pp_at(r, dm), ys, ppn, n, tails);
    }
}

/* One cell: fit (bcsd.py:197-228 / 115-147) + predict (bcsd.py:230-269 / 149-185).
 * x, y: [T] strided by ld; xp: [Tp] strided by ldp; out: [Tp] strided by ldo. */
static int bcsd_cell(int kind, const double* x, const double* y, int64_t ld, const double* xp, int64_t ldp,
                     const int32_t* ord, const int64_t* off, const int32_t* ordp, const int64_t* offp, int G,
                     int return_anoms, double* out, int64_t ldo, double* buf /* 8*nmax */, int nmax) {
    double* ys = buf;            /* sorted y segment */
    double* xg = buf + nmax;     /* predict segment  */
    double* u = buf + 2 * nmax;
    double* q = buf + 3 * nmax;
    double* shiftv = buf + 4 * nmax;
    double* work = buf + 5 * nmax; /* 3 * nmax */
    const double first = x ? x[0] : y[0];
    if (first != first) return ST_MASKED; /* core.py:35-37 */
    int status = ST_OK;
    /* base.py:18-20: validation happens before any arithmetic */
    const int64_t T = off[G], Tp = offp[G];
    for (int64_t t = 0; t < T; ++t)
        if (!isfinite(y[t * ld]) || (x && !isfinite(x[t * ld]))) return ST_NONFINITE;
    for (int g = 0; g < G && status == ST_OK; ++g) { /* climatology check precedes the mapper fit (bcsd.py:138-141) */
        const int n = (int)(off[g + 1] - off[g]);
        if (kind == KIND_PR && return_anoms && n > 0) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += y[(int64_t)ord[off[g] + i] * ld];
            if (s / n <= 0.0) status = ST_BAD_CLIMO;
        }
    }
    if (status != ST_OK) return status;
    for (int64_t t = 0; t < Tp; ++t)
        if (!isfinite(xp[t * ldp])) return ST_NONFINITE;
    for (int g = 0; g < G; ++g) {
        const int n = (int)(off[g + 1] - off[g]), m = (int)(offp[g + 1] - offp[g]);
        if (n == 0 || m == 0) continue;
        double xc = 0.0, yc = 0.0;
        for (int i = 0; i < n; ++i) {
            const int64_t t = ord[off[g] + i];
            ys[i] = y[t * ld];
            yc += ys[i];
            if (kind == KIND_TAS) xc += x[t * ld];
        }
        xc /= n; /* bcsd.py:222 */
        yc /= n; /* bcsd.py:223 / 138 */
        sort_doubles(ys, n, work); /* quantile.py:462 np.sort */
        for (int j = 0; j < m; ++j) xg[j] = xp[(int64_t)ordp[offp[g] + j] * ldp];
        if (kind == KIND_TAS) {
            for (int j = 0; j < m; ++j) { /* bcsd.py:247-256 */
                const int lo = j - 4 < 0 ? 0 : j - 4, hi = j + 5 > m ? m : j + 5;
                double s = 0.0;
                for (int i = lo; i < hi; ++i) s += xg[i];
                const double shift = s / (hi - lo) - xc;
                u[j] = xg[j] - shift;
                shiftv[j] = shift;
            }
            qm_segment(u, m, ys, n, q, work, nmax); /* bcsd.py:260 */
            for (int j = 0; j < m; ++j) {
                double r = shiftv[j] + q[j];      /* bcsd.py:263 */
                if (return_anoms) r = r - yc;     /* bcsd.py:266-267 */
                out[(int64_t)ordp[offp[g] + j] * ldo] = r;
            }
        } else {
            qm_segment(xg, m, ys, n, q, work, nmax); /* bcsd.py:167 */
            for (int j = 0; j < m; ++j) out[(int64_t)ordp[offp[g] + j] * ldo] = return_anoms ? q[j] / yc : q[j]; /* bcsd.py:170-185 */
        }
    }
    return ST_OK;
}

static int build_table(const int32_t* gid, int64_t T, int G, int32_t* ord, int64_t* off) {
    int nmax = 0;
    memset(off, 0, sizeof(int64_t) * (G + 1));
    for (int64_t t = 0; t < T; ++t) off[gid[t] + 1]++;
    for (int g = 0; g < G; ++g) {
        if (off[g + 1] > nmax) nmax = (int)off[g + 1];
        off[g + 1] += off[g];
    }
    int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * G);
    memcpy(cur, off, sizeof(int64_t) * G);
    for (int64_t t = 0; t < T; ++t) ord[cur[gid[t]]++] = (int32_t)t;
    free(cur);
    return nmax;
}

/* Grid driver (core.py:86-96,137-141): loops the per-cell model over the cell axis.
 * X may be NULL for PR.  Fields [T,C] / [Tp,C], cells contiguous (ld = C).  Returns 0. */
int sdo_bcsd_fit_predict(int kind, const double* X, const double* y, const double* Xp, const int32_t* gid,
                         const int32_t* gid_p, int G, int64_t T, int64_t Tp, int64_t C, int return_anoms, double* out,
                         int32_t* status, int nthreads) {
    int32_t* ord = (int32_t*)malloc(sizeof(int32_t) * T);
    int32_t* ordp = (int32_t*)malloc(sizeof(int32_t) * Tp);
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (G + 1));
    int64_t* offp = (int64_t*)malloc(sizeof(int64_t) * (G + 1));
    int nmax = build_table(gid, T, G, ord, off);
    const int nmaxp = build_table(gid_p, Tp, G, ordp, offp);
    if (nmaxp > nmax) nmax = nmaxp;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    /* Cells are processed in panels of PW adjacent cells: a thread copies the panel's [T, PW] strip into per-cell
     * series (rows of PW * 8 contiguous bytes: a 64-byte gather per row kept 128 threads waiting on the TLB), runs the
     * per-cell model on the contiguous series and scatters the result strip back.  Panels are dealt out statically,
     * each thread walks a contiguous range of cells. */
    enum { PW = 64 };
    const int64_t npan = (C + PW - 1) / PW;
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * 8 * (size_t)nmax);
        double* cx = (double*)malloc(sizeof(double) * PW * (size_t)(2 * T + 2 * Tp));
        double* cy = cx + (size_t)PW * T;
        double* cp = cy + (size_t)PW * T;
        double* co = cp + (size_t)PW * Tp;
#pragma omp for schedule(static)
        for (int64_t b = 0; b < npan; ++b) {
            const int64_t c0 = b * PW;
            const int w = (int)((C - c0) < PW ? (C - c0) : PW);
            /* (blocks of TB time steps: 64 write streams 117 KB apart, advanced one element at a time, miss the TLB on every
             * store; TB consecutive doubles per stream and block do not) */
            enum { TB = 64 };
            for (int64_t tb = 0; tb < T; tb += TB) {
                const int64_t te = tb + TB < T ? tb + TB : T;
                for (int k = 0; k < w; ++k) {
                    if (X) {
                        double* d = cx + k * T;
                        const double* sp = X + c0 + k;
                        for (int64_t t = tb; t < te; ++t) d[t] = sp[t * C];
                    }
                    double* d = cy + k * T;
                    const double* sp = y + c0 + k;
                    for (int64_t t = tb; t < te; ++t) d[t] = sp[t * C];
                }
            }
            for (int64_t tb = 0; tb < Tp; tb += TB) {
                const int64_t te = tb + TB < Tp ? tb + TB : Tp;
                for (int k = 0; k < w; ++k) {
                    double* d = cp + k * Tp;
                    const double* sp = Xp + c0 + k;
                    for (int64_t t = tb; t < te; ++t) d[t] = sp[t * C];
                }
            }
            for (int k = 0; k < w; ++k) {
                const int st = bcsd_cell(kind, X ? cx + k * T : NULL, cy + k * T, 1, cp + k * Tp, 1, ord, off, ordp, offp, G,
                                         return_anoms, co + k * Tp, 1, buf, nmax);
                status[c0 + k] = st;
                if (st != ST_OK)
                    for (int64_t t = 0; t < Tp; ++t) co[k * Tp + t] = NAN; /* core.py:119 */
            }
            for (int64_t tb = 0; tb < Tp; tb += TB) {
                const int64_t te = tb + TB < Tp ? tb + TB : Tp;
                for (int k = 0; k < w; ++k) {
                    const double* sp = co + k * Tp;
                    double* d = out + c0 + k;
                    for (int64_t t = tb; t < te; ++t) d[t * C] = sp[t];
                }
            }
        }
        free(buf);
        free(cx);
        free(tl_ints);
        tl_ints = NULL;
        tl_cap = 0;
    }
    free(ord); free(ordp); free(off); free(offp);
    return 0;
}

int sdo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
