// sd_analog_epilogue.h -- part of the translation unit csrc/sd_analog.hip (included there, inside its unnamed namespace; not a
// stand-alone header).  epilogues shared by the predict kernels: PureAnalog statistics, per-query least squares, logistic exceedance model.

// ------------------------------------------------------------------------------------------------
// epilogues (run by the thread that owns the query; lists are [k][nthr] in scratch)
// ------------------------------------------------------------------------------------------------
struct PredictArgs;
__device__ __forceinline__ void put_out(const PredictArgs& pa, int64_t tq, int64_t c, double pred, double prob, double err);

struct PredictArgs {
    int k, kind, has_thresh;
    double thresh;
    const int32_t* sample;  // device [Tq, ld_s] or null
    int64_t ld_s;
    double* out;            // [Tq,3,ld_out]; windowed path: cell-major staging [C][3][Tq] (oc_Tq > 0)
    int64_t ld_out;
    int64_t oc_Tq;          // > 0: out is the cell-major staging buffer of a Tq-long query series
    int64_t* inds;          // [Tq,k,ld_out] or null
    double* dist;           // [Tq,k,ld_out] or null
    int32_t* one_class;     // per-cell status words (predict side): SDI_ONE_CLASS is set here
    int skip_prob = 0;      // staging path: the probability plane is not written (the staging transpose derives it from the predictions)
};

__device__ __forceinline__ void put_out(const PredictArgs& pa, int64_t tq, int64_t c, double pred, double prob, double err) {
    if (pa.oc_Tq > 0) {  // consecutive queries of a cell are consecutive in memory: coalesced across the workgroup
        double* o = pa.out + c * 3 * pa.oc_Tq + tq;
        o[0] = pred;
        if (!pa.skip_prob) o[pa.oc_Tq] = prob;
        o[2 * pa.oc_Tq] = err;
    } else {
        pa.out[(tq * 3 + 0) * pa.ld_out + c] = pred;
        pa.out[(tq * 3 + 1) * pa.ld_out + c] = prob;
        pa.out[(tq * 3 + 2) * pa.ld_out + c] = err;
    }
}

__device__ __forceinline__ double nan_to_num(double v) {
    if (v != v) return 0.0;
    if (v == __longlong_as_double(0x7ff0000000000000ll)) return 1.7976931348623157e308;
    if (v == __longlong_as_double(0xfff0000000000000ll)) return -1.7976931348623157e308;
    return v;
}

// PureAnalog statistics for one query (gard.py:301-346).  a[i] = analog values in neighbour order,
// read through `av(i)`; rd(i) = reduced distance.
template <typename AV, typename RD>
__device__ void pure_analog_stats(const PredictArgs& pa, int k, int kind, int sample_i, AV av, RD rd, double* pred,
                                  double* prob, double* err) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double sum = 0.0, wsum = 0.0, awsum = 0.0;
    int nexc = 0;
    bool any_masked = false;
    for (int i = 0; i < k; ++i) {
        const double a = av(i);
        const bool exc = !pa.has_thresh || a > pa.thresh;  // gard.py:307
        nexc += exc ? 1 : 0;
        any_masked |= !exc;
        sum += a;
        if (kind == SD_ANALOG_WEIGHT) {
            const double d = sqrt(rd(i));
            const double w = 1.0 / (d == 0.0 ? 1e-20 : d);  // gard.py:322-323
            wsum += w;
            awsum += a * w;
        }
    }
    double p;
    if (kind == SD_ANALOG_BEST) p = av(0);                               // gard.py:311
    else if (kind == SD_ANALOG_SAMPLE) p = av(sample_i);                 // gard.py:313-317
    else if (kind == SD_ANALOG_WEIGHT) p = any_masked ? nan : awsum / wsum;  // gard.py:319-327 (NaN-masked average)
    else p = any_masked ? nan : sum / (double)k;                         // gard.py:329-333
    if (pa.has_thresh) {
        p = nan_to_num(p);  // gard.py:341
        *prob = (double)nexc / (double)k;  // gard.py:343
    } else {
        *prob = 1.0;  // gard.py:346
    }
    if (any_masked) {
        *err = nan;  // gard.py:342 plain .std() of a NaN-masked row
    } else {
        const double mean = sum / (double)k;
        double ss = 0.0;
        for (int i = 0; i < k; ++i) {
            const double d = av(i) - mean;
            ss += d * d;
        }
        *err = sqrt(ss / (double)k);  // ddof = 0 (gard.py:342,345)
    }
    *pred = p;
}

// AnalogRegression for one query (gard.py:194-224): centred normal equations over the analogs selected by `use`
// (all of them without a threshold; those above it otherwise, gard.py:215: ne of them, ne >= 1).
template <typename XV, typename YV, typename USE>
__device__ void analog_regression(int k, int F, XV xv /* (i,f) */, YV yv /* (i) */, USE use /* (i) */, int ne, const double* q,
                                  double* pred, double* err) {
    double xm[kMaxF], A[kMaxF][kMaxF + 1], coef[kMaxF];
    double ym = 0.0;
    for (int f = 0; f < F; ++f) xm[f] = 0.0;
    for (int i = 0; i < k; ++i) {
        if (!use(i)) continue;
        ym += yv(i);
        for (int f = 0; f < F; ++f) xm[f] += xv(i, f);
    }
    ym /= (double)ne;
    for (int f = 0; f < F; ++f) xm[f] /= (double)ne;
    for (int f = 0; f < F; ++f)
        for (int g = 0; g <= F; ++g) A[f][g] = 0.0;
    for (int i = 0; i < k; ++i) {
        if (!use(i)) continue;
        const double dy = yv(i) - ym;
        for (int f = 0; f < F; ++f) {
            const double df = xv(i, f) - xm[f];
            for (int g = f; g < F; ++g) A[f][g] += df * (xv(i, g) - xm[g]);
            A[f][F] += df * dy;
        }
    }
    for (int f = 0; f < F; ++f)
        for (int g = 0; g < f; ++g) A[f][g] = A[g][f];
    sdlsq::minnorm_solve(F, A, coef);  // like LinearRegression's lstsq (gard.py:215-217)
    double icpt = ym;
    for (int f = 0; f < F; ++f) icpt -= xm[f] * coef[f];
    double p = icpt;
    for (int f = 0; f < F; ++f) p += q[f] * coef[f];
    double ss = 0.0;
    for (int i = 0; i < k; ++i) {
        if (!use(i)) continue;
        double yh = icpt;
        for (int f = 0; f < F; ++f) yh += xv(i, f) * coef[f];
        const double d = yv(i) - yh;
        ss += d * d;
    }
    *pred = p;
    *err = sqrt(ss / (double)ne);  // root_mean_squared_error (gard.py:218-219)
}

// LogisticRegression() of sklearn (L2 penalty, C = 1, intercept not penalised; gard.py:177, 204-212) on the k analogs of a
// query: labels t_i = (y_i > thresh), both classes present.  Exact minimiser of
//     sum_i [log(1 + exp(z_i)) - t_i z_i] + |w|^2 / 2,   z_i = w . x_i + b
// by damped Newton steps (Cholesky of the (F+1) x (F+1) Hessian, step halved until the objective does not increase);
// sklearn stops its L-BFGS at a gradient of 1e-4 of the mean loss, i.e. within ~1e-3 of this optimum.  Returns z(q).
template <typename XV, typename TV>
__device__ double logistic_at_query(int k, int F, XV xv /* (i,f) */, TV tv /* (i) -> 0/1 */, const double* q) {
    const int n = F + 1;
    double th[kMaxF + 1], g[kMaxF + 1], d[kMaxF + 1], trial[kMaxF + 1], H[kMaxF + 1][kMaxF + 1];
    for (int a = 0; a < n; ++a) th[a] = 0.0;
    auto objective = [&](const double* t) {
        double f = 0.0;
        for (int i = 0; i < k; ++i) {
            double z = t[F];
            for (int a = 0; a < F; ++a) z += t[a] * xv(i, a);
            f += sdlsq::softplus(z) - (tv(i) ? z : 0.0);
        }
        for (int a = 0; a < F; ++a) f += 0.5 * t[a] * t[a];
        return f;
    };
    double f = objective(th);
    for (int it = 0; it < 60; ++it) {
        for (int a = 0; a < n; ++a) {
            g[a] = a < F ? th[a] : 0.0;
            for (int b = 0; b < n; ++b) H[a][b] = (a == b && a < F) ? 1.0 : 0.0;
        }
        for (int i = 0; i < k; ++i) {
            double z = th[F];
            for (int a = 0; a < F; ++a) z += th[a] * xv(i, a);
            const double sg = sdlsq::sigmoid(z), r = sg - (tv(i) ? 1.0 : 0.0), w = sg * (1.0 - sg);
            for (int a = 0; a < n; ++a) {
                const double xa = a < F ? xv(i, a) : 1.0;
                g[a] += r * xa;
                for (int b = 0; b <= a; ++b) H[a][b] += w * xa * (b < F ? xv(i, b) : 1.0);
            }
        }
        double gmax = 0.0;
        for (int a = 0; a < n; ++a) gmax = fmax(gmax, fabs(g[a]));
        if (gmax <= 1e-12 * (double)k) break;
        for (int a = 0; a < n; ++a) {
            H[a][a] += 1e-12;
            g[a] = -g[a];
        }
        if (!sdlsq::chol_solve(n, H, g, d)) break;
        double step = 1.0, fn = f;
        for (;;) {
            for (int a = 0; a < n; ++a) trial[a] = th[a] + step * d[a];
            fn = objective(trial);
            if (fn <= f || step < 1e-10) break;
            step *= 0.5;
        }
        for (int a = 0; a < n; ++a) th[a] = trial[a];
        f = fn;
    }
    double z = th[F];
    for (int a = 0; a < F; ++a) z += th[a] * q[a];
    return z;
}

// mode 0 = PureAnalog, 1 = AnalogRegression.  Lists in scratch: sd[i*nthr + tid], si[...].
template <typename IT>
__device__ void finish_query(int mode, const PredictArgs& pa, int F, int64_t T, int64_t c, int64_t tq, const double* q,
                             const double* __restrict__ Xc_cell, const double* __restrict__ yc_cell,
                             const double* sd, const IT* si, int nthr, bool cell_active) {
    const int tid = threadIdx.x;
    const int k = pa.k;
    double pred, prob = 1.0, err;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (!cell_active) {
        pred = prob = err = nan;
    } else if (mode == 0) {
        const int s = (pa.kind == SD_ANALOG_SAMPLE && pa.sample) ? pa.sample[tq * pa.ld_s + c] : 0;
        pure_analog_stats(
            pa, k, pa.kind, s < 0 ? 0 : (s >= k ? k - 1 : s), [&](int i) { return yc_cell[si[(int64_t)i * nthr + tid]]; },
            [&](int i) { return sd[(int64_t)i * nthr + tid]; }, &pred, &prob, &err);
    } else {
        auto xv = [&](int i, int f) { return Xc_cell[(int64_t)f * T + si[(int64_t)i * nthr + tid]]; };
        auto yv = [&](int i) { return yc_cell[si[(int64_t)i * nthr + tid]]; };
        if (pa.has_thresh) {  // gard.py:201-219
            auto exc = [&](int i) { return yv(i) > pa.thresh; };
            int ne = 0;
            for (int i = 0; i < k; ++i) ne += exc(i) ? 1 : 0;
            if (ne == 0) {
                // every analog at or below the threshold: the reference's LogisticRegression.fit raises (one class only)
                if (pa.one_class) atomicOr(&pa.one_class[c], SDI_ONE_CLASS);
                pred = prob = err = nan;
            } else {
                if (ne < k) prob = 1.0 - sdlsq::sigmoid(logistic_at_query(k, F, xv, exc, q));  // predict_proba(X)[0, 0] (gard.py:210)
                analog_regression(k, F, xv, yv, exc, ne, q, &pred, &err);
            }
        } else {
            analog_regression(k, F, xv, yv, [](int) { return true; }, k, q, &pred, &err);
        }
    }
    put_out(pa, tq, c, pred, prob, err);
    if (cell_active && pa.inds)
        for (int i = 0; i < k; ++i) pa.inds[(tq * k + i) * pa.ld_out + c] = si[(int64_t)i * nthr + tid];
    if (cell_active && pa.dist)
        for (int i = 0; i < k; ++i) pa.dist[(tq * k + i) * pa.ld_out + c] = sqrt(sd[(int64_t)i * nthr + tid]);
}
