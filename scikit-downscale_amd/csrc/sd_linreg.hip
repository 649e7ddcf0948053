// PureRegression (gard.py:367-504), batched over the cell axis: one ordinary least-squares fit of y on the F features per
// cell (sklearn LinearRegression = centred lstsq, gard.py:439-442), fit_error_ = RMSE of that fit; predict returns
// [pred, exceedance_prob, fit_error_] per sample (gard.py:462-470).  With a threshold (gard.py:416-437): a logistic
// regression of (y > thresh) on the features gives the exceedance probability (exact minimiser of sklearn's default
// L2-penalised objective by damped Newton steps; sklearn's L-BFGS stops within ~1e-3 of it), and the linear model and its
// RMSE use the exceeding samples only.  A cell whose samples all exceed drops its threshold like the reference does
// (probability 1); a cell without any exceeding sample is flagged (the reference's LinearRegression gets 0 samples).
//
// Fields stay in their time-major layout: a workgroup owns 64 adjacent cells x 8 time slices, every load is a
// 512-byte row fragment, so fit is two streaming passes (shifted sums and cross products, then the residuals) and
// predict one.
#include <algorithm>
#include <vector>

#include "sd_internal.h"
#include "sd_lsq.h"

struct sd_linreg_state {
    sd_ctx* ctx = nullptr;
    int64_t T = 0, C = 0;
    int F = 0;
    double* coef = nullptr;       // device [F][C]
    double* intercept = nullptr;  // device [C]
    double* rmse = nullptr;       // device [C]
    int32_t* status = nullptr;    // device [C] internal bitmask
    int has_thresh = 0;
    double thresh = 0.0;
    double* logit = nullptr;        // device [F+1][C]: logistic coefficients and intercept (has_thresh)
    int32_t* thresh_off = nullptr;  // device [C]: 1 = one class only, threshold dropped (gard.py:426-437)
};

namespace {

constexpr int kMaxF = sdlsq::kMaxF;
constexpr int kCells = 64, kSlices = 8;
constexpr int kFitUnroll = 4;  // time steps of a thread whose loads are in flight together in the fit pass

__device__ __forceinline__ bool lr_finite(double v) { return (__double_as_longlong(v) & 0x7ff0000000000000ll) != 0x7ff0000000000000ll; }

// sum over the time slices of a workgroup: part[slice][cell] -> every thread gets the total of its cell
__device__ __forceinline__ double slice_sum(double v, double* part, int cx, int ty) {
    __syncthreads();
    part[ty * kCells + cx] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < kSlices; ++s) t += part[s * kCells + cx];
    return t;
}

template <int F>
__global__ void __launch_bounds__(kCells * kSlices) linreg_fit_kernel(const double* __restrict__ X, const double* __restrict__ y,
                                                                      int64_t ld, int64_t T, int64_t C, int has_thresh, double thresh,
                                                                      double* __restrict__ coef_out, double* __restrict__ icpt_out,
                                                                      double* __restrict__ rmse_out, int32_t* __restrict__ status,
                                                                      int32_t* __restrict__ thresh_off) {
    __shared__ double part[kSlices * kCells];
    __shared__ double model[(F + 1) * kCells];  // coefficients and intercept of the tile's cells
    __shared__ double rss1[kCells];             // residual sum of squares from the sums of pass 1, or -1: the cell takes pass 2
    const int cx = threadIdx.x % kCells, ty = threadIdx.x / kCells;
    const int64_t c = (int64_t)blockIdx.x * kCells + cx;
    const bool live = c < C;
    // pass 1: sums and cross products of the data shifted by the cell's first sample (one pass; the shift keeps the
    // centring subtraction below benign), plus the mask / finite bookkeeping of core.py:35-37, base.py:18-20
    double x0[F], sx[F], S[F][F], b[F], sy = 0.0, syy = 0.0;
    bool bad = false;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        x0[f] = live ? X[(int64_t)f * ld + c] : 0.0;
        sx[f] = 0.0;
        b[f] = 0.0;
#pragma unroll
        for (int g = 0; g < F; ++g) S[f][g] = 0.0;
    }
    const double y0 = live ? y[c] : 0.0;
    double cnt = 0.0;  // samples entering the linear model: all, or those above the threshold (gard.py:439)
    if (live)
        for (int64_t t0 = ty; t0 < T; t0 += kSlices * kFitUnroll) {
            // (the loads of kFitUnroll time steps are requested together; the sums run in time order as before)
            double wv[kFitUnroll], xv[kFitUnroll][F];
#pragma unroll
            for (int u = 0; u < kFitUnroll; ++u) {
                const int64_t t = t0 + (int64_t)u * kSlices;
                const bool ok = t < T;
                wv[u] = ok ? y[t * ld + c] : y0;
#pragma unroll
                for (int f = 0; f < F; ++f) xv[u][f] = ok ? X[(t * F + f) * ld + c] : x0[f];
            }
#pragma unroll
            for (int u = 0; u < kFitUnroll; ++u) {
                if (t0 + (int64_t)u * kSlices >= T) break;
                double d[F];
                const double w = wv[u];
                bad |= !lr_finite(w);
                const bool in = !has_thresh || w > thresh;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const double v = xv[u][f];
                    bad |= !lr_finite(v);
                    d[f] = v - x0[f];
                }
                if (!in) continue;
                cnt += 1.0;
                const double e = w - y0;
                sy += e;
                syy += e * e;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    sx[f] += d[f];
                    b[f] += d[f] * e;
#pragma unroll
                    for (int g = f; g < F; ++g) S[f][g] += d[f] * d[g];
                }
            }
        }
    if (live && ty == 0 && x0[0] != x0[0]) atomicOr(&status[c], SDI_MASKED);
    if (live && bad) atomicOr(&status[c], SDI_NONFINITE);
    const double n = slice_sum(cnt, part, cx, ty);
    if (has_thresh && live && ty == 0) {
        thresh_off[c] = n == (double)T ? 1 : 0;                  // every sample exceeds: the reference drops the threshold
        if (n == 0.0) atomicOr(&status[c], SDI_ONE_CLASS);       // none does: LinearRegression would get 0 samples
    }
    double xm[F], dm[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        dm[f] = slice_sum(sx[f], part, cx, ty) / n;  // mean of the shifted feature
        xm[f] = x0[f] + dm[f];
    }
    const double em = slice_sum(sy, part, cx, ty) / n, ym = y0 + em;
    const double Syy = slice_sum(syy, part, cx, ty) - n * em * em;  // centred: sum e^2 - n mean(e)^2
    double A[kMaxF][kMaxF + 1];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        A[f][F] = slice_sum(b[f], part, cx, ty) - n * dm[f] * em;  // centred: sum d e - n mean(d) mean(e)
#pragma unroll
        for (int g = f; g < F; ++g) {
            const double v = slice_sum(S[f][g], part, cx, ty) - n * dm[f] * dm[g];
            A[f][g] = v;
            A[g][f] = v;
        }
    }
    if (ty == 0) {
        double coef[kMaxF], sxy[F];
#pragma unroll
        for (int f = 0; f < F; ++f) sxy[f] = A[f][F];  // (the solver works in place)
        sdlsq::minnorm_solve(F, A, coef);
        double icpt = ym;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            icpt -= xm[f] * coef[f];
            model[f * kCells + cx] = coef[f];
        }
        model[F * kCells + cx] = icpt;
        // residual sum of squares of the least-squares solution from the sums of pass 1: Syy - coef . Sxy (the normal equations).
        // Its relative error is ~ eps Syy / rss: far inside the parity contract unless the fit is (nearly) exact -- such a cell,
        // and every cell with a threshold (subset fit: kept as it was), sums its residuals in a second pass.  The decision is
        // per cell: a cell's result does not depend on the cells that share its tile.
        double rss = Syy;
#pragma unroll
        for (int f = 0; f < F; ++f) rss -= coef[f] * sxy[f];
        const bool second = has_thresh || !(rss > 1e-9 * Syy);
        rss1[cx] = second ? -1.0 : rss;
    }
    __syncthreads();
    // pass 2 (cells that need it): residuals of the fit (root_mean_squared_error, gard.py:441-442)
    double cf[F];
#pragma unroll
    for (int f = 0; f < F; ++f) cf[f] = model[f * kCells + cx];
    const double icpt = model[F * kCells + cx];
    const double rss_sums = rss1[cx];
    const bool second = rss_sums < 0.0;
    double ss = 0.0;
    if (live && second)
        for (int64_t t = ty; t < T; t += kSlices) {
            const double w = y[t * ld + c];
            if (has_thresh && !(w > thresh)) continue;
            double yh = icpt;
#pragma unroll
            for (int f = 0; f < F; ++f) yh += X[(t * F + f) * ld + c] * cf[f];
            const double r = w - yh;
            ss += r * r;
        }
    ss = slice_sum(ss, part, cx, ty);
    if (live && ty == 0) {
#pragma unroll
        for (int f = 0; f < F; ++f) coef_out[(int64_t)f * C + c] = cf[f];
        icpt_out[c] = icpt;
        rmse_out[c] = sqrt((second ? ss : rss_sums) / n);
    }
}

// LogisticRegression() of sklearn (L2, C = 1, intercept not penalised) of (y > thresh) on the features over the whole
// series of a cell (gard.py:416-420): minimise  sum_t [log(1 + exp(z_t)) - l_t z_t] + |w|^2 / 2,  z_t = w . x_t + b,
// by damped Newton steps.  Same tiling as the linear fit; every step is one streaming pass for gradient and Hessian and
// one per trial step length for the objective.  Cells with one class only keep zero coefficients (not used).
template <int F>
__global__ void __launch_bounds__(kCells * kSlices) linreg_logistic_kernel(const double* __restrict__ X, const double* __restrict__ y,
                                                                           int64_t ld, int64_t T, int64_t C, double thresh,
                                                                           const int32_t* __restrict__ status,
                                                                           const int32_t* __restrict__ thresh_off,
                                                                           double* __restrict__ logit /* [F+1][C] */) {
    constexpr int N = F + 1;
    __shared__ double part[kSlices * kCells];
    __shared__ double th_s[N * kCells], d_s[N * kCells], f_s[kCells], step_s[kCells];
    __shared__ int done_s[kCells];
    const int cx = threadIdx.x % kCells, ty = threadIdx.x / kCells;
    const int64_t c = (int64_t)blockIdx.x * kCells + cx;
    const bool live = c < C && status[c < C ? c : 0] == 0 && thresh_off[c < C ? c : 0] == 0;
    auto objective = [&](const double* t) {  // this thread's share
        double f = 0.0;
        if (live)
            for (int64_t tt = ty; tt < T; tt += kSlices) {
                double z = t[F];
#pragma unroll
                for (int a = 0; a < F; ++a) z += t[a] * X[(tt * F + a) * ld + c];
                f += sdlsq::softplus(z) - (y[tt * ld + c] > thresh ? z : 0.0);
            }
        return f;
    };
    if (ty == 0) {
#pragma unroll
        for (int a = 0; a < N; ++a) th_s[a * kCells + cx] = 0.0;
        done_s[cx] = live ? 0 : 1;
    }
    __syncthreads();
    double th[N];
#pragma unroll
    for (int a = 0; a < N; ++a) th[a] = 0.0;
    double fcur = slice_sum(objective(th), part, cx, ty);  // (penalty is 0 at the origin)
    for (int it = 0; it < 60; ++it) {
        if (__syncthreads_and(done_s[cx])) break;
        double g[N], H[N][N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            g[a] = 0.0;
#pragma unroll
            for (int b = 0; b < N; ++b) H[a][b] = 0.0;
        }
        if (live && !done_s[cx])
            for (int64_t tt = ty; tt < T; tt += kSlices) {
                double xa[N];
                double z = th[F];
#pragma unroll
                for (int a = 0; a < F; ++a) {
                    xa[a] = X[(tt * F + a) * ld + c];
                    z += th[a] * xa[a];
                }
                xa[F] = 1.0;
                const double sg = sdlsq::sigmoid(z), r = sg - (y[tt * ld + c] > thresh ? 1.0 : 0.0), w = sg * (1.0 - sg);
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    g[a] += r * xa[a];
#pragma unroll
                    for (int b = 0; b <= a; ++b) H[a][b] += w * xa[a] * xa[b];
                }
            }
        double gt[N], Ht[kMaxF + 1][kMaxF + 1];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            gt[a] = slice_sum(g[a], part, cx, ty) + (a < F ? th[a] : 0.0);
#pragma unroll
            for (int b = 0; b <= a; ++b) Ht[a][b] = slice_sum(H[a][b], part, cx, ty) + ((a == b && a < F) ? 1.0 : 0.0);
        }
        if (ty == 0 && !done_s[cx]) {
            double gmax = 0.0;
#pragma unroll
            for (int a = 0; a < N; ++a) gmax = fmax(gmax, fabs(gt[a]));
            double d[kMaxF + 1];
            bool stop = gmax <= 1e-12 * (double)T;
            if (!stop) {
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    Ht[a][a] += 1e-12;
                    gt[a] = -gt[a];
                }
                stop = !sdlsq::chol_solve(N, Ht, gt, d);
            }
            if (stop) {
                done_s[cx] = 1;
            } else {
#pragma unroll
                for (int a = 0; a < N; ++a) d_s[a * kCells + cx] = d[a];
                step_s[cx] = 1.0;
            }
        }
        __syncthreads();
        // line search: halve the step until the objective does not increase (at most 40 halvings)
        for (int ls = 0; ls < 40; ++ls) {
            const bool searching = !done_s[cx] && step_s[cx] > 0.0;  // (a negative step marks an accepted one)
            double trial[N];
            double pen = 0.0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
                trial[a] = th[a] + (searching ? step_s[cx] * d_s[a * kCells + cx] : 0.0);
                if (a < F) pen += 0.5 * trial[a] * trial[a];
            }
            const double fn = slice_sum(searching ? objective(trial) : 0.0, part, cx, ty) + pen;
            __syncthreads();
            if (ty == 0 && searching) {
                if (fn <= fcur || step_s[cx] < 1e-10) {
                    f_s[cx] = fn;
#pragma unroll
                    for (int a = 0; a < N; ++a) th_s[a * kCells + cx] = trial[a];
                    step_s[cx] = -1.0;  // accepted
                } else {
                    step_s[cx] *= 0.5;
                }
            }
            __syncthreads();
            const bool pending = !done_s[cx] && step_s[cx] > 0.0;
            if (!__syncthreads_or(pending)) break;
        }
#pragma unroll
        for (int a = 0; a < N; ++a) th[a] = th_s[a * kCells + cx];
        if (!done_s[cx] && step_s[cx] < 0.0) fcur = f_s[cx];
        __syncthreads();
    }
    if (ty == 0 && c < C) {
#pragma unroll
        for (int a = 0; a < N; ++a) logit[(int64_t)a * C + c] = th[a];
    }
}

template <int F>
__global__ void __launch_bounds__(kCells * kSlices) linreg_predict_kernel(const double* __restrict__ Xq, int64_t ld, int64_t Tq, int64_t C,
                                                                          const double* __restrict__ coef, const double* __restrict__ icpt_all,
                                                                          const double* __restrict__ rmse_all,
                                                                          const double* __restrict__ logit /* [F+1][C] or null */,
                                                                          const int32_t* __restrict__ thresh_off,
                                                                          const int32_t* __restrict__ fit_status, int32_t* __restrict__ status,
                                                                          double* __restrict__ out, int64_t ld_out) {
    const int cx = threadIdx.x % kCells, ty = threadIdx.x / kCells;
    const int64_t c = (int64_t)blockIdx.x * kCells + cx;
    if (c >= C) return;
    const bool active = fit_status[c] == 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double cf[F];
#pragma unroll
    for (int f = 0; f < F; ++f) cf[f] = coef[(int64_t)f * C + c];
    const double icpt = icpt_all[c], rmse = rmse_all[c];
    const bool with_prob = logit != nullptr && thresh_off[c] == 0;  // gard.py:455-459
    double lw[F + 1];
#pragma unroll
    for (int a = 0; a <= F; ++a) lw[a] = with_prob ? logit[(int64_t)a * C + c] : 0.0;
    bool bad = false;
    const int64_t t0 = (int64_t)blockIdx.y * kSlices * 16;
    for (int64_t t = t0 + ty; t < t0 + kSlices * 16 && t < Tq; t += kSlices) {
        double p = icpt, z = lw[F];
        bool fin = true;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const double v = Xq[(t * F + f) * ld + c];
            fin = fin && lr_finite(v);
            p += v * cf[f];
            z += v * lw[f];
        }
        bad |= !fin;
        const bool okq = active && fin;
        out[(t * 3 + 0) * ld_out + c] = okq ? p : nan;      // gard.py:465
        out[(t * 3 + 1) * ld_out + c] = okq ? (with_prob ? sdlsq::sigmoid(z) : 1.0) : nan;  // gard.py:455-459: predict_proba(X)[:, 1]
        out[(t * 3 + 2) * ld_out + c] = okq ? rmse : nan;   // gard.py:461-463
    }
    if (active && bad) atomicOr(&status[c], SDI_NONFINITE);
}

__global__ void __launch_bounds__(256) linreg_status_public_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int64_t C,
                                                                   int32_t* __restrict__ outp) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const int32_t bits = a[c] | (b ? b[c] : 0);
        outp[c] = sd_public_status(bits);
    }
}

template <int F>
int launch_fit(sd_ctx* ctx, const double* X, const double* y, int64_t ld, sd_linreg_state* st) {
    const dim3 grid((unsigned)((st->C + kCells - 1) / kCells));
    SD_LAUNCH(ctx, "linreg_fit_kernel", linreg_fit_kernel<F>, grid, dim3(kCells * kSlices), 0, X, y, ld, st->T, st->C, st->has_thresh,
              st->thresh, st->coef, st->intercept, st->rmse, st->status, st->thresh_off);
    if (st->has_thresh)
        SD_LAUNCH(ctx, "linreg_logistic_kernel", linreg_logistic_kernel<F>, grid, dim3(kCells * kSlices), 0, X, y, ld, st->T, st->C,
                  st->thresh, (const int32_t*)st->status, (const int32_t*)st->thresh_off, st->logit);
    return SD_OK;
}

template <int F>
int launch_predict(sd_ctx* ctx, const sd_linreg_state* st, const double* Xq, int64_t ld, int64_t Tq, int32_t* status_p, double* out,
                   int64_t ld_out) {
    const dim3 grid((unsigned)((st->C + kCells - 1) / kCells), (unsigned)((Tq + kSlices * 16 - 1) / (kSlices * 16)));
    SD_LAUNCH(ctx, "linreg_predict_kernel", linreg_predict_kernel<F>, grid, dim3(kCells * kSlices), 0, Xq, ld, Tq, st->C,
              (const double*)st->coef, (const double*)st->intercept, (const double*)st->rmse, (const double*)st->logit,
              (const int32_t*)st->thresh_off, (const int32_t*)st->status, status_p, out, ld_out);
    return SD_OK;
}

#define LINREG_DISPATCH_F(F, fn, ...)                      \
    switch (F) {                                           \
        case 1: SD_TRY(fn<1>(__VA_ARGS__)); break;         \
        case 2: SD_TRY(fn<2>(__VA_ARGS__)); break;         \
        case 3: SD_TRY(fn<3>(__VA_ARGS__)); break;         \
        case 4: SD_TRY(fn<4>(__VA_ARGS__)); break;         \
        case 5: SD_TRY(fn<5>(__VA_ARGS__)); break;         \
        case 6: SD_TRY(fn<6>(__VA_ARGS__)); break;         \
        case 7: SD_TRY(fn<7>(__VA_ARGS__)); break;         \
        default: SD_TRY(fn<8>(__VA_ARGS__)); break;        \
    }

}  // namespace

extern "C" {

int sd_linreg_state_destroy(sd_linreg_state* st) {
    if (!st) return SD_OK;
    if (st->ctx) {
        (void)hipSetDevice(st->ctx->device);
        (void)hipStreamSynchronize(st->ctx->stream);
    }
    sd_pool_release(st->ctx, st->coef);
    sd_pool_release(st->ctx, st->intercept);
    sd_pool_release(st->ctx, st->rmse);
    sd_pool_release(st->ctx, st->status);
    if (st->logit) sd_pool_release(st->ctx, st->logit);
    if (st->thresh_off) sd_pool_release(st->ctx, st->thresh_off);
    delete st;
    return SD_OK;
}

int sd_linreg_state_info(const sd_linreg_state* st, int64_t* T, int* F, int64_t* C) {
    SD_CHECK_ARG(st, "state is NULL");
    if (T) *T = st->T;
    if (F) *F = st->F;
    if (C) *C = st->C;
    return SD_OK;
}

int sd_linreg_state_export(const sd_linreg_state* st, double* coef, double* intercept, double* fit_error, double* logistic,
                           int32_t* thresh_dropped, int32_t* cell_status) {
    SD_CHECK_ARG(st, "state is NULL");
    sd_ctx* ctx = st->ctx;
    SD_HIP(hipSetDevice(ctx->device));
    if (coef) SD_HIP(hipMemcpyAsync(coef, st->coef, sizeof(double) * (size_t)st->F * st->C, hipMemcpyDeviceToHost, ctx->stream));
    if (intercept) SD_HIP(hipMemcpyAsync(intercept, st->intercept, sizeof(double) * st->C, hipMemcpyDeviceToHost, ctx->stream));
    if (fit_error) SD_HIP(hipMemcpyAsync(fit_error, st->rmse, sizeof(double) * st->C, hipMemcpyDeviceToHost, ctx->stream));
    if (logistic && st->logit)
        SD_HIP(hipMemcpyAsync(logistic, st->logit, sizeof(double) * (size_t)(st->F + 1) * st->C, hipMemcpyDeviceToHost, ctx->stream));
    if (thresh_dropped && st->thresh_off)
        SD_HIP(hipMemcpyAsync(thresh_dropped, st->thresh_off, sizeof(int32_t) * st->C, hipMemcpyDeviceToHost, ctx->stream));
    if (cell_status) {
        std::vector<int32_t> bits(st->C);
        SD_HIP(hipMemcpyAsync(bits.data(), st->status, sizeof(int32_t) * st->C, hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        for (int64_t c = 0; c < st->C; ++c) cell_status[c] = sd_public_status(bits[c]);
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

static int alloc_linreg(sd_ctx* ctx, sd_linreg_state* st) {
    const int F = st->F;
    const int64_t C = st->C;
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->coef, sizeof(double) * (size_t)F * C));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->intercept, sizeof(double) * C));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->rmse, sizeof(double) * C));
    SD_HIP(sd_pool_malloc(ctx, (void**)&st->status, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(st->status, 0, sizeof(int32_t) * C, ctx->stream));
    if (st->has_thresh) {
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->logit, sizeof(double) * (size_t)(F + 1) * C));
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->thresh_off, sizeof(int32_t) * C));
        SD_HIP(hipMemsetAsync(st->logit, 0, sizeof(double) * (size_t)(F + 1) * C, ctx->stream));
        SD_HIP(hipMemsetAsync(st->thresh_off, 0, sizeof(int32_t) * C, ctx->stream));
    }
    return SD_OK;
}

// fitted numbers -> device state (pickling, checkpoint / resume): logistic == NULL for a model without a threshold
int sd_linreg_state_import(sd_ctx* ctx, int64_t T, int F, int64_t C, const double* coef, const double* intercept, const double* fit_error,
                           const double* logistic, const int32_t* thresh_dropped, const int32_t* cell_status, sd_linreg_state** out) {
    SD_CHECK_ARG(ctx && coef && intercept && fit_error && out, "sd_linreg_state_import: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && F >= 1 && F <= kMaxF, "sd_linreg_state_import: bad sizes");
    SD_CHECK_ARG((logistic == nullptr) == (thresh_dropped == nullptr), "sd_linreg_state_import: logistic and thresh_dropped go together");
    *out = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    sd_linreg_state* st = new sd_linreg_state();
    st->ctx = ctx; st->T = T; st->F = F; st->C = C;
    st->has_thresh = logistic != nullptr;
    std::vector<int32_t> bits(C, 0);
    if (cell_status)
        for (int64_t c = 0; c < C; ++c) bits[c] = sd_internal_status(cell_status[c]);
    auto body = [&]() -> int {
        SD_TRY(alloc_linreg(ctx, st));
        SD_HIP(hipMemcpyAsync(st->coef, coef, sizeof(double) * (size_t)F * C, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipMemcpyAsync(st->intercept, intercept, sizeof(double) * C, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipMemcpyAsync(st->rmse, fit_error, sizeof(double) * C, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(hipMemcpyAsync(st->status, bits.data(), sizeof(int32_t) * C, hipMemcpyHostToDevice, ctx->stream));
        if (logistic) {
            SD_HIP(hipMemcpyAsync(st->logit, logistic, sizeof(double) * (size_t)(F + 1) * C, hipMemcpyHostToDevice, ctx->stream));
            SD_HIP(hipMemcpyAsync(st->thresh_off, thresh_dropped, sizeof(int32_t) * C, hipMemcpyHostToDevice, ctx->stream));
        }
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    };
    const int rc = body();
    if (rc != SD_OK) {
        sd_linreg_state_destroy(st);
        return rc;
    }
    *out = st;
    return SD_OK;
}

int sd_linreg_fit_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                      int has_thresh, double thresh, sd_linreg_state** out) {
    SD_CHECK_ARG(ctx && X_dev && y_dev && out, "sd_linreg_fit: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && ld >= C, "sd_linreg_fit: bad sizes");
    SD_CHECK_ARG(F >= 1 && F <= kMaxF, "sd_linreg_fit: F=%d outside [1,%d]", F, kMaxF);
    *out = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    sd_linreg_state* st = new sd_linreg_state();
    st->ctx = ctx;
    st->T = T;
    st->F = F;
    st->C = C;
    st->has_thresh = has_thresh ? 1 : 0;
    st->thresh = thresh;
    auto body = [&]() -> int {
        SD_TRY(alloc_linreg(ctx, st));
        LINREG_DISPATCH_F(F, launch_fit, ctx, X_dev, y_dev, ld, st);
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    };
    const int rc = body();
    if (rc != SD_OK) {
        sd_linreg_state_destroy(st);
        return rc;
    }
    *out = st;
    return SD_OK;
}

int sd_linreg_fit(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, int has_thresh, double thresh,
                  sd_linreg_state** out) {
    SD_CHECK_ARG(ctx && X && y && out, "sd_linreg_fit: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && F >= 1, "sd_linreg_fit: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy;
    SD_HIP(dX.alloc(ctx, sizeof(double) * (size_t)T * F * C));
    SD_HIP(dy.alloc(ctx, sizeof(double) * (size_t)T * C));
    SD_TRY(sd_copy_h2d(ctx, dX.p, X, sizeof(double) * (size_t)T * F * C));
    SD_TRY(sd_copy_h2d(ctx, dy.p, y, sizeof(double) * (size_t)T * C));
    return sd_linreg_fit_dev(ctx, dX.as<double>(), dy.as<double>(), C, T, F, C, has_thresh, thresh, out);
}

int sd_linreg_predict_dev(sd_ctx* ctx, const sd_linreg_state* st, const double* Xq_dev, int64_t ld, int64_t Tq, double* out_dev,
                          int64_t ld_out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xq_dev && out_dev, "sd_linreg_predict: NULL argument");
    SD_CHECK_ARG(Tq > 0 && ld >= st->C && ld_out >= st->C, "sd_linreg_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C;
    sd_scratch status_p, status_pub;
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    LINREG_DISPATCH_F(st->F, launch_predict, ctx, st, Xq_dev, ld, Tq, status_p.as<int32_t>(), out_dev, ld_out);
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "linreg_status_public_kernel", linreg_status_public_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                  (const int32_t*)st->status, (const int32_t*)status_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_linreg_predict(sd_ctx* ctx, const sd_linreg_state* st, const double* Xq, int64_t Tq, double* out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xq && out, "sd_linreg_predict: NULL argument");
    SD_CHECK_ARG(Tq > 0, "sd_linreg_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dout;
    const size_t in_bytes = sizeof(double) * (size_t)Tq * st->F * st->C, out_bytes = sizeof(double) * (size_t)Tq * 3 * st->C;
    SD_HIP(dX.alloc(ctx, in_bytes));
    SD_HIP(dout.alloc(ctx, out_bytes));
    SD_TRY(sd_copy_h2d(ctx, dX.p, Xq, in_bytes));
    SD_TRY(sd_linreg_predict_dev(ctx, st, dX.as<double>(), st->C, Tq, dout.as<double>(), st->C, cell_status));
    SD_TRY(sd_copy_d2h(ctx, out, dout.p, out_bytes));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

}  // extern "C"
