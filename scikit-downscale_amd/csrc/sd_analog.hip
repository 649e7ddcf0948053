// GARD analog models (PureAnalog / AnalogRegression), batched over the cell axis.
//
// Reference (file:line under skdownscale/pointwise_models/gard.py): AnalogBase.fit 58-87 (KDTree),
// PureAnalog.predict 273-364, AnalogRegression.predict/_predict_one_step 152-224 (thresh=None).
// KDTree.query is restated as: k training rows with the smallest reduced distance
// rdist = sum_f (q_f - x_f)^2 (accumulated f = 0..F-1, no FMA), ascending by (rdist, index).
//
// fit   : mask / finite check, cell-major copies Xc[C][F][T], yc[C][T] (tiled LDS transpose); for F == 1
//         additionally the sorted view of a cell: xs[C][T] (values by (x, index)), xi[C][T] (their training
//         indices), yx[C][T] (y in that order) and pq[C][T+1][2] (prefix sums of the centred yx and its
//         squares) -- analog_sort2_kernel: workgroup merge sort (sd_sortnet.h); rx[C][T+1], the cross term of
//         the one-feature regression, is added by analog_rx_kernel on the first AnalogRegression call.  For
//         F > 1 a copy of the training points sorted by feature 0 (ps, indices xi).
// predict, F == 1 (one persistent workgroup per cell, queries and outputs through cell-major staging):
//   analog_f1_mean3_kernel  mean_analogs without a threshold: window search over xs in LDS, then the two prefix-sum
//                           components staged through the same LDS array (three generations per cell);
//   analog_f1_mean_kernel   a single analog, AnalogRegression, weighted / thresholded kinds: window search over xs
//                           in LDS, prefix sums or the window of yx read from memory (single pass);
//   analog_f1_window_kernel the other PureAnalog kinds: k-NN window over xs, statistics from yx, both
//                           LDS-resident per value range;
//   analog_f1_predict_kernel / f1_walk_query  exact (rdist, index)-ordered two-pointer walk: 'sample_analogs',
//                           neighbour outputs, and any query whose window has a tie on its boundary.
// predict, F > 1: analog_slab_topk_kernel (sd_analog_topk.h; k <= 30, F <= 6: one wave per 64 queries sorted by feature 0, only the
//   reachable slab of the feature-0 sorted copy is scanned, the 64 x 64 mask of a chunk from the matrix cores, candidate lists pruned
//   by a register sorting network); analog_slab_predict_kernel (the same scan with scalar-loaded points and a top-k heap in LDS:
//   larger k / F and the batches the first hands back);
//   analog_bf2_predict_kernel (same scanner over the whole set in index order); analog_bf_predict_kernel
//   (LDS-staged tiles, lists in global scratch) for k > 208.
// fit + predict in one call (sd_analog_fit_predict*): analog_f1_fused_kernel -- the workgroup that merged a cell's sorted runs
//   answers its queries (the mean3 phases) without a fitted state in memory; cells it hands back and every other configuration
//   take fit -> predict internally.
// Epilogues: PureAnalog statistics (gard.py:303-346), per-query least squares (gard.py:194-224).
// Layout of the sources: kernels in sd_analog_fit.h (fit), sd_analog_epilogue.h, sd_analog_f1.h (F == 1 predict, fused kernel),
// sd_analog_fn.h, sd_analog_topk.h (F > 1 predict), included below; host code and the C entry points here.
#include <algorithm>
#include <cstdlib>

#include "sd_internal.h"
#include "sd_lsq.h"
#include "sd_sortnet.h"
#include "sd_wave.h"
#include "sd_wsort.h"

namespace {

constexpr int kMaxF = sdlsq::kMaxF;

__device__ __forceinline__ bool sd_finite(double v) { return (__double_as_longlong(v) & 0x7ff0000000000000ll) != 0x7ff0000000000000ll; }

// XCD-aware persistent mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of
// cells and let its workgroups take adjacent cells at the same time, so 8-byte column reads of
// neighbouring cells merge into full lines in that XCD's L2.
__device__ __forceinline__ int64_t first_cell(int64_t C, int64_t* step, int64_t* end) {
    const int nb = gridDim.x, b = blockIdx.x;
    if (nb % 8 != 0) {
        *step = nb;
        *end = C;
        return b;
    }
    const int64_t cx = (C + 7) / 8;
    const int x = b % 8, j = b / 8;
    *step = nb / 8;
    *end = (x + 1) * cx < C ? (x + 1) * cx : C;
    return x * cx + j;
}

#include "sd_analog_fit.h"
#include "sd_analog_runs.h"
#include "sd_analog_epilogue.h"
#include "sd_analog_f1.h"
#include "sd_analog_fn.h"
#include "sd_analog_topk.h"

__global__ void __launch_bounds__(256) analog_status_public_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                                   int64_t C, int32_t* __restrict__ outp) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const int32_t bits = a[c] | (b ? b[c] : 0);
        outp[c] = (bits & SDI_MASKED) ? SD_CELL_MASKED : (bits & SDI_NONFINITE) ? SD_CELL_NONFINITE : (bits & SDI_ONE_CLASS) ? SD_CELL_ONE_CLASS : SD_CELL_OK;
    }
}

int launch_prefix_sums(sd_ctx* ctx, const double* yx, int64_t T, int64_t C, double* pq, double* ybar, int keep_ybar) {
    const int nbp = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * 2);
    const size_t lds = sizeof(double) * (size_t)(T + 1);
    if (lds + 512 <= ctx->lds_max) {
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_prefix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SD_LAUNCH(ctx, "analog_prefix_kernel", analog_prefix_kernel, dim3(nbp), dim3(1024), lds, yx, T, C, pq, ybar, keep_ybar);
    } else {
        SD_LAUNCH(ctx, "analog_prefix_kernel", analog_prefix_direct_kernel, dim3(nbp), dim3(1024), 0, yx, T, C, pq, ybar, keep_ybar);
    }
    return SD_OK;
}

// exclusive prefix sums of the centred analog values (analog_prefix_kernel), built when a kernel that reads them from
// memory first runs on a state (calls on a context are serialised)
int ensure_prefix_sums(sd_ctx* ctx, const sd_analog_state* st) {
    if (st->pq != nullptr) return SD_OK;
    sd_analog_state* ms = const_cast<sd_analog_state*>(st);
    SD_HIP(sd_pool_malloc(ctx, (void**)&ms->pq, sizeof(double) * 2 * (size_t)(st->T + 1) * st->C));
    return launch_prefix_sums(ctx, (const double*)st->yx, st->T, st->C, ms->pq, ms->ybar, 1);
}

int predict_common(int mode, sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t ld, int64_t Tq, int k,
                   int kind, int has_thresh, double thresh, const int32_t* sample_dev, int64_t ld_s, double* out,
                   int64_t ld_out, int64_t* inds, double* dist, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xq && out, "sd_analog_predict: NULL argument");
    SD_CHECK_ARG(Tq > 0 && ld >= st->C && ld_out >= st->C, "sd_analog_predict: bad sizes");
    SD_CHECK_ARG(k >= 1 && k <= st->T, "sd_analog_predict: k=%d must be in [1, T=%lld]", k, (long long)st->T);
    SD_CHECK_ARG(mode == 1 || (kind >= SD_ANALOG_BEST && kind <= SD_ANALOG_MEAN), "sd_analog_predict: unknown kind %d", kind);
    SD_CHECK_ARG(!(mode == 0 && kind == SD_ANALOG_SAMPLE) || sample_dev, "sd_analog_predict: sample_analogs needs sample_inds");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C, T = st->T;
    const int F = st->F;
    // PureAnalog.predict with a single analog is 'best_analog' whatever the configured kind (gard.py:291-296: n_analogs == 1);
    // the entry point sees k only, so a one-sample training set (k_ = 1 with n_analogs > 1) is treated the same way
    if (mode == 0 && k == 1) kind = SD_ANALOG_BEST;
    PredictArgs pa;
    pa.k = k;
    pa.kind = kind;
    pa.has_thresh = has_thresh;
    pa.thresh = thresh;
    pa.sample = sample_dev;
    pa.ld_s = ld_s;
    pa.out = out;
    pa.ld_out = ld_out;
    pa.inds = inds;
    pa.dist = dist;
    pa.oc_Tq = 0;
    sd_scratch status_p, sc_d, sc_i, status_pub;
    SD_HIP(status_p.alloc(ctx, sizeof(int32_t) * C));
    SD_HIP(hipMemsetAsync(status_p.p, 0, sizeof(int32_t) * C, ctx->stream));
    pa.one_class = status_p.as<int32_t>();
    const bool f1 = st->xs != nullptr;
    const int nthr = f1 ? 1024 : kBfThreads;
    int nb = ctx->cu_count * (f1 ? 1 : 4);
    nb = (nb / 8) * 8;
    if (nb < 8) nb = 8;
    if ((int64_t)nb > ((C + 7) / 8) * 8) nb = (int)(((C + 7) / 8) * 8);
    SD_HIP(sc_d.alloc(ctx, sizeof(double) * (size_t)nb * k * nthr));
    SD_HIP(sc_i.alloc(ctx, sizeof(int32_t) * (size_t)nb * k * nthr));
    // (a thresholded regression needs the analogs themselves: logistic fit and subset OLS, gard.py:201-219)
    const bool window = f1 && (mode == 1 || kind != SD_ANALOG_SAMPLE) && !inds && !dist && st->yx != nullptr &&
                        !(mode == 1 && has_thresh) && sd_dev_env("SD_ANALOG_WALK") == nullptr;
    if (window) {
        // fewest value ranges such that xs and yx of a range (+ k entries of margin each side) fit the LDS
        int npass = 1;
        size_t lds = 0;
        for (;; ++npass) {
            const size_t cap = (size_t)((T + npass - 1) / npass) + 2 * (size_t)k + 1;
            lds = sizeof(double) * (2 * cap + 1);
            if (lds <= ctx->lds_max || npass >= 64) break;
        }
        SD_CHECK_ARG(lds <= ctx->lds_max, "sd_analog_predict: k=%d too large for the windowed path", k);
        // queries and outputs go through cell-major copies: the column accesses of a cell would be 8-byte
        // requests 8*ld bytes apart (one 64-byte sector each); the tiled transposes stream at HBM speed.
        // Cells are processed in chunks so that the staging buffers stay small (and cache-resident).
        const int64_t chunk = 16384;
        const int64_t cc_max = C < chunk ? C : chunk;
        sd_scratch qc, oc, qtags;
        SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
        SD_HIP(oc.alloc(ctx, sizeof(double) * (size_t)Tq * 3 * cc_max));
        // round 6: the queries of a cell in value order inside runs of 1 024 time steps (sd_analog_runs.h): the kernels below index
        // queries and results by position, the two staging kernels translate between positions and times
        bool runs_q = query_runs_apply(Tq);
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_window_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // single pass with only xs in LDS (statistics from the prefix sums, or the window of yx read from memory)
        const size_t lds_mean = sizeof(double) * (size_t)(T + 1);
        const bool mean_only = (mode == 1 ? k >= 3 : (kind == SD_ANALOG_MEAN || kind == SD_ANALOG_WEIGHT || k == 1)) && st->ybar != nullptr &&
                               lds_mean <= ctx->lds_max && sd_dev_env("SD_ANALOG_NOPREFIX") == nullptr;
        const bool phases = mean_only && mode == 0 && ((kind == SD_ANALOG_MEAN && !has_thresh) || k == 1) && T <= 1024 * 20 &&
                            sd_dev_env("SD_ANALOG_NOPHASES") == nullptr;
        // without a threshold the probability column is 1 wherever the prediction is not NaN (gard.py:346; AnalogRegression: gard.py:211-212):
        // the single-pass kernels do not write it either, the staging transpose derives it from the predictions
        const int skip_prob = (phases || mean_only) && !has_thresh ? 1 : 0;
        // Value-ordered runs pay where a query reads its window of analog values from memory (weights, thresholds, the regression):
        // neighbouring lanes then read overlapping lines.  The three-generation kernel reads nothing per query but LDS words, and
        // its search is bound by its spilled registers, not by bank conflicts: measured equal with sorted queries (33.7 against
        // 33.0 ms per 100 000 cells), while the run staging costs 6 ms more than the plain transposes -- it keeps the time order.
        runs_q = runs_q && !(phases && sd_dev_env("SD_ANALOG_RUNS_ALWAYS") == nullptr);
        if (runs_q) SD_HIP(qtags.alloc(ctx, sizeof(unsigned short) * (size_t)Tq * cc_max));
        const size_t lds_mean3 = lds_mean;
        if (mean_only) {
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean3_kernel<8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean3));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean3_kernel<16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean3));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_mean3_kernel<20>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mean3));
        }
        // AnalogRegression with a short window sums it directly (reg_batch: no prefix arrays to build or to read); the default
        // n_analogs = 200 keeps the prefix differences (two 16-byte loads instead of 200 values per query)
        const bool reg_direct = mean_only && mode == 1 && k <= kRegDirectK && sd_dev_env("SD_ANALOG_REG_PREFIX") == nullptr;
        // (the prefix sums serve the regression and the plain mean; weights and thresholds read the analog values themselves)
        if (mean_only && !phases && !reg_direct && (mode == 1 || (kind == SD_ANALOG_MEAN && !has_thresh && k > 1))) SD_TRY(ensure_prefix_sums(ctx, st));
        if (mean_only && mode == 1 && !reg_direct && st->rx == nullptr) {
            // first regression on this state: the cross-term prefix sums (calls on a context are serialised)
            sd_analog_state* ms = const_cast<sd_analog_state*>(st);
            SD_HIP(sd_pool_malloc(ctx, (void**)&ms->rx, sizeof(double) * (size_t)(T + 1) * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&ms->xbar, sizeof(double) * C));
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_rx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_mean));
            SD_LAUNCH(ctx, "analog_rx_kernel", analog_rx_kernel, dim3((unsigned)std::min<int64_t>(C, (int64_t)ctx->cu_count * 2)),
                      dim3(1024), lds_mean, (const double*)st->xs, (const double*)st->yx, (const double*)st->ybar, T, C, ms->rx,
                      ms->xbar);
        }
        for (int64_t cb = 0; cb < C; cb += chunk) {
            const int64_t cc = C - cb < chunk ? C - cb : chunk;
            dim3 tgrid((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32));
            if (runs_q) {
                SD_TRY(launch_query_runs(ctx, Xq + cb, ld, Tq, cc, qc.as<double>(), qtags.as<unsigned short>(), status_p.as<int32_t>() + cb));
            } else {
                SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, tgrid, dim3(256), 0, Xq + cb, ld, Tq, 1, 0, cc,
                          qc.as<double>(), status_p.as<int32_t>() + cb, 0);
            }
            PredictArgs pw = pa;
            pw.out = oc.as<double>();
            pw.oc_Tq = Tq;
            pw.skip_prob = skip_prob;
            int nbc = nb;
            if ((int64_t)nbc > ((cc + 7) / 8) * 8) nbc = (int)(((cc + 7) / 8) * 8);
            // workgroups per cell in the single-pass kernel (see its qsplit): only when every XCD still gets whole groups
            const char* eqs = sd_dev_env("SD_ANALOG_QSPLIT");
            int qs = eqs ? atoi(eqs) : (mode == 1 && !reg_direct ? 2 : 1);  // measured (ms per 16 384 cells), 1/2/4/8: regression 10.8/8.5/8.7/11.0, mean 5.5/5.7/6.4/8.3
            if (qs < 1 || nbc % (8 * qs) != 0 || cc < (int64_t)nbc || Tq < 4096) qs = 1;
            if (phases) {
                const int per = (int)((T + nthr - 1) / nthr);
                long long* trace_dev = nullptr;
                sd_scratch trace_buf;
                if (sd_dev_env("SD_M3_TRACE") != nullptr) {  // development library: phase clocks of the first cells of block 0
                    SD_HIP(trace_buf.alloc(ctx, sizeof(long long) * 128));
                    SD_HIP(hipMemsetAsync(trace_buf.p, 0, sizeof(long long) * 128, ctx->stream));
                    trace_dev = trace_buf.as<long long>();
                }
#define SD_MEAN3(PER)                                                                                                                  \
    SD_LAUNCH(ctx, "analog_f1_mean3_kernel", analog_f1_mean3_kernel<PER>, dim3(nbc), dim3(nthr), lds_mean3, (const double*)qc.p, Tq, T, \
              cc, (const double*)st->xs + cb * T, (const int32_t*)st->xi + cb * T, (const double*)st->ybar + cb,                      \
              (const double*)st->yx + cb * T, (const double*)st->X + cb * T, (const double*)st->y + cb * T,                           \
              (const int32_t*)st->status + cb, status_p.as<int32_t>() + cb, sc_d.as<double>(), sc_i.as<int32_t>(), pw, skip_prob, \
              trace_dev)
                if (per <= 8) SD_MEAN3(8);
                else if (per <= 16) SD_MEAN3(16);
                else SD_MEAN3(20);
#undef SD_MEAN3
                if (trace_dev != nullptr) {
                    long long h[128];
                    SD_HIP(hipMemcpyAsync(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
                    SD_HIP(hipStreamSynchronize(ctx->stream));
                    for (int r = 0; r < 8; ++r) {
                        fprintf(stderr, "mean3 trace cell %d:", r);
                        for (int j = 1; j <= 9; ++j) fprintf(stderr, " %lld", h[r * 16 + j] - h[r * 16 + j - 1]);
                        fprintf(stderr, "\n");
                    }
                }
            } else if (mean_only) {
                SD_LAUNCH(ctx, "analog_f1_mean_kernel", analog_f1_mean_kernel, dim3(nbc), dim3(nthr), lds_mean, mode,
                          (const double*)qc.p, Tq, T, cc, (const double*)st->xs + cb * T, (const int32_t*)st->xi + cb * T,
                          (const double*)st->pq + 2 * cb * (T + 1), (const double*)st->ybar + cb,
                          (const double*)st->rx + cb * (T + 1), (const double*)st->xbar + cb, (const double*)st->yx + cb * T,
                          (const double*)st->X + cb * T,
                          (const double*)st->y + cb * T, (const int32_t*)st->status + cb, status_p.as<int32_t>() + cb,
                          sc_d.as<double>(), sc_i.as<int32_t>(), pw, qs, reg_direct ? 1 : 0);
            } else {
                SD_LAUNCH(ctx, "analog_f1_window_kernel", analog_f1_window_kernel, dim3(nbc), dim3(nthr), lds, mode,
                          (const double*)qc.p, Tq, Tq, T, cc, npass, (const double*)st->xs + cb * T, (const int32_t*)st->xi + cb * T,
                          (const double*)st->yx + cb * T, (const double*)st->X + cb * T, (const double*)st->y + cb * T,
                          (const int32_t*)st->status + cb, status_p.as<int32_t>() + cb, sc_d.as<double>(), sc_i.as<int32_t>(), pw);
            }
            if (runs_q) {
                SD_TRY(launch_untranspose_runs(ctx, (const double*)oc.p, (const unsigned short*)qtags.p, Tq, cc, out + cb, ld_out, skip_prob));
            } else {
                SD_LAUNCH(ctx, "analog_untranspose_kernel", analog_untranspose_kernel,
                          dim3((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32), skip_prob ? 2 : 3), dim3(256), 0,
                          (const double*)oc.p, Tq, cc, out + cb, ld_out, skip_prob);
            }
        }
        SD_HIP(hipStreamSynchronize(ctx->stream));  // qc / oc go back to the block cache at scope exit
    } else if (f1) {
        const size_t lds = sizeof(double) * T;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_predict_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SD_LAUNCH(ctx, "analog_f1_predict_kernel", analog_f1_predict_kernel, dim3(nb), dim3(nthr), lds, mode, Xq, ld, Tq,
                  T, C, (const double*)st->xs, (const int32_t*)st->xi, (const double*)st->X, (const double*)st->y,
                  (const int32_t*)st->status, status_p.as<int32_t>(), sc_d.as<double>(), sc_i.as<int32_t>(), pa);
    } else if (F > 1 && st->ps != nullptr && bf2_lds_bytes(k, F, 2) <= ctx->lds_max && sort2_width(Tq, ctx->lds_max) != 0 &&
               sd_dev_env("SD_ANALOG_NOSLAB") == nullptr) {
        SD_TRY(predict_slab(ctx, mode, st, Xq, ld, Tq, status_p.as<int32_t>(), pa));
    } else if (bf2_lds_bytes(k, F, 4) <= ctx->lds_max && sd_dev_env("SD_ANALOG_BF1") == nullptr) {
        int32_t* sp = status_p.as<int32_t>();
        switch (F) {
            case 1: SD_TRY(launch_bf2<1>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 2: SD_TRY(launch_bf2<2>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 3: SD_TRY(launch_bf2<3>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 4: SD_TRY(launch_bf2<4>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 5: SD_TRY(launch_bf2<5>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 6: SD_TRY(launch_bf2<6>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            case 7: SD_TRY(launch_bf2<7>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
            default: SD_TRY(launch_bf2<8>(ctx, mode, st, Xq, ld, Tq, sp, pa)); break;
        }
    } else {
        const size_t lds = sizeof(double) * F * kBfChunk;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_bf_predict_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SD_LAUNCH(ctx, "analog_bf_predict_kernel", analog_bf_predict_kernel, dim3(nb), dim3(nthr), lds, mode, Xq, ld, Tq,
                  T, F, C, (const double*)st->X, (const double*)st->y, (const int32_t*)st->status,
                  status_p.as<int32_t>(), sc_d.as<double>(), sc_i.as<int32_t>(), pa);
    }
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "analog_status_public_kernel", analog_status_public_kernel, dim3((unsigned)((C + 255) / 256)),
                  dim3(256), 0, (const int32_t*)st->status, (const int32_t*)status_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int predict_host(int mode, sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int kind,
                 int has_thresh, double thresh, const int32_t* sample, double* out, int64_t* inds, double* dist,
                 int32_t* cell_status) {
    SD_CHECK_ARG(ctx && st && Xq && out, "sd_analog_predict: NULL argument");
    SD_CHECK_ARG(Tq > 0 && k >= 1, "sd_analog_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t C = st->C;
    sd_scratch dq, dout, dinds, ddist, dsamp;
    const size_t qb = sizeof(double) * (size_t)Tq * st->F * C, ob = sizeof(double) * (size_t)Tq * 3 * C;
    SD_HIP(dq.alloc(ctx, qb));
    SD_HIP(dout.alloc(ctx, ob));
    SD_TRY(sd_copy_h2d(ctx, dq.p, Xq, qb));
    if (inds) SD_HIP(dinds.alloc(ctx, sizeof(int64_t) * (size_t)Tq * k * C));
    if (dist) SD_HIP(ddist.alloc(ctx, sizeof(double) * (size_t)Tq * k * C));
    if (sample) {
        SD_HIP(dsamp.alloc(ctx, sizeof(int32_t) * (size_t)Tq * C));
        SD_TRY(sd_copy_h2d(ctx, dsamp.p, sample, sizeof(int32_t) * (size_t)Tq * C));
    }
    SD_TRY(predict_common(mode, ctx, st, dq.as<double>(), C, Tq, k, kind, has_thresh, thresh, dsamp.as<int32_t>(), C,
                          dout.as<double>(), C, dinds.as<int64_t>(), ddist.as<double>(), cell_status));
    SD_TRY(sd_copy_d2h(ctx, out, dout.p, ob));
    if (inds) SD_HIP(hipMemcpyAsync(inds, dinds.p, sizeof(int64_t) * (size_t)Tq * k * C, hipMemcpyDeviceToHost, ctx->stream));
    if (dist) SD_HIP(hipMemcpyAsync(dist, ddist.p, sizeof(double) * (size_t)Tq * k * C, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

// columns `list[0 .. nw)` of a [R, ld] field <-> a packed [R, nw] field (cells the fused kernel handed back)
__global__ void __launch_bounds__(256) analog_gather_cells_kernel(const double* __restrict__ src, int64_t ld, int64_t R,
                                                                  const int32_t* __restrict__ list, int64_t nw,
                                                                  double* __restrict__ dst) {
    const int64_t total = R * nw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nw, j = i - r * nw;
        dst[i] = src[r * ld + list[j]];
    }
}
__global__ void __launch_bounds__(256) analog_scatter_cells_kernel(const double* __restrict__ src, int64_t R,
                                                                   const int32_t* __restrict__ list, int64_t nw,
                                                                   double* __restrict__ dst, int64_t ld) {
    const int64_t total = R * nw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nw, j = i - r * nw;
        dst[r * ld + list[j]] = src[i];
    }
}

template <int K>
int launch_fused_k(sd_ctx* ctx, int nbc, size_t lds, const double* runs, int np, const int32_t* odd, const double* Xc, const double* yc,
                   const double* qc, int64_t Tq, int64_t T, int64_t cc, const int32_t* st_fit, int32_t* st_p, int32_t* worklist,
                   int32_t* work_count, int64_t cell0, const PredictArgs& pw, int skip_prob) {
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_f1_fused_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "analog_f1_fused_kernel", analog_f1_fused_kernel<K>, dim3(nbc), dim3(1024), lds, runs, np, odd, Xc, yc, qc, Tq, T, cc, st_fit,
              st_p, worklist, work_count, cell0, pw, skip_prob);
    return SD_OK;
}

// LDS of analog_f1_fused_kernel: keys (np + 1 doubles), co-ranks (1025 ints, padded), tags (T x 16 bit); + its static arrays
size_t fused_lds_bytes(int np, int64_t T) { return sizeof(double) * (size_t)(np + 1) + sizeof(int) * 1026 + ((sizeof(uint16_t) * (size_t)T + 15) & ~(size_t)15); }

// the split path on device fields: fit -> predict -> drop the state
int fit_predict_split(sd_ctx* ctx, const double* X, const double* y, int64_t ld, int64_t T, int F, int64_t C, const double* Xq,
                      int64_t ld_q, int64_t Tq, int k, int kind, int has_thresh, double thresh, double* out, int64_t ld_out,
                      int32_t* cell_status) {
    sd_analog_state* st = nullptr;
    SD_TRY(sd_analog_fit_dev(ctx, X, y, ld, T, F, C, &st));
    const int rc = predict_common(0, ctx, st, Xq, ld_q, Tq, k, kind, has_thresh, thresh, nullptr, 0, out, ld_out, nullptr, nullptr, cell_status);
    sd_analog_state_destroy(st);
    return rc;
}

int fit_predict_dev(sd_ctx* ctx, const double* X, const double* y, int64_t ld, int64_t T, int F, int64_t C, const double* Xq,
                    int64_t ld_q, int64_t Tq, int k, int kind, int has_thresh, double thresh, double* out, int64_t ld_out,
                    int32_t* cell_status) {
    SD_CHECK_ARG(ctx && X && y && Xq && out, "sd_analog_fit_predict: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && Tq > 0 && ld >= C && ld_q >= C && ld_out >= C, "sd_analog_fit_predict: bad sizes");
    SD_CHECK_ARG(F >= 1 && F <= kMaxF, "sd_analog_fit_predict: F=%d outside [1,%d]", F, kMaxF);
    SD_CHECK_ARG(k >= 1 && k <= T, "sd_analog_fit_predict: k=%d must be in [1, T=%lld]", k, (long long)T);
    SD_CHECK_ARG(kind >= SD_ANALOG_BEST && kind <= SD_ANALOG_MEAN && kind != SD_ANALOG_SAMPLE, "sd_analog_fit_predict: kind %d (sample_analogs needs the split calls)", kind);
    SD_HIP(hipSetDevice(ctx->device));
    if (k == 1) kind = SD_ANALOG_BEST;  // (as in predict_common: gard.py:291-296)
    const int K = F == 1 ? sort2_width(T, ctx->lds_max) : 0;
    const int64_t chunk_t = 64 * (int64_t)(K > 0 ? K : 1);
    const int np = (int)(((T + chunk_t - 1) / chunk_t) * chunk_t);
    const bool fused = K != 0 && tile_sort_applies(K, T, C, ctx->lds_max) && ((kind == SD_ANALOG_MEAN && !has_thresh) || k == 1) &&
                       Tq <= (int64_t)kPhQ * 1024 && fused_lds_bytes(np, T) + 512 <= ctx->lds_max && sd_dev_env("SD_ANALOG_NOFUSE") == nullptr;
    if (!fused) return fit_predict_split(ctx, X, y, ld, T, F, C, Xq, ld_q, Tq, k, kind, has_thresh, thresh, out, ld_out, cell_status);

    PredictArgs pa;
    pa.k = k; pa.kind = kind; pa.has_thresh = has_thresh; pa.thresh = thresh;
    pa.sample = nullptr; pa.ld_s = 0; pa.out = out; pa.ld_out = ld_out; pa.inds = nullptr; pa.dist = nullptr; pa.oc_Tq = 0;
    const int64_t chunk = 16384;
    const int64_t cc_max = C < chunk ? C : chunk;
    sd_scratch Xc, yc, runs, st_fit, st_p, odd, list, qc, oc, status_pub;
    SD_HIP(Xc.alloc(ctx, sizeof(double) * (size_t)T * C));
    SD_HIP(yc.alloc(ctx, sizeof(double) * (size_t)T * C));
    SD_HIP(runs.alloc(ctx, sizeof(double) * (size_t)np * C));
    SD_HIP(st_fit.alloc(ctx, sizeof(int32_t) * (size_t)C));
    SD_HIP(st_p.alloc(ctx, sizeof(int32_t) * (size_t)C));
    SD_HIP(odd.alloc(ctx, sizeof(int32_t) * (size_t)C));
    SD_HIP(list.alloc(ctx, sizeof(int32_t) * (size_t)(C + 1)));
    SD_HIP(qc.alloc(ctx, sizeof(double) * (size_t)Tq * cc_max));
    SD_HIP(oc.alloc(ctx, sizeof(double) * (size_t)Tq * 3 * cc_max));
    sd_scratch qtags;
    // (the fused kernel gains nothing from value-ordered queries -- see predict_common -- : development switch only)
    const bool runs_q = query_runs_apply(Tq) && sd_dev_env("SD_ANALOG_RUNS_ALWAYS") != nullptr;
    if (runs_q) SD_HIP(qtags.alloc(ctx, sizeof(unsigned short) * (size_t)Tq * cc_max));
    pa.one_class = st_p.as<int32_t>();
    int32_t* work_count = list.as<int32_t>();
    int32_t* worklist = work_count + 1;
    SD_HIP(hipMemsetAsync(st_fit.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
    SD_HIP(hipMemsetAsync(st_p.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
    SD_HIP(hipMemsetAsync(odd.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
    SD_HIP(hipMemsetAsync(work_count, 0, sizeof(int32_t), ctx->stream));
    SD_TRY(launch_tile_sort(ctx, K, X, y, ld, T, C, Xc.as<double>(), yc.as<double>(), runs.as<double>(), np, st_fit.as<int32_t>(), odd.as<int32_t>()));
    const int skip_prob = !has_thresh ? 1 : 0;
    const size_t lds = fused_lds_bytes(np, T);
    int nb = (ctx->cu_count / 8) * 8;
    if (nb < 8) nb = 8;
    if ((int64_t)nb > ((C + 7) / 8) * 8) nb = (int)(((C + 7) / 8) * 8);
    for (int64_t cb = 0; cb < C; cb += chunk) {
        const int64_t cc = C - cb < chunk ? C - cb : chunk;
        dim3 tgrid((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32));
        if (runs_q) {
            SD_TRY(launch_query_runs(ctx, Xq + cb, ld_q, Tq, cc, qc.as<double>(), qtags.as<unsigned short>(), st_p.as<int32_t>() + cb));
        } else {
            SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, tgrid, dim3(256), 0, Xq + cb, ld_q, Tq, 1, 0, cc, qc.as<double>(),
                      st_p.as<int32_t>() + cb, 0);
        }
        PredictArgs pw = pa;
        pw.out = oc.as<double>();
        pw.oc_Tq = Tq;
        int nbc = nb;
        if ((int64_t)nbc > ((cc + 7) / 8) * 8) nbc = (int)(((cc + 7) / 8) * 8);
        const double* r = runs.as<double>() + cb * (int64_t)np;
        const double* xc = Xc.as<double>() + cb * T;
        const double* yy = yc.as<double>() + cb * T;
        const int32_t* sf = st_fit.as<int32_t>() + cb;
        int32_t* sp = st_p.as<int32_t>() + cb;
        const int32_t* od = odd.as<int32_t>() + cb;
        int rc = SD_OK;
        switch (K) {
            case 13: rc = launch_fused_k<13>(ctx, nbc, lds, r, np, od, xc, yy, qc.as<double>(), Tq, T, cc, sf, sp, worklist, work_count, cb, pw, skip_prob); break;
            case 15: rc = launch_fused_k<15>(ctx, nbc, lds, r, np, od, xc, yy, qc.as<double>(), Tq, T, cc, sf, sp, worklist, work_count, cb, pw, skip_prob); break;
            default: rc = launch_fused_k<17>(ctx, nbc, lds, r, np, od, xc, yy, qc.as<double>(), Tq, T, cc, sf, sp, worklist, work_count, cb, pw, skip_prob); break;
        }
        SD_TRY(rc);
        if (runs_q) {
            SD_TRY(launch_untranspose_runs(ctx, (const double*)oc.p, (const unsigned short*)qtags.p, Tq, cc, out + cb, ld_out, skip_prob));
        } else {
            SD_LAUNCH(ctx, "analog_untranspose_kernel", analog_untranspose_kernel,
                      dim3((unsigned)((cc + 31) / 32), (unsigned)((Tq + 31) / 32), skip_prob ? 2 : 3), dim3(256), 0, (const double*)oc.p, Tq, cc,
                      out + cb, ld_out, skip_prob);
        }
    }
    int32_t nw = 0;
    SD_HIP(hipMemcpyAsync(&nw, work_count, sizeof(nw), hipMemcpyDeviceToHost, ctx->stream));
    if (cell_status) {
        SD_HIP(status_pub.alloc(ctx, sizeof(int32_t) * C));
        SD_LAUNCH(ctx, "analog_status_public_kernel", analog_status_public_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                  (const int32_t*)st_fit.p, (const int32_t*)st_p.p, C, status_pub.as<int32_t>());
        SD_HIP(hipMemcpyAsync(cell_status, status_pub.p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    }
    SD_HIP(hipStreamSynchronize(ctx->stream));
#ifdef SD_DEV
    if (sd_dev_env("SD_ANALOG_COUNT")) fprintf(stderr, "analog fit_predict: %d of %lld cells handed back to the split path\n", nw, (long long)C);
    if (sd_dev_env("SD_FUSED_TRACE")) {  // phase clocks (100 MHz ticks of s_memtime) of the first cells of workgroup 0, last chunk
        long long h[128];
        SD_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(sd_fused_trace), sizeof(h)));
        for (int r = 0; r < 8; ++r) {
            fprintf(stderr, "fused trace cell %d: runs %lld merge %lld tags+xs %lld search %lld yx %lld prefix %lld outputs %lld (means %lld, prefix of squares %lld, barrier %lld, spreads + stores %lld)\n", r, h[r * 16 + 1] - h[r * 16],
                    h[r * 16 + 2] - h[r * 16 + 1], h[r * 16 + 3] - h[r * 16 + 2], h[r * 16 + 4] - h[r * 16 + 3], h[r * 16 + 5] - h[r * 16 + 4],
                    h[r * 16 + 6] - h[r * 16 + 5], h[r * 16 + 7] - h[r * 16 + 6], h[r * 16 + 8] - h[r * 16 + 6], h[r * 16 + 9] - h[r * 16 + 8],
                    h[r * 16 + 10] - h[r * 16 + 9], h[r * 16 + 7] - h[r * 16 + 10]);
        }
    }
#endif
    if (nw == 0) return SD_OK;
    // cells handed back (ties among the training values or on a window boundary): the split path answers them.  Few: on packed
    // copies of their columns; many: the whole grid in place (the same numbers either way).
    // (what the split path finds for the cells it recomputes replaces what the fused pass reported for them)
    if ((int64_t)nw * 2 > C) return fit_predict_split(ctx, X, y, ld, T, F, C, Xq, ld_q, Tq, k, kind, has_thresh, thresh, out, ld_out, cell_status);
    sd_scratch Xw, yw, Qw, Ow;
    SD_HIP(Xw.alloc(ctx, sizeof(double) * (size_t)T * nw));
    SD_HIP(yw.alloc(ctx, sizeof(double) * (size_t)T * nw));
    SD_HIP(Qw.alloc(ctx, sizeof(double) * (size_t)Tq * nw));
    SD_HIP(Ow.alloc(ctx, sizeof(double) * (size_t)Tq * 3 * nw));
    auto blocks = [&](int64_t total) { return dim3((unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)ctx->cu_count * 16)); };
    SD_LAUNCH(ctx, "analog_gather_cells_kernel", analog_gather_cells_kernel, blocks(T * nw), dim3(256), 0, X, ld, T, (const int32_t*)worklist, (int64_t)nw, Xw.as<double>());
    SD_LAUNCH(ctx, "analog_gather_cells_kernel", analog_gather_cells_kernel, blocks(T * nw), dim3(256), 0, y, ld, T, (const int32_t*)worklist, (int64_t)nw, yw.as<double>());
    SD_LAUNCH(ctx, "analog_gather_cells_kernel", analog_gather_cells_kernel, blocks(Tq * nw), dim3(256), 0, Xq, ld_q, Tq, (const int32_t*)worklist, (int64_t)nw, Qw.as<double>());
    std::vector<int32_t> st_w((size_t)nw), cells_w((size_t)nw);
    SD_TRY(fit_predict_split(ctx, Xw.as<double>(), yw.as<double>(), nw, T, 1, nw, Qw.as<double>(), nw, Tq, k, kind, has_thresh, thresh, Ow.as<double>(), nw,
                             cell_status ? st_w.data() : nullptr));
    SD_LAUNCH(ctx, "analog_scatter_cells_kernel", analog_scatter_cells_kernel, blocks(3 * Tq * nw), dim3(256), 0, (const double*)Ow.p, 3 * Tq, (const int32_t*)worklist,
              (int64_t)nw, out, ld_out);
    if (cell_status) SD_HIP(hipMemcpyAsync(cells_w.data(), worklist, sizeof(int32_t) * (size_t)nw, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    if (cell_status)  // the recomputed cells report what the split path found for them
        for (int32_t j = 0; j < nw; ++j) cell_status[cells_w[(size_t)j]] = st_w[(size_t)j];
    return SD_OK;
}

}  // namespace

extern "C" {

int sd_analog_state_destroy(sd_analog_state* st) {
    if (!st) return SD_OK;
    if (st->ctx) {
        (void)hipSetDevice(st->ctx->device);
        (void)hipStreamSynchronize(st->ctx->stream);
    }
    sd_pool_release(st->ctx, st->X);
    sd_pool_release(st->ctx, st->y);
    sd_pool_release(st->ctx, st->status);
    sd_pool_release(st->ctx, st->xs);
    sd_pool_release(st->ctx, st->xi);
    sd_pool_release(st->ctx, st->yx);
    sd_pool_release(st->ctx, st->pq);
    sd_pool_release(st->ctx, st->ybar);
    sd_pool_release(st->ctx, st->rx);
    sd_pool_release(st->ctx, st->xbar);
    sd_pool_release(st->ctx, st->ps);
    delete st;
    return SD_OK;
}

int sd_analog_state_info(const sd_analog_state* st, int64_t* T, int* F, int64_t* C) {
    SD_CHECK_ARG(st, "state is NULL");
    if (T) *T = st->T;
    if (F) *F = st->F;
    if (C) *C = st->C;
    return SD_OK;
}

int sd_analog_fit_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                      sd_analog_state** out) {
    SD_CHECK_ARG(ctx && X_dev && y_dev && out, "sd_analog_fit: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && ld >= C, "sd_analog_fit: bad sizes");
    SD_CHECK_ARG(F >= 1 && F <= kMaxF, "sd_analog_fit: F=%d outside [1,%d]", F, kMaxF);
    *out = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    sd_analog_state* st = new sd_analog_state();
    st->ctx = ctx;
    st->T = T;
    st->F = F;
    st->C = C;
    auto body = [&]() -> int {
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->X, sizeof(double) * (size_t)T * F * C));
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->y, sizeof(double) * (size_t)T * C));
        SD_HIP(sd_pool_malloc(ctx, (void**)&st->status, sizeof(int32_t) * C));
        SD_HIP(hipMemsetAsync(st->status, 0, sizeof(int32_t) * C, ctx->stream));
        const size_t lds = (size_t)T * (sizeof(double) + sizeof(uint16_t));
        const bool f1_sorted = F == 1 && T <= 65535 && (lds <= ctx->lds_max || sort2_width(T, ctx->lds_max) != 0) &&
                               sizeof(double) * (size_t)(T + 1) <= ctx->lds_max;
        const int K2t = (f1_sorted && !sd_dev_env("SD_ANALOG_SORT1")) ? sort2_width(T, ctx->lds_max) : 0;
        bool tiled = K2t != 0 && tile_sort_applies(K2t, T, C, ctx->lds_max);
        sd_scratch runs_buf, odd_buf;
        int np_runs = 0;
        if (tiled) {
            const int64_t chunk = 64 * K2t;
            np_runs = (int)(((T + chunk - 1) / chunk) * chunk);
            if (runs_buf.alloc(ctx, sizeof(double) * (size_t)np_runs * (size_t)C) != hipSuccess) {  // no room for the runs: the two-transpose path
                (void)hipGetLastError();
                tiled = false;
            }
        }
        if (tiled) {
            // F == 1: one tile-shaped kernel makes the cell-major copies and the sorted runs of 64 * K keys (csrc: analog_tile_sort_kernel)
            SD_HIP(odd_buf.alloc(ctx, sizeof(int32_t) * (size_t)C));
            SD_HIP(hipMemsetAsync(odd_buf.p, 0, sizeof(int32_t) * (size_t)C, ctx->stream));
            SD_TRY(launch_tile_sort(ctx, K2t, X_dev, y_dev, ld, T, C, st->X, st->y, runs_buf.as<double>(), np_runs, st->status,
                                    odd_buf.as<int32_t>()));
        } else {
            dim3 grid((unsigned)((C + 31) / 32), (unsigned)((T + 31) / 32));
            for (int f = 0; f < F; ++f)
                SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, grid, dim3(256), 0, X_dev, ld, T, F, f, C,
                          st->X, st->status, 1);
            SD_LAUNCH(ctx, "analog_transpose_kernel", analog_transpose_kernel, grid, dim3(256), 0, y_dev, ld, T, 1, 0, C,
                      st->y, st->status, 0);
        }
        if (F == 1 && T <= 65535 && (lds <= ctx->lds_max || sort2_width(T, ctx->lds_max) != 0) &&
            sizeof(double) * (size_t)(T + 1) <= ctx->lds_max) {
            // sorted view for the 1-D fast path: values, original indices, and y in the same order
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->xs, sizeof(double) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->xi, sizeof(int32_t) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->yx, sizeof(double) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->ybar, sizeof(double) * C));
            const int K2 = sd_dev_env("SD_ANALOG_SORT1") ? 0 : sort2_width(T, ctx->lds_max);
            if (K2 == 0)
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_sort_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (K2 != 0) {
                // (no prefix sums yet: the BASELINE path -- analog_f1_mean3_kernel -- builds its own on chip; the kernels
                // that read them from memory get them from ensure_prefix_sums on their first call)
                Sort2Args a{st->X, T, 0, st->y, T, C, st->xs, st->xi, st->yx, nullptr, st->ybar};
                if (tiled && K2 == K2t) {
                    a.runs = runs_buf.as<double>();
                    a.np_runs = np_runs;
                    a.odd_flags = odd_buf.as<int32_t>();
                }
                SD_TRY(launch_sort2_width(ctx, K2, a));
            } else {
                SD_HIP(sd_pool_malloc(ctx, (void**)&st->pq, sizeof(double) * 2 * (size_t)(T + 1) * C));
                int nb = (int)std::min<int64_t>(C, (int64_t)ctx->cu_count * 4);
                SD_LAUNCH(ctx, "analog_sort_kernel", analog_sort_kernel, dim3(nb), dim3(1024), lds, (const double*)st->X,
                          (const double*)st->y, T, C, st->xs, st->xi, st->yx);
                SD_TRY(launch_prefix_sums(ctx, (const double*)st->yx, T, C, st->pq, st->ybar, 0));
            }
            SD_HIP(hipStreamSynchronize(ctx->stream));
        }
        const int Ks = F > 1 && sd_dev_env("SD_ANALOG_NOSLAB") == nullptr ? sort2_width(T, ctx->lds_max) : 0;
        if (Ks != 0) {
            // F > 1: training points in feature-0 order for the slab search (analog_slab_predict_kernel)
            sd_scratch keys;
            SD_HIP(keys.alloc(ctx, sizeof(double) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->xi, sizeof(int32_t) * (size_t)T * C));
            SD_HIP(sd_pool_malloc(ctx, (void**)&st->ps, sizeof(double) * (size_t)T * F * C));
            const Sort2Args a{st->X, (int64_t)F * T, 1, nullptr, T, C, keys.as<double>(), st->xi,
                              nullptr, nullptr, nullptr};
            SD_TRY(launch_sort2_width(ctx, Ks, a));
            SD_LAUNCH(ctx, "analog_gather_sorted_kernel", analog_gather_sorted_kernel,
                      dim3((unsigned)std::min<int64_t>(C, (int64_t)ctx->cu_count * 64)), dim3(256), 0, (const double*)st->X,
                      (const int32_t*)st->xi, T, F, C, st->ps);
            SD_HIP(hipStreamSynchronize(ctx->stream));
        }
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    };
    int rc = body();
    if (rc != SD_OK) {
        sd_analog_state_destroy(st);
        return rc;
    }
    *out = st;
    return SD_OK;
}

int sd_analog_fit(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, sd_analog_state** out) {
    SD_CHECK_ARG(ctx && X && y && out, "sd_analog_fit: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && F >= 1, "sd_analog_fit: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy;
    SD_HIP(dX.alloc(ctx, sizeof(double) * (size_t)T * F * C));
    SD_HIP(dy.alloc(ctx, sizeof(double) * (size_t)T * C));
    SD_TRY(sd_copy_h2d(ctx, dX.p, X, sizeof(double) * (size_t)T * F * C));
    SD_TRY(sd_copy_h2d(ctx, dy.p, y, sizeof(double) * (size_t)T * C));
    return sd_analog_fit_dev(ctx, dX.as<double>(), dy.as<double>(), C, T, F, C, out);
}

int sd_analog_predict_dev(sd_ctx* ctx, const sd_analog_state* st, const double* Xq_dev, int64_t ld, int64_t Tq, int k,
                          int kind, int has_thresh, double thresh, const int32_t* sample_inds_dev, double* out_dev,
                          int64_t ld_out, int64_t* inds_dev, double* dist_dev, int32_t* cell_status) {
    return predict_common(0, ctx, st, Xq_dev, ld, Tq, k, kind, has_thresh, thresh, sample_inds_dev, ld, out_dev, ld_out,
                          inds_dev, dist_dev, cell_status);
}

int sd_analog_predict(sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int kind,
                      int has_thresh, double thresh, const int32_t* sample_inds, double* out, int64_t* inds,
                      double* dist, int32_t* cell_status) {
    return predict_host(0, ctx, st, Xq, Tq, k, kind, has_thresh, thresh, sample_inds, out, inds, dist, cell_status);
}

int sd_analog_fit_predict_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                              const double* Xq_dev, int64_t ld_q, int64_t Tq, int k, int kind, int has_thresh, double thresh,
                              double* out_dev, int64_t ld_out, int32_t* cell_status) {
    return fit_predict_dev(ctx, X_dev, y_dev, ld, T, F, C, Xq_dev, ld_q, Tq, k, kind, has_thresh, thresh, out_dev, ld_out, cell_status);
}

int sd_analog_fit_predict(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, const double* Xq, int64_t Tq, int k,
                          int kind, int has_thresh, double thresh, double* out, int32_t* cell_status) {
    SD_CHECK_ARG(ctx && X && y && Xq && out, "sd_analog_fit_predict: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && Tq > 0 && F >= 1, "sd_analog_fit_predict: bad sizes");
    SD_HIP(hipSetDevice(ctx->device));
    sd_scratch dX, dy, dq, dout;
    const size_t xb = sizeof(double) * (size_t)T * F * C, yb = sizeof(double) * (size_t)T * C, qb = sizeof(double) * (size_t)Tq * F * C,
                 ob = sizeof(double) * (size_t)Tq * 3 * C;
    SD_HIP(dX.alloc(ctx, xb));
    SD_HIP(dy.alloc(ctx, yb));
    SD_HIP(dq.alloc(ctx, qb));
    SD_HIP(dout.alloc(ctx, ob));
    SD_TRY(sd_copy_h2d(ctx, dX.p, X, xb));
    SD_TRY(sd_copy_h2d(ctx, dy.p, y, yb));
    SD_TRY(sd_copy_h2d(ctx, dq.p, Xq, qb));
    SD_TRY(fit_predict_dev(ctx, dX.as<double>(), dy.as<double>(), C, T, F, C, dq.as<double>(), C, Tq, k, kind, has_thresh, thresh, dout.as<double>(), C,
                           cell_status));
    return sd_copy_d2h(ctx, out, dout.p, ob);
}

int sd_analogreg_predict_dev(sd_ctx* ctx, const sd_analog_state* st, const double* Xq_dev, int64_t ld, int64_t Tq,
                             int k, int has_thresh, double thresh, double* out_dev, int64_t ld_out, int32_t* cell_status) {
    return predict_common(1, ctx, st, Xq_dev, ld, Tq, k, SD_ANALOG_MEAN, has_thresh ? 1 : 0, thresh, nullptr, ld, out_dev, ld_out, nullptr,
                          nullptr, cell_status);
}

int sd_analogreg_predict(sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int has_thresh, double thresh,
                         double* out, int32_t* cell_status) {
    return predict_host(1, ctx, st, Xq, Tq, k, SD_ANALOG_MEAN, has_thresh ? 1 : 0, thresh, nullptr, out, nullptr, nullptr, cell_status);
}

}  // extern "C"
