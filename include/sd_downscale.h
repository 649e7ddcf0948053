/* sd_downscale.h -- C ABI of the MI355X-native pointwise-downscaling engine.
 *
 * Drop-in boundary for scikit-downscale's per-grid-cell statistical hot path.  Every entry point
 * replaces a *batch over the cell axis* of one reference call (citations are file:line under
 * /root/reference/skdownscale/pointwise_models/):
 *
 *   sd_bcsd_fit*            <- core.py:86-96 looping BcsdTemperature.fit (bcsd.py:197-228) or
 *                              BcsdPrecipitation.fit (bcsd.py:115-147) over all cells
 *   sd_bcsd_predict*        <- core.py:137-141 looping BcsdTemperature.predict (bcsd.py:230-269) or
 *                              BcsdPrecipitation.predict (bcsd.py:149-185)
 *   sd_bcsd_fit_predict_dev <- both of the above fused (no persisted quantile state)
 *   sd_analog_fit*          <- AnalogBase.fit (gard.py:58-87)
 *   sd_analog_predict*      <- PureAnalog.predict (gard.py:273-364)
 *   sd_analogreg_predict*   <- AnalogRegression.predict, thresh=None (gard.py:152-224)
 *
 * Conventions
 *   - Plain pointers and sizes only.  All fields are float64, time-major with the cell axis
 *     fastest: X[t*ld + c] (the layout core.py:427-440 `_to_feature_x` produces), feature arrays
 *     X[(t*F + f)*ld + c].  `ld` >= C is the row stride in elements (host variants: ld == C).
 *   - `group_id[t]` in [0,G) is the time group (calendar month - 1, groupers.py:11-12); it is the
 *     same for every cell and always lives in HOST memory.
 *   - Host variants (`sd_xxx`) take host pointers and copy in/out; `_dev` variants take device
 *     pointers (HBM-resident fields) and never touch PCIe.  No pointer is retained after a call
 *     returns; persistent state lives behind opaque handles owned by the library.
 *   - Every function returns SD_OK (0) or an SD_ERR_* code; sd_last_error() gives the
 *     thread-local message.  Per-cell conditions (masked / non-finite / bad climatology) are not
 *     errors of the call: they are reported in `cell_status[C]` and the host raises like the
 *     reference does (base.py:18-20, bcsd.py:140-141, core.py:35-37).
 *   - A context is bound to one GPU and one HIP stream; calls on one context are serialised.
 */
#ifndef SD_DOWNSCALE_H
#define SD_DOWNSCALE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_VERSION 103  /* bump on every change of an exported signature: the Python loader refuses other versions */

/* return codes */
#define SD_OK 0
#define SD_ERR_INVALID 1     /* bad argument (NULL, size <= 0, group id out of range ...) */
#define SD_ERR_HIP 2         /* HIP runtime failure; message carries hipGetErrorString */
#define SD_ERR_UNSUPPORTED 3 /* valid request outside the engine's limits (e.g. segment too long) */
#define SD_ERR_NOMEM 4

/* BCSD variants */
#define SD_BCSD_TAS 0 /* BcsdTemperature */
#define SD_BCSD_PR 1  /* BcsdPrecipitation */

/* per-cell status */
#define SD_CELL_OK 0
#define SD_CELL_MASKED 1    /* core.py:35-37: first sample of X is NaN -> cell skipped, output NaN */
#define SD_CELL_NONFINITE 2 /* base.py:18-20: NaN/inf inside an active cell -> ValueError on host */
#define SD_CELL_BAD_CLIMO 3 /* bcsd.py:140-141: y climatology <= 0 with return_anoms */
#define SD_CELL_ONE_CLASS 4 /* gard.py:204-212: a thresholded regression met samples of one class only where the reference's
                             * LogisticRegression.fit raises -> ValueError on host */

/* PureAnalog kinds (gard.py:258-262) */
#define SD_ANALOG_BEST 0
#define SD_ANALOG_SAMPLE 1
#define SD_ANALOG_WEIGHT 2
#define SD_ANALOG_MEAN 3

/* quantile-mapping regressors (quantile.py:160-395, 556-636) */
#define SD_QM_REGRESSOR 0        /* QuantileMappingReressor */
#define SD_QM_EDCDF_DIFFERENCE 1 /* EquidistantCdfMatcher(kind='difference') */
#define SD_QM_EDCDF_RATIO 2      /* EquidistantCdfMatcher(kind='ratio') */

/* CunnaneTransformer (sd_qm_cunnane): direction and tail handling (quantile.py:424, 485-486, 527-528) */
#define SD_CUNNANE_FORWARD 0 /* transform: values -> plotting positions */
#define SD_CUNNANE_INVERSE 1 /* inverse_transform: plotting positions -> values */
#define SD_EXTRAP_NONE 0     /* extrapolate=None or '1to1': np.interp clamps at both ends */
#define SD_EXTRAP_MIN 1      /* lower tail extended */
#define SD_EXTRAP_MAX 2      /* upper tail extended */
#define SD_EXTRAP_BOTH 3
#define SD_EXTRAP_1TO1 4     /* regressors only: samples beyond the fitted X range keep their offset to it (quantile.py:277-310) */

/* synthetic field kinds (sd_synth_fill) */
#define SD_SYNTH_GAUSS 0
#define SD_SYNTH_PRECIP 1

typedef struct sd_ctx sd_ctx;
typedef struct sd_bcsd_state sd_bcsd_state;
typedef struct sd_analog_state sd_analog_state;
typedef struct sd_qm_state sd_qm_state;
typedef struct sd_linreg_state sd_linreg_state;
typedef struct sd_comm sd_comm;
#define SD_COMM_ID_BYTES 128 /* RCCL's ncclUniqueId */

/* ---- library / context ---------------------------------------------------------------------- */
int sd_version(void);
const char* sd_last_error(void);
int sd_device_count(int* count);
int sd_ctx_create(int device, sd_ctx** out);
int sd_ctx_destroy(sd_ctx* ctx);
int sd_ctx_synchronize(sd_ctx* ctx);
/* States and per-call scratch recycle device blocks through a per-context cache (at most 1/2 of the HBM, given back when an allocation fails);
 * this returns the cached blocks and the rank workspace to the driver. */
int sd_ctx_release_cached(sd_ctx* ctx);
int sd_ctx_device_info(sd_ctx* ctx, char* name, size_t name_len, int* compute_units, int64_t* hbm_bytes);

/* ---- device memory (so a non-torch host can keep fields resident in HBM) --------------------- */
int sd_dev_alloc(sd_ctx* ctx, size_t bytes, void** dptr);
int sd_dev_free(sd_ctx* ctx, void* dptr);
int sd_memcpy_h2d(sd_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int sd_memcpy_d2h(sd_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* float32 transport: a float32 grid crosses PCIe as float32 and is widened / narrowed on the device (exact
 * widening, round-to-nearest narrowing: what the reference's NumPy promotion / result cast do on the host, core.py:119).
 * n elements; src and dst device pointers, queued on the context's stream. */
int sd_convert_f32_to_f64_dev(sd_ctx* ctx, const float* src_dev, int64_t n, double* dst_dev);
int sd_convert_f64_to_f32_dev(sd_ctx* ctx, const double* src_dev, int64_t n, float* dst_dev);
int sd_memcpy_d2d(sd_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);

/* ---- timing: HIP events on the context's stream ---------------------------------------------- */
int sd_timer_start(sd_ctx* ctx);
int sd_timer_stop(sd_ctx* ctx, float* elapsed_ms); /* synchronises the stream */
/* per-kernel profile: when enabled every launch is bracketed by events (serialising). */
int sd_prof_enable(sd_ctx* ctx, int on);
int sd_prof_reset(sd_ctx* ctx);
int sd_prof_query(sd_ctx* ctx, const char* kernel_name, double* total_ms, int64_t* launches);
int sd_prof_names(sd_ctx* ctx, char* buf, size_t buf_len); /* ';'-separated kernel names */

/* ---- deterministic synthetic fields generated in HBM (mirror: skdownscale_amd/synth.py) ------- */
int sd_synth_fill(sd_ctx* ctx, double* out_dev, int64_t T, int64_t C, int64_t ld, int64_t c_offset, int64_t c_full,
                  int kind, uint64_t seed, uint32_t stream, const double* base_host /* [T] or NULL */, double amp,
                  double cell_scale, double p_dry, int32_t stream2 /* <0: none */, double amp2);

/* ---- BCSD quantile mapping --------------------------------------------------------------------
 * X: [T,C] model-historical field (TAS: used for x_climo; PR: only validated, may be NULL ->
 * mask/validation then use y).  y: [T,C] observations.  group_id: host int32[T].
 * The `return_anoms` argument of the fit entry points and of sd_bcsd_state_import / sd_bcsd_state_info is a bit set:
 * SD_BCSD_RETURN_ANOMS (bcsd.py:27,266-267 / 170-185) | SD_BCSD_QM_DETREND (qm_kwargs={'detrend': True}: quantile.py:95-98,
 * 128-145 -- every group's series loses its least-squares line over the sample index, trend.py:51-83, before the CDFs are
 * built; the predict line is added back re-based on the fitted intercept).  Plain 0 / 1 keep their old meaning.  Detrended
 * mapping serves group segments of up to 19 456 samples (SD_ERR_UNSUPPORTED beyond). */
#define SD_BCSD_RETURN_ANOMS 1
#define SD_BCSD_QM_DETREND 2
int sd_bcsd_fit(sd_ctx* ctx, int kind, const double* X, const double* y, const int32_t* group_id, int G, int64_t T,
                int64_t C, int return_anoms, sd_bcsd_state** out);
int sd_bcsd_fit_dev(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld,
                    const int32_t* group_id, int G, int64_t T, int64_t C, int return_anoms, sd_bcsd_state** out);
/* out: [Tp,C]; cell_status: host int32[C] (may be NULL).  Cells whose fit status != OK get NaN. */
int sd_bcsd_predict(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp, const int32_t* group_id_p, int64_t Tp,
                    double* out, int32_t* cell_status);
int sd_bcsd_predict_dev(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp_dev, int64_t ld,
                        const int32_t* group_id_p, int64_t Tp, double* out_dev, int64_t ld_out, int32_t* cell_status);
/* fused fit+predict for resident fields: one pass, no persisted state. */
/* Fit on explicitly listed groups: group_order[group_offsets[G]] = time indices group by group (a time step may belong
 * to several groups, or to none: the +-15-day day-of-year windows of time_grouper='daily_nasa-nex', groupers.py:19-89,
 * bcsd.py:36-38,50-55), group_offsets[G+1].  The state's series length (sd_bcsd_state_info T, y_sorted of the export)
 * is the number of listed entries. */
int sd_bcsd_fit_groups(sd_ctx* ctx, int kind, const double* X, const double* y, const int32_t* group_order,
                       const int64_t* group_offsets, int G, int64_t T, int64_t C, int return_anoms, sd_bcsd_state** out);
int sd_bcsd_fit_groups_dev(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld, const int32_t* group_order,
                           const int64_t* group_offsets, int G, int64_t T, int64_t C, int return_anoms, sd_bcsd_state** out);
/* Predict with a climate-trend grouper of its own (bcsd.py:247-267): the 9-sample rolling mean runs over the groups of
 * trend_group_id (climate_trend, G_trend groups), the climatologies and the quantile mapping use group_id_p (groups of
 * the fitted state).  BcsdPrecipitation has no trend: same as sd_bcsd_predict. */
int sd_bcsd_predict_trend(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp, const int32_t* group_id_p,
                          const int32_t* trend_group_id, int G_trend, int64_t Tp, double* out, int32_t* cell_status);
int sd_bcsd_predict_trend_dev(sd_ctx* ctx, const sd_bcsd_state* st, const double* Xp_dev, int64_t ld, const int32_t* group_id_p,
                              const int32_t* trend_group_id, int G_trend, int64_t Tp, double* out_dev, int64_t ld_out,
                              int32_t* cell_status);
int sd_bcsd_fit_predict_dev(sd_ctx* ctx, int kind, const double* X_dev, const double* y_dev, int64_t ld,
                            const int32_t* group_id, int G, int64_t T, int64_t C, int return_anoms,
                            const double* Xp_dev, int64_t ld_p, const int32_t* group_id_p, int64_t Tp,
                            double* out_dev, int64_t ld_out, int32_t* cell_status);
int sd_bcsd_state_info(const sd_bcsd_state* st, int* kind, int* G, int64_t* T, int64_t* C, int* return_anoms);
int sd_bcsd_state_status(const sd_bcsd_state* st, int32_t* cell_status /* host [C] */);
/* Plain-array view of the fitted state (pickling / get_attr): y_sorted is CELL-major [C][T] with the
 * G group segments of a cell stored back to back at group_offsets[g]; climatologies are [C][G]. */
int sd_bcsd_state_export(const sd_bcsd_state* st, double* y_sorted, double* x_climo, double* y_climo,
                         int32_t* cell_status, int64_t* group_offsets /* [G+1] */);
int sd_bcsd_state_import(sd_ctx* ctx, int kind, int G, int64_t T, int64_t C, int return_anoms, const double* y_sorted,
                         const double* x_climo, const double* y_climo, const int32_t* cell_status,
                         const int64_t* group_offsets, sd_bcsd_state** out);
/* SD_BCSD_QM_DETREND states: slope and intercept of the fitted segments' lines, [C][G][2] (x_trend_fit_.lr_model_.coef_ /
 * .intercept_ of every group's QuantileMapper, quantile.py:97,145).  set_trend completes a state made by
 * sd_bcsd_state_import (only the intercepts enter predictions). */
/* Tail handling of the fitted inverse CDF (qm_kwargs={'qt_kwargs': {'extrapolate': ..., 'n_endpoints': ...}}: bcsd.py:59-67 ->
 * quantile.py:418-431, 523-545): which side continues along the least-squares line through the first / last n_endpoints
 * points of the fitted CDF (SD_QT_TAIL_LOWER | SD_QT_TAIL_UPPER = 'both', the default; 'min' = lower only, 'max' = upper only,
 * None / '1to1' = neither: np.interp then holds the end value), and n_endpoints >= 1 (default 10).  Reaches predictions only when
 * a predict group is longer than its fit group.  Applies to later sd_bcsd_predict* calls on the state; group segments of up to
 * 2 112 samples (SD_ERR_UNSUPPORTED beyond for non-default settings).  (`alpha` / `beta` of the CunnaneTransformer never reach
 * the result in the reference -- quantile.py:462 calls plotting_positions(n) with its defaults -- and have no entry here.) */
#define SD_QT_TAIL_LOWER 1
#define SD_QT_TAIL_UPPER 2
int sd_bcsd_state_set_tails(sd_bcsd_state* st, int extrapolate, int n_endpoints);
int sd_bcsd_state_get_trend(const sd_bcsd_state* st, double* y_trend);
int sd_bcsd_state_set_trend(sd_bcsd_state* st, const double* y_trend);
int sd_bcsd_state_destroy(sd_bcsd_state* st);

/* ---- GARD analogs ----------------------------------------------------------------------------
 * X: [T,F,C], y: [T,C], Xq: [Tq,F,C]; out: [Tq,3,C] columns pred / exceedance_prob /
 * prediction_error (gard.py:254-255).  inds: int64 [Tq,k,C] training indices, ascending distance
 * (may be NULL); dist: float64 [Tq,k,C] (may be NULL).  sample_inds: int32 [Tq,C], required for
 * SD_ANALOG_SAMPLE (the host draws them with np.random.randint like gard.py:315).  k == 1 is 'best_analog' whatever
 * `kind` says (gard.py:291-296). */
int sd_analog_fit(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, sd_analog_state** out);
int sd_analog_fit_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                      sd_analog_state** out);
int sd_analog_predict(sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int kind,
                      int has_thresh, double thresh, const int32_t* sample_inds, double* out, int64_t* inds,
                      double* dist, int32_t* cell_status);
int sd_analog_predict_dev(sd_ctx* ctx, const sd_analog_state* st, const double* Xq_dev, int64_t ld, int64_t Tq, int k,
                          int kind, int has_thresh, double thresh, const int32_t* sample_inds_dev, double* out_dev,
                          int64_t ld_out, int64_t* inds_dev, double* dist_dev, int32_t* cell_status);
/* fit + predict in one call, no fitted state left behind (AnalogBase.fit, gard.py:58-87, followed by PureAnalog.predict,
 * gard.py:273-364, on the same object): for callers that drop the estimator after predicting.  Results are bit-identical to
 * sd_analog_fit* -> sd_analog_predict*; F == 1, `SD_ANALOG_MEAN` without a threshold (or k == 1) on series the tile-shaped fit
 * serves run as one fused per-cell kernel, everything else takes the two calls internally.  SD_ANALOG_SAMPLE is not served
 * (it needs sample_inds: use the split calls). */
int sd_analog_fit_predict(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, const double* Xq, int64_t Tq,
                          int k, int kind, int has_thresh, double thresh, double* out, int32_t* cell_status);
int sd_analog_fit_predict_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                              const double* Xq_dev, int64_t ld_q, int64_t Tq, int k, int kind, int has_thresh, double thresh,
                              double* out_dev, int64_t ld_out, int32_t* cell_status);
/* AnalogRegression.predict (gard.py:152-224).  has_thresh: exceedance_prob from a logistic regression of (analog value >
 * thresh) on the analogs' features (the reference reports predict_proba(x)[0, 0], the probability of NOT exceeding;
 * 1.0 when every analog exceeds), linear model and RMSE on the exceeding analogs; a query without any exceeding analog
 * sets SD_CELL_ONE_CLASS for its cell (the reference raises there). */
int sd_analogreg_predict(sd_ctx* ctx, const sd_analog_state* st, const double* Xq, int64_t Tq, int k, int has_thresh, double thresh,
                         double* out, int32_t* cell_status);
int sd_analogreg_predict_dev(sd_ctx* ctx, const sd_analog_state* st, const double* Xq_dev, int64_t ld, int64_t Tq,
                             int k, int has_thresh, double thresh, double* out_dev, int64_t ld_out, int32_t* cell_status);
int sd_analog_state_info(const sd_analog_state* st, int64_t* T, int* F, int64_t* C);
int sd_analog_state_destroy(sd_analog_state* st);

/* ---- quantile-mapping regressors ----------------------------------------------------------------
 * Replace core.py:86-96 / 137-141 looping QuantileMappingReressor (quantile.py:160-395) or
 * EquidistantCdfMatcher (quantile.py:556-636) per cell.  X, y: [T, C] float64, cells contiguous; the whole series is one
 * segment.  model: SD_QM_REGRESSOR / SD_QM_EDCDF_DIFFERENCE / SD_QM_EDCDF_RATIO.  extrapolate: SD_EXTRAP_NONE (None),
 * SD_EXTRAP_MIN / MAX / BOTH (synthetic end points at -+1e20 from a least-squares line through the n_endpoints outermost
 * points, quantile.py:312-387) or SD_EXTRAP_1TO1.  Series up to 19 456 samples. */
int sd_qm_fit(sd_ctx* ctx, const double* X, const double* y, int64_t T, int64_t C, sd_qm_state** out);
int sd_qm_fit_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int64_t C,
                  sd_qm_state** out);
int sd_qm_predict(sd_ctx* ctx, const sd_qm_state* st, int model, int extrapolate, int n_endpoints, const double* Xp, int64_t Tp,
                  double* out, int32_t* cell_status);
int sd_qm_predict_dev(sd_ctx* ctx, const sd_qm_state* st, int model, int extrapolate, int n_endpoints, const double* Xp_dev,
                      int64_t ld, int64_t Tp, double* out_dev, int64_t ld_out, int32_t* cell_status);
/* CunnaneTransformer.transform / inverse_transform (quantile.py:465-545) on the sorted X of a fitted state
 * (sd_qm_fit accepts y = NULL for this use).  X: [Tp, C]; out: [Tp, C].
 * FORWARD: np.interp(x, sorted X, Cunnane positions); a value beyond an extended tail yields -inf / +inf (the
 * reference's own tail code for this direction, quantile.py:497/501, cannot run: it calls .values on an ndarray).
 * INVERSE: np.interp(p, positions, sorted X); beyond an extended tail the centred least-squares line through the
 * n_endpoints first / last (position, value) pairs (quantile.py:532-543). */
int sd_qm_cunnane(sd_ctx* ctx, const sd_qm_state* st, int direction, int extrapolate, int n_endpoints, const double* X,
                  int64_t Tp, double* out, int32_t* cell_status);
int sd_qm_cunnane_dev(sd_ctx* ctx, const sd_qm_state* st, int direction, int extrapolate, int n_endpoints,
                      const double* X_dev, int64_t ld, int64_t Tp, double* out_dev, int64_t ld_out,
                      int32_t* cell_status);
int sd_qm_state_info(const sd_qm_state* st, int64_t* T, int64_t* C);
/* sorted fit series [C][T] (cell-major) and per-cell status; any pointer may be NULL */
int sd_qm_state_export(const sd_qm_state* st, double* x_sorted, double* y_sorted, int32_t* cell_status);
int sd_qm_state_destroy(sd_qm_state* st);

/* ---- PureRegression ----------------------------------------------------------------------------------
 * Replaces core.py:86-96 / 137-141 looping gard.py:410-470: per cell an ordinary least-squares fit of y [T, C] on
 * X [T, F, C] (centred lstsq like sklearn's LinearRegression, minimum-norm for collinear features) and its RMSE
 * (fit_error_); predict writes out [Tq, 3, C] = pred / exceedance_prob / fit_error_ (gard.py:254-255 column order).
 * has_thresh (gard.py:416-437): the linear model uses the samples with y > thresh, exceedance_prob =
 * predict_proba(X)[:, 1] of a logistic regression of (y > thresh) on the features (L2, C = 1: sklearn's defaults); a cell
 * whose samples all exceed drops its threshold (probability 1), one without any exceeding sample gets SD_CELL_ONE_CLASS. */
int sd_linreg_fit(sd_ctx* ctx, const double* X, const double* y, int64_t T, int F, int64_t C, int has_thresh, double thresh,
                  sd_linreg_state** out);
int sd_linreg_fit_dev(sd_ctx* ctx, const double* X_dev, const double* y_dev, int64_t ld, int64_t T, int F, int64_t C,
                      int has_thresh, double thresh, sd_linreg_state** out);
int sd_linreg_predict(sd_ctx* ctx, const sd_linreg_state* st, const double* Xq, int64_t Tq, double* out,
                      int32_t* cell_status);
int sd_linreg_predict_dev(sd_ctx* ctx, const sd_linreg_state* st, const double* Xq_dev, int64_t ld, int64_t Tq,
                          double* out_dev, int64_t ld_out, int32_t* cell_status);
int sd_linreg_state_info(const sd_linreg_state* st, int64_t* T, int* F, int64_t* C);
/* coef [F][C], intercept [C], fit_error [C], logistic [F+1][C] (coefficients, then the intercept; models with a threshold),
 * thresh_dropped [C], status [C]; any pointer may be NULL */
int sd_linreg_state_export(const sd_linreg_state* st, double* coef, double* intercept, double* fit_error, double* logistic,
                           int32_t* thresh_dropped, int32_t* cell_status);
/* device state from exported numbers (pickling, checkpoint / resume); logistic = thresh_dropped = NULL without a threshold */
int sd_linreg_state_import(sd_ctx* ctx, int64_t T, int F, int64_t C, const double* coef, const double* intercept, const double* fit_error,
                           const double* logistic, const int32_t* thresh_dropped, const int32_t* cell_status, sd_linreg_state** out);
int sd_linreg_state_destroy(sd_linreg_state* st);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (no PyTorch) -----------------------------------
 * The reference's only parallelism is dask's map_blocks over spatial chunks (core.py:256-262, 300-336) and a client-side
 * collection of the result; here cells are block-partitioned over the ranks of one node, fit / predict need no exchange
 * (cells are independent, core.py:87), and the predicted shards are gathered to a root GPU.
 * Bootstrap: rank 0 calls sd_comm_unique_id and hands the 128 bytes to the other ranks by any channel (the Python side
 * uses a TCP socket next to MASTER_PORT); every rank then calls sd_comm_create on its own context.
 * sd_comm_gather_field: every rank contributes one contiguous [T, cells[rank]] float64 field; the root receives block r at
 * offset sum_{q<r} T * cells[q] of root_dev (layout [rank][T][C_r]: no padding, no concatenation copy).  The transfer is
 * ordered behind the work already queued on the context and runs on the communicator's own stream: with wait = 0 the next
 * chunk of cells can be computed meanwhile, sd_comm_wait joins. */
int sd_comm_unique_id(char* id /* [SD_COMM_ID_BYTES] */);
int sd_comm_create(sd_ctx* ctx, const char* id /* [SD_COMM_ID_BYTES] */, int rank, int world, sd_comm** out);
int sd_comm_destroy(sd_comm* comm);
int sd_comm_info(const sd_comm* comm, int* rank, int* world);
/* what RCCL reports about the communicator: ncclGetVersion's code, ncclCommCount (the ranks RCCL sees) and its device */
int sd_comm_rccl_info(const sd_comm* comm, int* version, int* ranks, int* device);
int sd_comm_barrier(sd_comm* comm);
int sd_comm_allreduce_max(sd_comm* comm, double value, double* result);
int sd_comm_gather_field(sd_comm* comm, const double* local_dev, int64_t T, const int64_t* cells /* [world] */,
                         double* root_dev /* root only */, int root, int wait);
int sd_comm_wait(sd_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* SD_DOWNSCALE_H */
