// Register-sort BCSD paths (sd_bcsd_fx.hip: one fused kernel per segment; sd_bcsd_rs.hip: the
// RANK / APPLY / FIT kernels): parameters and launchers.
#pragma once
#include "sd_internal.h"

namespace sdrs {

// MODE_RANK + MODE_APPLY = fit+predict (or predict from a state) as two kernels: RANK ranks every x_fut sample
// within its shifted segment by an explicit search, APPLY sorts y_obs (or reads the fitted state), maps the ranks and
// restores the shift.  They serve BcsdPrecipitation, and every BcsdTemperature segment the fused kernel hands back
// (work list).  MODE_FIT = the y side alone, writing the fitted state.
enum { MODE_FIT = 0, MODE_RANK = 3, MODE_APPLY = 4 };

struct Params {
    int kind, G, return_anoms, RS;
    int64_t C, Tf, ntiles;
    const double* X; const double* y; int64_t ld;    // fit fields [Tf, ld]
    const double* Xp; int64_t ld_p;                    // predict field [Tp, ld_p]
    double* out; int64_t ld_out;
    const int32_t* ord_f; const int32_t* off_f;        // fit group table
    const int32_t* ord_p; const int32_t* off_p;        // predict group table
    const int32_t* qidx; const double* qval;           // inverse-CDF tables, indexed off_p[g] + rank
    double* ys; double* x_climo; double* y_climo;      // state: [C][Tf], [C][G], [C][G]
    int32_t* status_fit; int32_t* status_p;
    uint32_t* ranks;                                   // RANK -> APPLY: [C*G][(K+1)/2][64] packed 16-bit ranks
    double* shift;                                     // [C*G][K][64] rolling mean - x_climo of every sample (lane layout)
    int from_state;  // 1 = predict from a fitted state (x_climo, y_climo, ys given), 0 = fit on the fly from X, y
    int identity;    // 1: every group has equal fit / predict length (inverse CDF = identity on ranks)
    long long* trace;  // SD_RS_TRACING builds: per-phase wall-clock stamps of sampled workgroups
    // set by the launchers: groups served by this launch (0 = all; bit g otherwise) and the per-segment strides of the
    // hand-off slabs (those of the widest kernel of the call, so that launches of different widths share the slabs)
    unsigned long long gmask;
    int slab_nr, slab_k;
    // Work list of (tile, group) items = tile * G + group.  The fused kernel appends the items it cannot serve
    // (tied samples, see sd_bcsd_fx.hip); RANK / APPLY launched with use_worklist walk the list with a
    // fixed grid instead of covering every (tile, group).
    int64_t* worklist;
    int* work_count;  // device counter (items appended, may exceed work_cap: the excess is lost and reported)
    int work_cap;
    int use_worklist;
    // second list of the compacting precipitation kernel (segments with too many wet days for its narrow sort): served by
    // bcsd_fxp_kernel<K, true, true> launched with use_worklist = 2
    int64_t* worklist2;
    int* work_count2;
    // QuantileMapper(detrend=True) (quantile.py:95-98, 128-145): both series lose their least-squares line over the sample
    // index before the CDFs; the predict line comes back afterwards, re-based on the fitted intercept.  RANK / APPLY / FIT
    // only (the fused kernel is bypassed).
    int detrend;
    double* trend_u;       // RANK -> APPLY: [C*G][2] slope, intercept of the predict segment's line
    double* y_trend;       // state [C][G][2]: slope, intercept of the fitted segment's line
    int n_endpoints;  // points of the OLS tail lines (quantile.py:426, 537-541); the launchers turn 0 into the default 10
    int dev_flags;  // development library only (SD_FZ_ABLATE): phases skipped to time the rest; results are then wrong
};

}  // namespace sdrs

bool sd_bcsd_rs_supported(int nmax);
int sd_bcsd_rs_width(int nmax);  // samples per lane (K) of the kernels serving segments of up to nmax samples
int sd_bcsd_rs_row_stride(int nmax);
// workspace bytes of the RANK -> APPLY hand-off slabs (both multiples of 256)
void sd_bcsd_rs_handoff_bytes(int nmax, int64_t C, int G, size_t* rank_bytes, size_t* shift_bytes);
// group_len (host, [G], may be NULL): longest segment (fit or predict) of every group.  When the call needs the 21-wide
// kernels but some groups fit 19 samples per lane (30-day months of a daily series), those groups get their own launch
// of the narrower, ~10 % cheaper kernels.
int sd_bcsd_rs_launch(sd_ctx* ctx, int mode, const sdrs::Params& p, int nmax, const int* group_len = nullptr);
// Fused kernels (sd_bcsd_fx.hip; BcsdTemperature and BcsdPrecipitation, segments of up to 1 536 samples): x side, y side,
// inverse CDF and shift / ratio of a segment in one workgroup pass on the register-resident 32-bit key sort of sd_wsort.h;
// segments they cannot serve are appended to p.worklist (the caller then runs RANK + APPLY with use_worklist).
bool sd_bcsd_fx_supported(int nmax);
int sd_bcsd_fx_launch(sd_ctx* ctx, const sdrs::Params& p, int nmax, const int* group_len = nullptr);
