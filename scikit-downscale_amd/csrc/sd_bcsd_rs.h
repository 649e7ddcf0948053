// Register/LDS merge-sort BCSD path (sd_bcsd_rs.hip): parameters and launcher.
#pragma once
#include "sd_internal.h"

namespace sdrs {

// MODE_RANK + MODE_APPLY = fit+predict (or predict from a state) as two kernels: RANK ranks every x_fut sample
// within its shifted segment, APPLY sorts y_obs (or reads the fitted state), maps the ranks and restores the shift.
// MODE_BOTH = RANK then APPLY of the same segment inside one workgroup: the ranks stay in registers and the second
// read of the x_fut tile comes from L2 / Infinity Cache (the workgroup fetched it microseconds earlier).
enum { MODE_FIT = 0, MODE_BOTH = 2, MODE_RANK = 3, MODE_APPLY = 4 };

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
    double* shift;                                     // RANK -> APPLY (TAS, optional): [C*G][K][64] rolling mean - x_climo
    int from_state;  // RANK/APPLY: 1 = predict from a fitted state (x_climo, y_climo, ys given), 0 = fit on the fly from X, y
    int identity;  // 1: every group has equal fit / predict length (inverse CDF = identity on ranks)
    long long* trace;  // development: per-phase wall-clock stamps of sampled workgroups (SD_RS_TRACE=1)
    int ablate;  // development knob (SD_RS_ABLATE bitmask): skip phases to measure their marginal cost
    // set by sd_bcsd_rs_launch: groups served by this launch (0 = all; bit g otherwise) and the per-segment strides of the
    // hand-off slabs (those of the widest kernel of the call, so that launches of different widths share the slabs)
    unsigned long long gmask;
    int slab_nr, slab_k;
};

}  // namespace sdrs

bool sd_bcsd_rs_supported(int nmax);
int sd_bcsd_rs_row_stride(int nmax);
// workspace bytes of the RANK -> APPLY hand-off slabs (both multiples of 256)
void sd_bcsd_rs_handoff_bytes(int nmax, int64_t C, int G, size_t* rank_bytes, size_t* shift_bytes);
// group_len (host, [G], may be NULL): longest segment (fit or predict) of every group.  When the call needs the 21-wide
// kernels but some groups fit 19 samples per lane (30-day months of a daily series), those groups get their own launch
// of the narrower, ~10 % cheaper kernels.
int sd_bcsd_rs_launch(sd_ctx* ctx, int mode, const sdrs::Params& p, int nmax, const int* group_len = nullptr);
