// Internal definitions shared by the HIP translation units (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/sd_downscale.h"

// Internal per-cell status is a bitmask (atomicOr from many workgroups, order independent); it is
// folded into the public SD_CELL_* code with the reference's precedence on the way out.
// Environment variables that select alternative code paths (A/B measurements) exist only in development builds
// (-DSD_DEV: `make dev` -> lib/libsd_downscale_dev.so).  The production library never reads the environment.
#ifdef SD_DEV
#include <cstdlib>
static inline const char* sd_dev_env(const char* name) { return getenv(name); }
#else
static inline const char* sd_dev_env(const char*) { return nullptr; }
#endif

#define SDI_MASKED 1
#define SDI_NONFINITE 2
#define SDI_BAD_CLIMO 4
#define SDI_ONE_CLASS 8

struct sd_prof_entry {
    double ms = 0.0;
    int64_t launches = 0;
};

// Device copies of group tables (time indices ordered by (group, time) + offsets) are cached per context: a fit /
// predict pair, or repeated calls on the same calendar, upload them once (keyed by the group ids themselves).
struct sd_gt_cache_entry {
    std::vector<int32_t> gid;
    int G = 0;
    int32_t* order = nullptr;  // device [T]
    int32_t* off = nullptr;    // device [G+1]
    int nmax = 0;
    std::vector<int64_t> host_off;
    uint64_t last_use = 0;
};

struct sd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;  // sd_timer_*
    hipEvent_t p0 = nullptr, p1 = nullptr;  // (spare pair)
    bool prof_on = false;
    std::map<std::string, sd_prof_entry> prof;
    // per-kernel profile: an event pair around every launch, recorded without waiting; the elapsed times are read when the profile
    // is queried (sd_prof_resolve).  Waiting for every launch -- the first form -- put a host round trip between any two kernels of a
    // step: ~0.2 ms of the 15 ms of the headline step went to the measurement of its seven launches.
    struct sd_prof_pending {
        hipEvent_t e0, e1;
        const char* name;  // (string literals of the SD_LAUNCH sites)
    };
    std::vector<sd_prof_pending> prof_pending;
    std::vector<hipEvent_t> prof_free;
    hipEvent_t prof_open = nullptr;  // the begin event of the launch in progress
    int cu_count = 0;
    size_t lds_max = 0;
    // grow-only device workspace reused by calls on this context (hipMalloc/hipFree of GB-sized
    // scratch costs tens of ms per call)
    void* ws_ptr = nullptr;
    size_t ws_size = 0;
    // size-keyed cache of device blocks released by states / per-call scratch: hipMalloc + hipFree of a GB-sized
    // block cost tens of milliseconds, a fit -> predict -> destroy cycle would spend more time there than in kernels
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_live;
    size_t pool_cached = 0, pool_cap = 0;
    std::vector<sd_gt_cache_entry> gt_cache;  // at most kGtCacheEntries, least recently used entry replaced
    uint64_t gt_clock = 0;
    // pinned staging ring of the host-buffer entry points (sd_copy_h2d / sd_copy_d2h), created on first use
    static constexpr int kStageBufs = 3;
    static constexpr size_t kStageBytes = (size_t)64 << 20;
    void* stage[kStageBufs] = {nullptr, nullptr, nullptr};
    hipEvent_t stage_ev[kStageBufs] = {nullptr, nullptr, nullptr};
    hipEvent_t stage_join = nullptr;
    hipStream_t copy_stream = nullptr;
    // a second, smaller ring + stream for results going back to the host while the next inputs come in (sd_copy_d2h_2d)
    static constexpr int kDrainBufs = 2;
    void* drain[kDrainBufs] = {nullptr, nullptr};
    hipEvent_t drain_ev[kDrainBufs] = {nullptr, nullptr};
    hipStream_t drain_stream = nullptr;
};
void sd_gt_cache_clear(sd_ctx* ctx);
// Large host <-> device copies of the host-buffer entry points.  Pageable host memory goes through a ring of pinned
// staging buffers: several host threads copy a chunk into (out of) a pinned buffer while the DMA engine moves the
// previous chunk, so the transfer runs at the PCIe rate instead of the runtime's single-threaded pageable path.
// sd_copy_h2d: returns once every chunk is queued; later work on ctx->stream is ordered behind the transfer.
// sd_copy_d2h: ordered behind the work already queued on ctx->stream; returns when the host buffer is complete.
int sd_copy_h2d(sd_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int sd_copy_d2h(sd_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
// Column blocks of row-major fields ([rows] x width bytes, pitches in bytes), through the same kind of pinned rings.
// sd_copy_h2d_2d: like sd_copy_h2d.  sd_copy_d2h_2d: the source must be complete (no stream ordering); it runs on a ring and a
// stream of its own, so that it can be called from a second host thread while the first one uploads the next block.
int sd_copy_h2d_2d(sd_ctx* ctx, void* dst_dev, size_t dpitch, const void* src_host, size_t spitch, size_t width, size_t rows);
int sd_copy_d2h_2d(sd_ctx* ctx, void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t rows);
// asks for transparent huge pages on the 2 MB-aligned interior of a host range that is about to be written for the first
// time (a fresh result array; NumPy does the same for its own large allocations): first-touch faults of 4 KB pages cap a
// copy at ~14 GB/s on the GPU box, of huge pages at > 100 GB/s (csrc/microbench/pcie_rate.hip)
void sd_advise_result_buffer(void* p, size_t bytes);

// Device memory through the context's block cache (exact-size reuse).  Blocks go back with sd_pool_release;
// the cache is bounded by pool_cap (a quarter of the device memory) and emptied by sd_ctx_release_cached /
// sd_ctx_destroy or when an allocation fails.
hipError_t sd_pool_malloc(sd_ctx* ctx, void** p, size_t bytes);
void sd_pool_release(sd_ctx* ctx, void* p);
void sd_pool_trim(sd_ctx* ctx);

struct sd_bcsd_state {
    sd_ctx* ctx = nullptr;
    int kind = 0, G = 0, return_anoms = 1;
    int detrend = 0;  // QuantileMapper(detrend=True): ys holds the sorted detrended observations
    int qt_tails = 3, qt_endpoints = 10;  // sd_bcsd_state_set_tails: OLS continuation of the fitted CDF (lower | upper), points per line
    int64_t T = 0, C = 0;
    std::vector<int64_t> goff;  // host copy, [G+1]
    int nmax = 0;
    double* ys = nullptr;        // device [C][T] cell-major, group segments back to back
    double* x_climo = nullptr;   // device [C][G]
    double* y_climo = nullptr;   // device [C][G]
    double* y_trend = nullptr;   // device [C][G][2]: slope, intercept of the fitted segments' least-squares lines (detrend)
    int32_t* status = nullptr;   // device [C] internal bitmask
    int32_t* goff_dev = nullptr; // device [G+1]
};

struct sd_analog_state {
    sd_ctx* ctx = nullptr;
    int64_t T = 0, C = 0;
    int F = 0;
    double* X = nullptr;       // device [T,F,C]
    double* y = nullptr;       // device [T,C]
    int32_t* status = nullptr; // device [C] internal bitmask
    // F == 1 fast path: per cell training values sorted ascending + original indices
    double* xs = nullptr;   // device [C][T]
    int32_t* xi = nullptr;  // device [C][T]
    double* yx = nullptr;   // device [C][T]: y in the order of xs (analog values of a sorted-x window are contiguous)
    double* pq = nullptr;   // device [C][T+1][2]: exclusive prefix sums of (yx - ybar) and (yx - ybar)^2 (window mean / std in 2 loads)
    double* ybar = nullptr; // device [C]: mean of y
    double* rx = nullptr;   // device [C][T+1]: exclusive prefix sums of (xs - xbar)(yx - ybar); filled by the first one-feature AnalogRegression call
    double* xbar = nullptr; // device [C]: mean of x
    // F > 1 slab search: training points sorted by feature 0 (original indices in xi)
    double* ps = nullptr;   // device [C][F][T]
};

int sd_set_error(int code, const char* fmt, ...);
// Returns a device pointer to at least `bytes` of context-owned scratch (valid until the next call on
// the context asks for more).  Calls on a context are serialised, so one buffer is enough.
int sd_workspace(sd_ctx* ctx, size_t bytes, void** out);

#define SD_CHECK_ARG(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) return sd_set_error(SD_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define SD_HIP(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return sd_set_error(_e == hipErrorOutOfMemory ? SD_ERR_NOMEM : SD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                hipGetErrorString(_e), __FILE__, __LINE__);                               \
    } while (0)

#define SD_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc != SD_OK) return _rc; \
    } while (0)

// Launch helper: optional per-kernel event timing (sd_prof_enable).
int sd_prof_begin(sd_ctx* ctx);
int sd_prof_end(sd_ctx* ctx, const char* name);

#define SD_LAUNCH(ctx, name, kernel, grid, block, lds, ...)                                  \
    do {                                                                                     \
        SD_TRY(sd_prof_begin(ctx));                                                          \
        hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);            \
        SD_HIP(hipGetLastError());                                                           \
        SD_TRY(sd_prof_end(ctx, name));                                                      \
    } while (0)

// RAII device scratch buffer from the context's block cache (callers synchronise the stream before returning).
struct sd_scratch {
    void* p = nullptr;
    sd_ctx* owner = nullptr;
    hipError_t alloc(sd_ctx* ctx, size_t bytes) {
        owner = ctx;
        return sd_pool_malloc(ctx, &p, bytes);
    }
    ~sd_scratch() {
        if (p) sd_pool_release(owner, p);
    }
    template <typename T>
    T* as() { return static_cast<T*>(p); }
};

// Host-side group table: time indices ordered by (group, time) + offsets.
struct sd_group_table {
    std::vector<int32_t> order;
    std::vector<int64_t> off;  // [G+1]
    int nmax = 0;
};
int sd_build_group_table(const int32_t* gid, int64_t T, int G, sd_group_table* out);

// public status from internal bitmask
__host__ __device__ static inline int32_t sd_public_status(int32_t bits) {
    if (bits & SDI_MASKED) return SD_CELL_MASKED;
    if (bits & SDI_NONFINITE) return SD_CELL_NONFINITE;
    if (bits & SDI_BAD_CLIMO) return SD_CELL_BAD_CLIMO;
    if (bits & SDI_ONE_CLASS) return SD_CELL_ONE_CLASS;
    return SD_CELL_OK;
}
static inline int32_t sd_internal_status(int32_t code) {
    switch (code) {
        case SD_CELL_MASKED: return SDI_MASKED;
        case SD_CELL_NONFINITE: return SDI_NONFINITE;
        case SD_CELL_BAD_CLIMO: return SDI_BAD_CLIMO;
        case SD_CELL_ONE_CLASS: return SDI_ONE_CLASS;
        default: return 0;
    }
}
