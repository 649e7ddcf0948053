// Context, device memory, timers, per-kernel profile and the on-device synthetic field generator.
#include <cstring>

#include <algorithm>
#include <cstring>
#include <thread>

#include <sys/mman.h>

#include "sd_internal.h"

static thread_local std::string g_last_error;

int sd_set_error(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

void sd_gt_cache_clear(sd_ctx* ctx) {
    for (auto& e : ctx->gt_cache) {
        if (e.order) (void)hipFree(e.order);
        if (e.off) (void)hipFree(e.off);
    }
    ctx->gt_cache.clear();
}

// ---- pinned, chunked, double-buffered host <-> device copies ------------------------------------------------------
namespace {

constexpr size_t kDirectCopyBytes = (size_t)8 << 20;  // smaller copies take the runtime's own path

int stage_init(sd_ctx* ctx) {
    if (ctx->copy_stream) return SD_OK;
    SD_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    SD_HIP(hipEventCreateWithFlags(&ctx->stage_join, hipEventDisableTiming));
    for (int b = 0; b < sd_ctx::kStageBufs; ++b) {
        SD_HIP(hipHostMalloc(&ctx->stage[b], sd_ctx::kStageBytes, hipHostMallocDefault));
        SD_HIP(hipEventCreateWithFlags(&ctx->stage_ev[b], hipEventDisableTiming));
    }
    return SD_OK;
}

void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    unsigned nt = std::thread::hardware_concurrency() / 8;
    nt = nt < 2 ? 2 : (nt > 16 ? 16 : nt);
    const size_t part = ((bytes / nt) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) {
        const size_t off = t * part;
        if (off >= bytes) break;
        const size_t n = std::min(part, bytes - off);
        th.emplace_back([=]() { memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, n); });
    }
    memcpy(dst, src, std::min(part, bytes));
    for (auto& t : th) t.join();
}

// rows x width bytes between a pitched and a packed buffer (to_packed: pitched -> packed), rows split over a few threads
void parallel_rows(char* packed, char* pitched, size_t pitch, size_t width, size_t rows, bool to_packed) {
    // (measured on the 256-thread host of the GPU box: 16 threads move a column block at 33 GB/s against 50 GB/s for a
    // contiguous field -- a TLB miss per row on the pitched side; 32 threads are slower, their start-up shows per 64 MB chunk)
    unsigned nt = std::thread::hardware_concurrency() / 8;
    nt = nt < 2 ? 2 : (nt > 16 ? 16 : nt);
    const size_t part = (rows + nt - 1) / nt;
    auto work = [=](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            if (to_packed) memcpy(packed + r * width, pitched + r * pitch, width);
            else memcpy(pitched + r * pitch, packed + r * width, width);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) {
        const size_t r0 = t * part;
        if (r0 >= rows) break;
        th.emplace_back(work, r0, std::min(rows, r0 + part));
    }
    work(0, std::min(rows, part));
    for (auto& t : th) t.join();
}

int drain_init(sd_ctx* ctx) {
    if (ctx->drain_stream) return SD_OK;
    SD_HIP(hipStreamCreateWithFlags(&ctx->drain_stream, hipStreamNonBlocking));
    for (int b = 0; b < sd_ctx::kDrainBufs; ++b) {
        SD_HIP(hipHostMalloc(&ctx->drain[b], sd_ctx::kStageBytes, hipHostMallocDefault));
        SD_HIP(hipEventCreateWithFlags(&ctx->drain_ev[b], hipEventDisableTiming));
    }
    return SD_OK;
}

}  // namespace

int sd_copy_h2d_2d(sd_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows) {
    if (width == spitch && width == dpitch) return sd_copy_h2d(ctx, dst, src, width * rows);
    if (rows == 0) return SD_OK;
    SD_CHECK_ARG(width > 0 && width <= sd_ctx::kStageBytes, "sd_copy_h2d_2d: row of %zu bytes", width);
    SD_TRY(stage_init(ctx));
    SD_HIP(hipEventRecord(ctx->stage_join, ctx->stream));
    SD_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->stage_join, 0));
    const size_t per = sd_ctx::kStageBytes / width;
    size_t r0 = 0;
    for (int i = 0; r0 < rows; ++i) {
        const int b = i % sd_ctx::kStageBufs;
        const size_t n = std::min(per, rows - r0);
        if (i >= sd_ctx::kStageBufs) SD_HIP(hipEventSynchronize(ctx->stage_ev[b]));
        parallel_rows(static_cast<char*>(ctx->stage[b]), const_cast<char*>(static_cast<const char*>(src)) + r0 * spitch, spitch, width, n, true);
        if (dpitch == width) {  // packed destination: one linear transfer (2-D copies are served by a slower path)
            SD_HIP(hipMemcpyAsync(static_cast<char*>(dst) + r0 * width, ctx->stage[b], n * width, hipMemcpyHostToDevice, ctx->copy_stream));
        } else {
            SD_HIP(hipMemcpy2DAsync(static_cast<char*>(dst) + r0 * dpitch, dpitch, ctx->stage[b], width, width, n, hipMemcpyHostToDevice,
                                    ctx->copy_stream));
        }
        SD_HIP(hipEventRecord(ctx->stage_ev[b], ctx->copy_stream));
        r0 += n;
    }
    SD_HIP(hipEventRecord(ctx->stage_join, ctx->copy_stream));
    SD_HIP(hipStreamWaitEvent(ctx->stream, ctx->stage_join, 0));
    SD_HIP(hipStreamSynchronize(ctx->copy_stream));
    return SD_OK;
}

int sd_copy_d2h_2d(sd_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows) {
    SD_CHECK_ARG(width > 0 && width <= sd_ctx::kStageBytes, "sd_copy_d2h_2d: row of %zu bytes", width);
    SD_HIP(hipSetDevice(ctx->device));  // (possibly the first HIP call of this host thread)
    SD_TRY(drain_init(ctx));
    const size_t per = sd_ctx::kStageBytes / width;
    const size_t nch = (rows + per - 1) / per;
    auto issue = [&](size_t i) -> int {
        const size_t r0 = i * per, n = std::min(per, rows - r0);
        const int b = (int)(i % sd_ctx::kDrainBufs);
        if (spitch == width) {
            SD_HIP(hipMemcpyAsync(ctx->drain[b], static_cast<const char*>(src) + r0 * width, n * width, hipMemcpyDeviceToHost, ctx->drain_stream));
        } else {
            SD_HIP(hipMemcpy2DAsync(ctx->drain[b], width, static_cast<const char*>(src) + r0 * spitch, spitch, width, n, hipMemcpyDeviceToHost,
                                    ctx->drain_stream));
        }
        SD_HIP(hipEventRecord(ctx->drain_ev[b], ctx->drain_stream));
        return SD_OK;
    };
    if (nch > 0) SD_TRY(issue(0));
    for (size_t i = 0; i < nch; ++i) {
        if (i + 1 < nch) SD_TRY(issue(i + 1));  // (the other buffer: drained in iteration i - 1)
        const size_t r0 = i * per, n = std::min(per, rows - r0);
        const int b = (int)(i % sd_ctx::kDrainBufs);
        SD_HIP(hipEventSynchronize(ctx->drain_ev[b]));
        parallel_rows(static_cast<char*>(ctx->drain[b]), static_cast<char*>(dst) + r0 * dpitch, dpitch, width, n, false);
    }
    return SD_OK;
}

int sd_copy_h2d(sd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes < kDirectCopyBytes) {
        SD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SD_OK;
    }
    SD_TRY(stage_init(ctx));
    // the destination may still be read by work queued earlier (recycled blocks): the transfer starts behind it
    SD_HIP(hipEventRecord(ctx->stage_join, ctx->stream));
    SD_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->stage_join, 0));
    size_t off = 0;
    for (int i = 0; off < bytes; ++i) {
        const int b = i % sd_ctx::kStageBufs;
        const size_t n = std::min(sd_ctx::kStageBytes, bytes - off);
        if (i >= sd_ctx::kStageBufs) SD_HIP(hipEventSynchronize(ctx->stage_ev[b]));  // the DMA out of this buffer is done
        parallel_memcpy(ctx->stage[b], static_cast<const char*>(src) + off, n);
        SD_HIP(hipMemcpyAsync(static_cast<char*>(dst) + off, ctx->stage[b], n, hipMemcpyHostToDevice, ctx->copy_stream));
        SD_HIP(hipEventRecord(ctx->stage_ev[b], ctx->copy_stream));
        off += n;
    }
    SD_HIP(hipEventRecord(ctx->stage_join, ctx->copy_stream));
    SD_HIP(hipStreamWaitEvent(ctx->stream, ctx->stage_join, 0));
    // the staging buffers are reused by the next call: wait for the last DMAs here (the kernels need the data anyway)
    SD_HIP(hipStreamSynchronize(ctx->copy_stream));
    return SD_OK;
}

void sd_advise_result_buffer(void* p, size_t bytes) {
    const uintptr_t huge = (uintptr_t)2 << 20;
    const uintptr_t lo = ((uintptr_t)p + huge - 1) & ~(huge - 1), hi = ((uintptr_t)p + bytes) & ~(huge - 1);
    if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);  // (a hint: failure changes nothing)
}

int sd_copy_d2h(sd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    sd_advise_result_buffer(dst, bytes);
    if (bytes < kDirectCopyBytes) {
        SD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
        return SD_OK;
    }
    SD_TRY(stage_init(ctx));
    SD_HIP(hipEventRecord(ctx->stage_join, ctx->stream));
    SD_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->stage_join, 0));
    const int nchunks = (int)((bytes + sd_ctx::kStageBytes - 1) / sd_ctx::kStageBytes);
    auto issue = [&](int i) -> int {
        const size_t off = (size_t)i * sd_ctx::kStageBytes, n = std::min(sd_ctx::kStageBytes, bytes - off);
        const int b = i % sd_ctx::kStageBufs;
        SD_HIP(hipMemcpyAsync(ctx->stage[b], static_cast<const char*>(src) + off, n, hipMemcpyDeviceToHost, ctx->copy_stream));
        SD_HIP(hipEventRecord(ctx->stage_ev[b], ctx->copy_stream));
        return SD_OK;
    };
    for (int i = 0; i < nchunks && i < sd_ctx::kStageBufs - 1; ++i) SD_TRY(issue(i));
    for (int i = 0; i < nchunks; ++i) {
        const int next = i + sd_ctx::kStageBufs - 1;
        if (next < nchunks) SD_TRY(issue(next));  // its buffer was drained in iteration i - 1
        const size_t off = (size_t)i * sd_ctx::kStageBytes, n = std::min(sd_ctx::kStageBytes, bytes - off);
        const int b = i % sd_ctx::kStageBufs;
        SD_HIP(hipEventSynchronize(ctx->stage_ev[b]));
        parallel_memcpy(static_cast<char*>(dst) + off, ctx->stage[b], n);
    }
    return SD_OK;
}

extern "C" {

int sd_version(void) { return SD_VERSION; }

const char* sd_last_error(void) { return g_last_error.c_str(); }

int sd_device_count(int* count) {
    SD_CHECK_ARG(count, "sd_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return sd_set_error(SD_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return SD_OK;
}

int sd_ctx_create(int device, sd_ctx** out) {
    SD_CHECK_ARG(out, "sd_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    SD_HIP(hipGetDeviceCount(&n));
    SD_CHECK_ARG(device >= 0 && device < n, "sd_ctx_create: device %d out of range (have %d)", device, n);
    SD_HIP(hipSetDevice(device));
    sd_ctx* ctx = new sd_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    SD_HIP(hipGetDeviceProperties(&prop, device));
    ctx->cu_count = prop.multiProcessorCount;
    ctx->lds_max = prop.sharedMemPerBlock;
    if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ctx->lds_max = 160 * 1024;  // CDNA4: 160 KiB LDS per CU
    ctx->pool_cap = (size_t)prop.totalGlobalMem / 2;  // released on demand: sd_pool_malloc trims the cache when hipMalloc fails
    SD_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    SD_HIP(hipEventCreate(&ctx->t0));
    SD_HIP(hipEventCreate(&ctx->t1));
    SD_HIP(hipEventCreate(&ctx->p0));
    SD_HIP(hipEventCreate(&ctx->p1));
    *out = ctx;
    return SD_OK;
}

int sd_ctx_destroy(sd_ctx* ctx) {
    if (!ctx) return SD_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ctx->t0);
    (void)hipEventDestroy(ctx->t1);
    (void)hipEventDestroy(ctx->p0);
    (void)hipEventDestroy(ctx->p1);
    for (auto& pe : ctx->prof_pending) {
        (void)hipEventDestroy(pe.e0);
        (void)hipEventDestroy(pe.e1);
    }
    for (auto ev : ctx->prof_free) (void)hipEventDestroy(ev);
    if (ctx->prof_open) (void)hipEventDestroy(ctx->prof_open);
    if (ctx->ws_ptr) (void)hipFree(ctx->ws_ptr);
    for (int b = 0; b < sd_ctx::kStageBufs; ++b) {
        if (ctx->stage[b]) (void)hipHostFree(ctx->stage[b]);
        if (ctx->stage_ev[b]) (void)hipEventDestroy(ctx->stage_ev[b]);
    }
    if (ctx->stage_join) (void)hipEventDestroy(ctx->stage_join);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (int b = 0; b < sd_ctx::kDrainBufs; ++b) {
        if (ctx->drain[b]) (void)hipHostFree(ctx->drain[b]);
        if (ctx->drain_ev[b]) (void)hipEventDestroy(ctx->drain_ev[b]);
    }
    if (ctx->drain_stream) (void)hipStreamDestroy(ctx->drain_stream);
    sd_gt_cache_clear(ctx);
    sd_pool_trim(ctx);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SD_OK;
}

int sd_ctx_release_cached(sd_ctx* ctx) {
    SD_CHECK_ARG(ctx, "ctx is NULL");
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    sd_gt_cache_clear(ctx);
    sd_pool_trim(ctx);
    if (ctx->ws_ptr) {
        SD_HIP(hipFree(ctx->ws_ptr));
        ctx->ws_ptr = nullptr;
        ctx->ws_size = 0;
    }
    return SD_OK;
}

int sd_ctx_synchronize(sd_ctx* ctx) {
    SD_CHECK_ARG(ctx, "ctx is NULL");
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_ctx_device_info(sd_ctx* ctx, char* name, size_t name_len, int* compute_units, int64_t* hbm_bytes) {
    SD_CHECK_ARG(ctx, "ctx is NULL");
    hipDeviceProp_t prop;
    SD_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return SD_OK;
}

int sd_dev_alloc(sd_ctx* ctx, size_t bytes, void** dptr) {
    SD_CHECK_ARG(ctx && dptr, "sd_dev_alloc: NULL argument");
    *dptr = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    if (bytes == 0) bytes = 8;
    SD_HIP(hipMalloc(dptr, bytes));
    return SD_OK;
}

int sd_dev_free(sd_ctx* ctx, void* dptr) {
    SD_CHECK_ARG(ctx, "ctx is NULL");
    if (!dptr) return SD_OK;
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    SD_HIP(hipFree(dptr));
    return SD_OK;
}

int sd_memcpy_h2d(sd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    SD_CHECK_ARG(ctx && dst && src, "sd_memcpy_h2d: NULL argument");
    SD_HIP(hipSetDevice(ctx->device));
    SD_TRY(sd_copy_h2d(ctx, dst, src, bytes));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

int sd_memcpy_d2h(sd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    SD_CHECK_ARG(ctx && dst && src, "sd_memcpy_d2h: NULL argument");
    SD_HIP(hipSetDevice(ctx->device));
    SD_TRY(sd_copy_d2h(ctx, dst, src, bytes));
    return SD_OK;
}

namespace {
// float32 <-> float64 over n elements: 4 per thread through 16-byte loads / stores where the element count allows
__global__ void __launch_bounds__(256) widen_kernel(const float* __restrict__ src, int64_t n, double* __restrict__ dst) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        *reinterpret_cast<double2*>(dst + i) = make_double2((double)v.x, (double)v.y);
        *reinterpret_cast<double2*>(dst + i + 2) = make_double2((double)v.z, (double)v.w);
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = (double)src[j];
    }
}
__global__ void __launch_bounds__(256) narrow_kernel(const double* __restrict__ src, int64_t n, float* __restrict__ dst) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const double2 a = *reinterpret_cast<const double2*>(src + i), b = *reinterpret_cast<const double2*>(src + i + 2);
        *reinterpret_cast<float4*>(dst + i) = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = (float)src[j];
    }
}
}  // namespace

int sd_convert_f32_to_f64_dev(sd_ctx* ctx, const float* src_dev, int64_t n, double* dst_dev) {
    SD_CHECK_ARG(ctx && src_dev && dst_dev && n >= 0, "sd_convert_f32_to_f64_dev: bad argument");
    SD_CHECK_ARG((reinterpret_cast<uintptr_t>(src_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_dev) & 15) == 0,
                 "sd_convert_f32_to_f64_dev: 16-byte aligned device pointers expected");
    if (n == 0) return SD_OK;
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t nb = (n + 1023) / 1024;
    SD_CHECK_ARG(nb < ((int64_t)1 << 31), "sd_convert_f32_to_f64_dev: too many elements for one launch");
    SD_LAUNCH(ctx, "widen_kernel", widen_kernel, dim3((unsigned)nb), dim3(256), 0, src_dev, n, dst_dev);
    return SD_OK;
}

int sd_convert_f64_to_f32_dev(sd_ctx* ctx, const double* src_dev, int64_t n, float* dst_dev) {
    SD_CHECK_ARG(ctx && src_dev && dst_dev && n >= 0, "sd_convert_f64_to_f32_dev: bad argument");
    SD_CHECK_ARG((reinterpret_cast<uintptr_t>(src_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_dev) & 15) == 0,
                 "sd_convert_f64_to_f32_dev: 16-byte aligned device pointers expected");
    if (n == 0) return SD_OK;
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t nb = (n + 1023) / 1024;
    SD_CHECK_ARG(nb < ((int64_t)1 << 31), "sd_convert_f64_to_f32_dev: too many elements for one launch");
    SD_LAUNCH(ctx, "narrow_kernel", narrow_kernel, dim3((unsigned)nb), dim3(256), 0, src_dev, n, dst_dev);
    return SD_OK;
}

int sd_memcpy_d2d(sd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    SD_CHECK_ARG(ctx && dst && src, "sd_memcpy_d2d: NULL argument");
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SD_OK;
}

int sd_timer_start(sd_ctx* ctx) {
    SD_CHECK_ARG(ctx, "ctx is NULL");
    SD_HIP(hipEventRecord(ctx->t0, ctx->stream));
    return SD_OK;
}

int sd_timer_stop(sd_ctx* ctx, float* elapsed_ms) {
    SD_CHECK_ARG(ctx && elapsed_ms, "sd_timer_stop: NULL argument");
    SD_HIP(hipEventRecord(ctx->t1, ctx->stream));
    SD_HIP(hipEventSynchronize(ctx->t1));
    SD_HIP(hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
    return SD_OK;
}

// the pending event pairs -> per-kernel totals (waits for the stream once)
static int sd_prof_resolve(sd_ctx* ctx) {
    if (ctx->prof_pending.empty()) return SD_OK;
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipEventSynchronize(ctx->prof_pending.back().e1));
    for (auto& pe : ctx->prof_pending) {
        float ms = 0.f;
        SD_HIP(hipEventElapsedTime(&ms, pe.e0, pe.e1));
        auto& e = ctx->prof[pe.name];
        e.ms += ms;
        e.launches += 1;
        ctx->prof_free.push_back(pe.e0);
        ctx->prof_free.push_back(pe.e1);
    }
    ctx->prof_pending.clear();
    return SD_OK;
}

int sd_prof_enable(sd_ctx* ctx, int on) {
    SD_CHECK_ARG(ctx, "ctx is NULL");
    if (!on) SD_TRY(sd_prof_resolve(ctx));
    ctx->prof_on = on != 0;
    return SD_OK;
}

int sd_prof_reset(sd_ctx* ctx) {
    SD_CHECK_ARG(ctx, "ctx is NULL");
    SD_TRY(sd_prof_resolve(ctx));
    ctx->prof.clear();
    return SD_OK;
}

int sd_prof_query(sd_ctx* ctx, const char* kernel_name, double* total_ms, int64_t* launches) {
    SD_CHECK_ARG(ctx && kernel_name, "sd_prof_query: NULL argument");
    SD_TRY(sd_prof_resolve(ctx));
    auto it = ctx->prof.find(kernel_name);
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return SD_OK;
}

int sd_prof_names(sd_ctx* ctx, char* buf, size_t buf_len) {
    SD_CHECK_ARG(ctx && buf && buf_len, "sd_prof_names: NULL argument");
    SD_TRY(sd_prof_resolve(ctx));
    std::string s;
    for (auto& kv : ctx->prof) {
        if (!s.empty()) s += ";";
        s += kv.first;
    }
    snprintf(buf, buf_len, "%s", s.c_str());
    return SD_OK;
}

}  // extern "C"

hipError_t sd_pool_malloc(sd_ctx* ctx, void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    auto it = ctx->pool_free.find(bytes);
    if (it != ctx->pool_free.end()) {
        *p = it->second;
        ctx->pool_free.erase(it);
        ctx->pool_cached -= bytes;
    } else {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipErrorOutOfMemory && !ctx->pool_free.empty()) {  // give the cached blocks back and retry once
            (void)hipGetLastError();
            sd_pool_trim(ctx);
            e = hipMalloc(p, bytes);
        }
        if (e != hipSuccess) return e;
    }
    ctx->pool_live[*p] = bytes;
    return hipSuccess;
}

void sd_pool_release(sd_ctx* ctx, void* p) {
    if (!p) return;
    if (!ctx) {
        (void)hipFree(p);
        return;
    }
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) {  // not ours (imported pointer): plain free
        (void)hipFree(p);
        return;
    }
    const size_t bytes = it->second;
    ctx->pool_live.erase(it);
    if (ctx->pool_cached + bytes > ctx->pool_cap) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p);
        return;
    }
    ctx->pool_free.emplace(bytes, p);
    ctx->pool_cached += bytes;
}

void sd_pool_trim(sd_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    ctx->pool_free.clear();
    ctx->pool_cached = 0;
}

int sd_workspace(sd_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->ws_size) {
        SD_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->ws_ptr) SD_HIP(hipFree(ctx->ws_ptr));
        ctx->ws_ptr = nullptr;
        ctx->ws_size = 0;
        const size_t want = bytes + bytes / 8;
        SD_HIP(hipMalloc(&ctx->ws_ptr, want));
        ctx->ws_size = want;
    }
    *out = ctx->ws_ptr;
    return SD_OK;
}

static int sd_prof_event(sd_ctx* ctx, hipEvent_t* ev) {
    if (!ctx->prof_free.empty()) {
        *ev = ctx->prof_free.back();
        ctx->prof_free.pop_back();
        return SD_OK;
    }
    SD_HIP(hipEventCreate(ev));
    return SD_OK;
}

int sd_prof_begin(sd_ctx* ctx) {
    if (ctx->prof_on) {
        if (ctx->prof_pending.size() >= 8192) SD_TRY(sd_prof_resolve(ctx));  // (bounds the events alive: a wait every 8 192 launches)
        SD_TRY(sd_prof_event(ctx, &ctx->prof_open));
        SD_HIP(hipEventRecord(ctx->prof_open, ctx->stream));
    }
    return SD_OK;
}

int sd_prof_end(sd_ctx* ctx, const char* name) {
    if (ctx->prof_on && ctx->prof_open != nullptr) {
        hipEvent_t e1 = nullptr;
        SD_TRY(sd_prof_event(ctx, &e1));
        SD_HIP(hipEventRecord(e1, ctx->stream));
        ctx->prof_pending.push_back({ctx->prof_open, e1, name});
        ctx->prof_open = nullptr;
    }
    return SD_OK;
}

int sd_build_group_table(const int32_t* gid, int64_t T, int G, sd_group_table* out) {
    SD_CHECK_ARG(gid && out && G > 0 && T > 0, "group table: bad arguments");
    out->off.assign(G + 1, 0);
    for (int64_t t = 0; t < T; ++t) {
        SD_CHECK_ARG(gid[t] >= 0 && gid[t] < G, "group_id[%lld] = %d outside [0,%d)", (long long)t, gid[t], G);
        out->off[gid[t] + 1]++;
    }
    out->nmax = 0;
    for (int g = 0; g < G; ++g) {
        if (out->off[g + 1] > out->nmax) out->nmax = (int)out->off[g + 1];
        out->off[g + 1] += out->off[g];
    }
    out->order.resize(T);
    std::vector<int64_t> cur(out->off.begin(), out->off.end() - 1);
    for (int64_t t = 0; t < T; ++t) out->order[cur[gid[t]]++] = (int32_t)t;
    return SD_OK;
}

// ------------------------------------------------------------------------------------------------
// Synthetic fields (bit-identical mirror of skdownscale_amd/synth.py; compiled -ffp-contract=off)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t sd_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double sd_u01(uint64_t h0, uint64_t ctr4, int j) {
    const uint64_t x = sd_splitmix64(h0 ^ (ctr4 + (uint64_t)j));
    return (double)(x >> 11) * 0x1.0p-53;
}

__device__ __forceinline__ double sd_gauss(uint64_t h0, uint64_t ctr4) {
    const double u0 = sd_u01(h0, ctr4, 0), u1 = sd_u01(h0, ctr4, 1), u2 = sd_u01(h0, ctr4, 2), u3 = sd_u01(h0, ctr4, 3);
    return (((u0 + u1) + (u2 + u3)) - 2.0) * 1.7320508075688772;
}

__global__ void __launch_bounds__(256) sd_synth_kernel(double* __restrict__ out, int64_t T, int64_t C, int64_t ld,
                                                       int64_t c_offset, int64_t c_full, int kind, uint64_t h0,
                                                       uint64_t h0b, int has2, const double* __restrict__ base,
                                                       double amp, double cell_scale, double p_dry, double amp2) {
    const int64_t total = T * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / C;
        const int64_t cl = i - t * C;
        const uint64_t c = (uint64_t)(cl + c_offset);
        const uint64_t ctr4 = ((uint64_t)t * (uint64_t)c_full + c) * 4ull;
        double v;
        if (kind == SD_SYNTH_GAUSS) {
            const double g = sd_gauss(h0, ctr4);
            const double b = base ? base[t] : 0.0;
            const double off = cell_scale * (double)(c % 101ull);
            v = (b + off) + amp * g;
            if (has2) v = v + amp2 * sd_gauss(h0b, ctr4);
        } else {
            const double u0 = sd_u01(h0, ctr4, 0), u1 = sd_u01(h0, ctr4, 1), u2 = sd_u01(h0, ctr4, 2),
                         u3 = sd_u01(h0, ctr4, 3);
            const double wet = ((amp * u1) * u2) * u3;
            v = (u0 < p_dry) ? 0.0 : wet;
        }
        out[t * ld + cl] = v;
    }
}

static uint64_t host_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

extern "C" int sd_synth_fill(sd_ctx* ctx, double* out_dev, int64_t T, int64_t C, int64_t ld, int64_t c_offset,
                             int64_t c_full, int kind, uint64_t seed, uint32_t stream, const double* base_host,
                             double amp, double cell_scale, double p_dry, int32_t stream2, double amp2) {
    SD_CHECK_ARG(ctx && out_dev, "sd_synth_fill: NULL argument");
    SD_CHECK_ARG(T > 0 && C > 0 && ld >= C && c_full >= c_offset + C, "sd_synth_fill: bad sizes");
    SD_CHECK_ARG(kind == SD_SYNTH_GAUSS || kind == SD_SYNTH_PRECIP, "sd_synth_fill: unknown kind %d", kind);
    SD_HIP(hipSetDevice(ctx->device));
    const uint64_t h0 = host_splitmix64(seed ^ host_splitmix64((uint64_t)stream));
    const uint64_t h0b = stream2 >= 0 ? host_splitmix64(seed ^ host_splitmix64((uint64_t)stream2)) : 0;
    sd_scratch base;
    if (base_host) {
        SD_HIP(base.alloc(ctx, sizeof(double) * T));
        SD_HIP(hipMemcpyAsync(base.p, base_host, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
    }
    const int64_t total = T * C;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    SD_LAUNCH(ctx, "sd_synth_kernel", sd_synth_kernel, dim3((unsigned)blocks), dim3(256), 0, out_dev, T, C, ld, c_offset,
              c_full, kind, h0, h0b, stream2 >= 0 ? 1 : 0, base.as<double>(), amp, cell_scale, p_dry, amp2);
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SD_OK;
}
