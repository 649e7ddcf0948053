// Microbenchmark (round 6, review item 8): a caller's pageable field moved to the device
//   (a) through the engine's scheme -- a ring of pinned staging buffers, host threads copying chunk i + 1 while the DMA engine moves chunk i
//       (sd_ctx.hip: sd_copy_h2d) --,
//   (b) by registering the caller's pages (hipHostRegister) and one hipMemcpyAsync straight out of them, registration and
//       unregistration inside the time (a one-shot fit / predict call sees both) and outside it (a caller that keeps its arrays),
//   (c) by the runtime's own pageable path (plain hipMemcpy).
// Same for the way back (device -> a pageable result array).  Contiguous fields and column blocks (rows of `width` bytes at a pitch).
// Build: hipcc -O3 --offload-arch=gfx950 host_register.hip -o host_register -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void parallel_memcpy(void* dst, const void* src, size_t bytes, unsigned nt) {
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

int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 2;
    const size_t bytes = gb << 30;
    const size_t kStage = (size_t)64 << 20;
    unsigned nt = std::thread::hardware_concurrency() / 8;
    nt = nt < 2 ? 2 : (nt > 16 ? 16 : nt);
    printf("field of %zu GiB, %u copy threads, staging chunks of %zu MiB\n", gb, nt, kStage >> 20);
    char* host = static_cast<char*>(aligned_alloc(4096, bytes));
    memset(host, 1, bytes);  // touched: resident pages, as a caller's array is
    char* back = static_cast<char*>(aligned_alloc(4096, bytes));
    memset(back, 0, bytes);
    void* dev;
    CK(hipMalloc(&dev, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    void* stage[3];
    hipEvent_t ev[3];
    for (int b = 0; b < 3; ++b) {
        CK(hipHostMalloc(&stage[b], kStage, hipHostMallocDefault));
        CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
    }
    auto staged_h2d = [&]() {
        size_t off = 0;
        for (int i = 0; off < bytes; ++i) {
            const int b = i % 3;
            const size_t n = std::min(kStage, bytes - off);
            if (i >= 3) CK(hipEventSynchronize(ev[b]));
            parallel_memcpy(stage[b], host + off, n, nt);
            CK(hipMemcpyAsync(static_cast<char*>(dev) + off, stage[b], n, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[b], s));
            off += n;
        }
        CK(hipStreamSynchronize(s));
    };
    auto staged_d2h = [&]() {
        const int nch = (int)((bytes + kStage - 1) / kStage);
        auto issue = [&](int i) {
            const size_t off = (size_t)i * kStage, n = std::min(kStage, bytes - off);
            CK(hipMemcpyAsync(stage[i % 3], static_cast<char*>(dev) + off, n, hipMemcpyDeviceToHost, s));
            CK(hipEventRecord(ev[i % 3], s));
        };
        for (int i = 0; i < nch && i < 2; ++i) issue(i);
        for (int i = 0; i < nch; ++i) {
            if (i + 2 < nch) issue(i + 2);
            const size_t off = (size_t)i * kStage, n = std::min(kStage, bytes - off);
            CK(hipEventSynchronize(ev[i % 3]));
            parallel_memcpy(back + off, stage[i % 3], n, nt);
        }
    };
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        staged_h2d();
        double t1 = now();
        printf("h2d staged ring                       %7.1f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        t0 = now();
        CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
        double tr = now();
        CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double tc = now();
        CK(hipHostUnregister(host));
        t1 = now();
        printf("h2d register + copy + unregister      %7.1f ms  %5.1f GB/s   (register %.1f ms, copy %.1f ms = %.1f GB/s, unregister %.1f ms)\n",
               (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (tr - t0) * 1e3, (tc - tr) * 1e3, bytes / (tc - tr) / 1e9, (t1 - tc) * 1e3);
        t0 = now();
        CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        t1 = now();
        printf("h2d runtime pageable path             %7.1f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        t0 = now();
        staged_d2h();
        t1 = now();
        printf("d2h staged ring                       %7.1f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        t0 = now();
        CK(hipHostRegister(back, bytes, hipHostRegisterDefault));
        tr = now();
        CK(hipMemcpyAsync(back, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        tc = now();
        CK(hipHostUnregister(back));
        t1 = now();
        printf("d2h register + copy + unregister      %7.1f ms  %5.1f GB/s   (register %.1f ms, copy %.1f ms = %.1f GB/s, unregister %.1f ms)\n",
               (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (tr - t0) * 1e3, (tc - tr) * 1e3, bytes / (tc - tr) / 1e9, (t1 - tc) * 1e3);
        t0 = now();
        CK(hipMemcpy(back, dev, bytes, hipMemcpyDeviceToHost));
        t1 = now();
        printf("d2h runtime pageable path             %7.1f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    // both directions at once out of registered memory (what a cell-blocked predict could overlap): the link's duplex rate
    CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    CK(hipHostRegister(back, bytes, hipHostRegisterDefault));
    void* dev2;
    CK(hipMalloc(&dev2, bytes));
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        CK(hipMemcpyAsync(back, dev2, bytes, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s));
        CK(hipStreamSynchronize(s2));
        double t1 = now();
        printf("registered, h2d and d2h concurrently  %7.1f ms  %5.1f GB/s per direction\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    // column block out of a registered field: 2-D copy of rows of `width` bytes at pitch 8 * width
    {
        const size_t width = 100000, pitch = 8 * width, rows = bytes / pitch;
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CK(hipMemcpy2DAsync(dev, width, host, pitch, width, rows, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            double t1 = now();
            printf("registered, 2-D h2d of a column block %7.1f ms  %5.1f GB/s (%zu rows of %zu B at pitch %zu)\n", (t1 - t0) * 1e3,
                   width * rows / (t1 - t0) / 1e9, rows, width, pitch);
        }
    }
    return 0;
}
