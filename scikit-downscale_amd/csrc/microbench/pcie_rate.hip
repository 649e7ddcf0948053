// Host <-> device copy rates of the box: pinned buffers, one or two streams, each direction and both at once; pageable for
// comparison.  Decides how sd_copy_h2d / sd_copy_d2h (sd_ctx.hip) should drive the copies.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(e)                                                                 \
    do {                                                                      \
        hipError_t _e = (e);                                                  \
        if (_e != hipSuccess) {                                               \
            printf("%s failed: %s\n", #e, hipGetErrorString(_e));            \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t N = (size_t)1 << 30;  // 1 GiB per buffer
    char *h0, *h1, *d0, *d1;
    CK(hipHostMalloc(&h0, N, hipHostMallocDefault));
    CK(hipHostMalloc(&h1, N, hipHostMallocDefault));
    CK(hipMalloc(&d0, N));
    CK(hipMalloc(&d1, N));
    memset(h0, 1, N);
    memset(h1, 2, N);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    auto run = [&](const char* name, auto fn, double bytes) {
        fn();
        (void)hipDeviceSynchronize();
        const double t0 = now();
        for (int r = 0; r < 3; ++r) fn();
        (void)hipDeviceSynchronize();
        const double dt = (now() - t0) / 3;
        printf("%-44s %6.1f GB/s\n", name, bytes / dt / 1e9);
    };
    run("H2D pinned, 1 stream", [&] { (void)hipMemcpyAsync(d0, h0, N, hipMemcpyHostToDevice, s0); }, (double)N);
    run("H2D pinned, 2 streams (halves)", [&] {
        (void)hipMemcpyAsync(d0, h0, N / 2, hipMemcpyHostToDevice, s0);
        (void)hipMemcpyAsync(d0 + N / 2, h0 + N / 2, N / 2, hipMemcpyHostToDevice, s1);
    }, (double)N);
    run("H2D pinned, 64 MB chunks, 1 stream", [&] {
        for (size_t o = 0; o < N; o += (size_t)64 << 20) (void)hipMemcpyAsync(d0 + o, h0 + o, (size_t)64 << 20, hipMemcpyHostToDevice, s0);
    }, (double)N);
    run("D2H pinned, 1 stream", [&] { (void)hipMemcpyAsync(h1, d1, N, hipMemcpyDeviceToHost, s1); }, (double)N);
    run("H2D + D2H at once (2 streams)", [&] {
        (void)hipMemcpyAsync(d0, h0, N, hipMemcpyHostToDevice, s0);
        (void)hipMemcpyAsync(h1, d1, N, hipMemcpyDeviceToHost, s1);
    }, 2.0 * N);
    char* p = (char*)malloc(N);
    memset(p, 3, N);
    run("H2D pageable (hipMemcpy)", [&] { (void)hipMemcpy(d0, p, N, hipMemcpyHostToDevice); }, (double)N);
    run("D2H pageable (hipMemcpy)", [&] { (void)hipMemcpy(p, d1, N, hipMemcpyDeviceToHost); }, (double)N);
    // host memcpy rates: 1 thread and 16 threads, pageable -> pinned
    auto par = [&](char* dst, const char* src, unsigned nt) {
        std::vector<std::thread> th;
        const size_t part = N / nt;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back([=] { memcpy(dst + t * part, src + t * part, part); });
        for (auto& t : th) t.join();
    };
    for (unsigned nt : {1u, 4u, 8u, 16u, 32u}) {
        par(h0, p, nt);
        const double t0 = now();
        for (int r = 0; r < 3; ++r) par(h0, p, nt);
        printf("memcpy pageable -> pinned, %2u threads          %6.1f GB/s\n", nt, 3.0 * N / (now() - t0) / 1e9);
    }
    // first-touch cost of a result array: fresh pages, 4 KB vs transparent huge pages, thread counts
    {
        FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        char line[128] = "?";
        if (f) {
            if (!fgets(line, sizeof line, f)) line[0] = 0;
            fclose(f);
        }
        printf("transparent_hugepage/enabled: %s", line);
    }
    auto par_n = [&](char* dst, const char* src, unsigned nt) {
        std::vector<std::thread> th;
        const size_t part = ((N / nt) + 4095) & ~(size_t)4095;
        for (unsigned t = 0; t < nt; ++t) {
            const size_t off = t * part;
            if (off >= N) break;
            const size_t n = part < N - off ? part : N - off;
            th.emplace_back([=] { memcpy(dst + off, src + off, n); });
        }
        for (auto& t : th) t.join();
    };
    for (int huge = 0; huge < 2; ++huge)
        for (unsigned nt : {16u, 64u}) {
            char* fresh = (char*)aligned_alloc((size_t)2 << 20, N);
            if (huge) madvise(fresh, N, MADV_HUGEPAGE);
            const double t0 = now();
            par_n(fresh, h1, nt);
            const double dt = now() - t0;
            const double t1 = now();
            par_n(fresh, h1, nt);
            printf("memcpy pinned -> fresh pages%s, %2u threads   %6.1f GB/s   (touched: %6.1f GB/s)\n", huge ? " (MADV_HUGEPAGE)" : "                ",
                   nt, (double)N / dt / 1e9, (double)N / (now() - t1) / 1e9);
            free(fresh);
        }
    return 0;
}
