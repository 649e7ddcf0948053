// Correctness + throughput check of the register wave sort (csrc/sd_wsort.h) on random keys.
//   hipcc -O3 --offload-arch=gfx950 csrc/microbench/wsort_test.hip -o /tmp/wsort_test && /tmp/wsort_test
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sd_wsort.h"

template <int K>
__global__ void __launch_bounds__(512, 4) sort_kernel(unsigned* keys, int nseg, int reps, int lanes_used) {
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.x * 8 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    unsigned* base = keys + (size_t)seg * 64 * K + lane * K;
    unsigned k[K];
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = base[i];
    for (int r = 0; r < reps; ++r) {
        if (r > 0) {
            // scramble again (bijection on the keys, so they stay distinct)
#pragma unroll
            for (int i = 0; i < K; ++i) k[i] = (k[i] * 2654435761u) ^ 0x5bd1e995u;  // (timing loop only)
        }
        sdws::wave_sort<K>(k, lane, lanes_used);
    }
#pragma unroll
    for (int i = 0; i < K; ++i) base[i] = k[i];
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

template <int K>
int run(int nseg, int reps) {
    const size_t n = (size_t)nseg * 64 * K;
    std::vector<unsigned> h(n), ref(n), out(n);
    unsigned s = 12345u + K;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((s >> 4) ^ (unsigned)(i * 2654435761u)) & 0x7fffffffu;  // below the pads of the partly used waves
    }
    unsigned* d;
    CK(hipMalloc(&d, n * 4));
    int bad = 0;
    for (int used : {64, 62, 33, 32, 17, 5, 1}) {
        std::vector<unsigned> hh = h;
        // lanes >= used hold pads (maximal keys)
        for (int sg = 0; sg < nseg; ++sg)
            for (int j = used * K; j < 64 * K; ++j) hh[(size_t)sg * 64 * K + j] = 0xfff00000u + (unsigned)j;
        CK(hipMemcpy(d, hh.data(), n * 4, hipMemcpyHostToDevice));
        sort_kernel<K><<<(nseg + 7) / 8, 512>>>(d, nseg, 1, used);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), d, n * 4, hipMemcpyDeviceToHost));
        ref = hh;
        for (int sg = 0; sg < nseg; ++sg) std::sort(ref.begin() + (size_t)sg * 64 * K, ref.begin() + (size_t)(sg + 1) * 64 * K);
        size_t nb = 0;
        for (int sg = 0; sg < nseg; ++sg)
            for (int j = 0; j < used * K; ++j) nb += out[(size_t)sg * 64 * K + j] != ref[(size_t)sg * 64 * K + j];
        printf("K=%d lanes_used=%d: %zu mismatches of %zu\n", K, used, nb, (size_t)nseg * used * K);
        bad += nb != 0;
    }
    // throughput: reps sorts per segment
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    sort_kernel<K><<<(nseg + 7) / 8, 512>>>(d, nseg, reps, 64);
    CK(hipEventRecord(e0));
    sort_kernel<K><<<(nseg + 7) / 8, 512>>>(d, nseg, reps, 64);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double sorts = (double)nseg * reps;
    printf("K=%d: %d segments x %d sorts in %.3f ms -> %.2f ns per wave-sort per SIMD-slot (1024 SIMDs), %.1f M sorts/s\n", K,
           nseg, reps, ms, ms * 1e6 / (sorts / 1024.0), sorts / ms / 1e3);
    CK(hipFree(d));
    return bad;
}

int main(int argc, char** argv) {
    const int nseg = argc > 1 ? atoi(argv[1]) : 65536;
    const int reps = argc > 2 ? atoi(argv[2]) : 16;
    int bad = 0;
    bad += run<4>(nseg, reps);
    bad += run<12>(nseg, reps);
    bad += run<16>(nseg, reps);
    bad += run<20>(nseg, reps);
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
