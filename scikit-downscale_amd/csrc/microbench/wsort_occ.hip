// Microbenchmark (round 6): what a wave of the fused BCSD kernel gets out of its SIMD when few waves share it.
//
// The fused kernel runs two workgroups of 8 waves per CU = 4 waves per SIMD, but the two workgroups alternate between tile
// waits and arithmetic, so most of the time a SIMD has the 2 waves of ONE workgroup to issue from.  This test times the
// register sort (sd_wsort.h, K = 20: 1 420 vector + 120 LDS-crossbar instructions per wave) at 1, 2, 4 (and 8) resident waves
// per SIMD -- occupancy set by dynamic LDS -- in two code shapes:
//   loop      the sort in a loop (11 KB of code, instruction-cache resident)
//   straight  REP back-to-back copies of the sort (REP x 11 KB of straight-line code: what the 50 KB fused kernel looks like to
//             the instruction cache, which two CUs share)
// Prints ns per sort per SIMD (= per-SIMD throughput) and ns per sort per wave (= latency a wave sees).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sd_wsort.h"

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e = (x);                                                                         \
        if (e != hipSuccess) {                                                                      \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);            \
            exit(1);                                                                                \
        }                                                                                           \
    } while (0)

constexpr int K = 20;

template <int REP>
__global__ void __launch_bounds__(256, 1) sort_kernel(unsigned* keys, int loops) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    unsigned* base = keys + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 * K + lane * K;
    unsigned k[K];
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = base[i];
    for (int r = 0; r < loops; ++r) {
#pragma unroll
        for (int c = 0; c < REP; ++c) {
#pragma unroll
            for (int i = 0; i < K; ++i) k[i] = (k[i] * 2654435761u) ^ (0x5bd1e995u + (unsigned)c);  // scramble (a bijection: keys stay distinct)
            sdws::wave_sort<K>(k, lane, 64);
        }
    }
#pragma unroll
    for (int i = 0; i < K; ++i) base[i] = k[i];
    if (smem[threadIdx.x] == 123 && loops < 0) base[0] = 0;
}

template <int REP>
void run(unsigned* d, int waves_per_simd, int sorts_per_wave) {
    // one 256-thread workgroup = one wave per SIMD; dynamic LDS limits the workgroups per CU
    const size_t lds = waves_per_simd >= 8 ? 160 * 1024 / 8 : 160 * 1024 / waves_per_simd - (waves_per_simd > 1 ? 256 : 0);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_kernel<REP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = 256 * waves_per_simd * 4;  // four generations of workgroups
    const int loops = sorts_per_wave / REP;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(sort_kernel<REP>, dim3(blocks), dim3(256), lds, 0, d, loops);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    const double sorts = (double)blocks * 4 * loops * REP;
    const double per_simd = best * 1e6 / (sorts / 1024.0);
    printf("  %d waves/SIMD, %-8s (%3d KB of code): %8.1f ns per sort per SIMD, %8.1f ns per sort for a wave\n", waves_per_simd,
           REP == 1 ? "loop" : "straight", REP * 11, per_simd, per_simd * waves_per_simd);
}

int main() {
    const size_t n = (size_t)256 * 8 * 4 * 4 * 64 * K;
    std::vector<unsigned> h(n);
    unsigned s = 777u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((s >> 4) ^ (unsigned)(i * 2654435761u)) & 0x7fffffffu;
    }
    unsigned* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    printf("register sort of 64 x %d keys (sd_wsort.h), by resident waves per SIMD and code shape\n", K);
    for (int w : {1, 2, 4, 8}) {
        run<1>(d, w, 48);
        run<4>(d, w, 48);
        run<12>(d, w, 48);
    }
    return 0;
}
