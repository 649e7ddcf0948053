// Correctness + throughput check of the counting rank (csrc/sd_wrank.h) on quantised Gaussian / uniform / log-normal keys.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 csrc/microbench/wrank_test.hip -o csrc/microbench/wrank_test
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define SD_WRANK_STAMPS
#include "sd_wrank.h"

constexpr int kWsBytes = sdwr::kHistBytes + 4 * (64 * 20 + sdwr::kMaxBin + 4);

template <int K, int T>
__global__ void __launch_bounds__(512, 4) rank_kernel(const unsigned* keys, unsigned* ranks, int* info, int nseg, int n, int reps, unsigned long long* phase) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = blockIdx.x * 8 + wave;
    if (seg >= nseg) return;
    const unsigned ws = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * kWsBytes;
    const unsigned* base = keys + (size_t)seg * 64 * K + lane * K;
    unsigned q[K], r[K];
#pragma unroll
    for (int i = 0; i < K; ++i) q[i] = base[i];
    bool ok = true;
    int cmax = 0;
    unsigned long long st[7], acc[6] = {0, 0, 0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
        cmax = sdwr::wave_rank<K, T>(q, n, lane, ws, r, &ok, st);
        for (int k = 0; k < 6; ++k) acc[k] += st[k + 1] - st[k];
        if (rep + 1 < reps) {
#pragma unroll
            for (int i = 0; i < K; ++i) q[i] += r[i] >> 12;  // (always 0) keeps the repetitions from being merged
        }
    }
    unsigned* out = ranks + (size_t)seg * 64 * K + lane * K;
#pragma unroll
    for (int i = 0; i < K; ++i) out[i] = r[i];
    if (lane == 0) {
        if (phase != nullptr)
            for (int k = 0; k < 6; ++k) atomicAdd(&phase[k], acc[k]);
        info[2 * seg] = cmax;
        info[2 * seg + 1] = ok ? 1 : 0;
    }
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

template <int K, int T>
int run(int nseg, int n, int reps, int dist) {
    const size_t tot = (size_t)nseg * 64 * K;
    std::vector<unsigned> h(tot, 0xffffffffu), out(tot);
    std::mt19937_64 rng(1234 + dist);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::uniform_real_distribution<double> ud(0.0, 1.0);
    for (int s = 0; s < nseg; ++s) {
        std::vector<double> v(n);
        double lo = 1e300, hi = -1e300, sum = 0, sq = 0;
        for (int j = 0; j < n; ++j) {
            v[j] = dist == 0 ? nd(rng) : dist == 1 ? ud(rng) : std::exp(1.5 * nd(rng));
            lo = std::min(lo, v[j]);
            hi = std::max(hi, v[j]);
            sum += v[j];
            sq += v[j] * v[j];
        }
        const double mean = sum / n, sd = std::sqrt(std::max(sq / n - mean * mean, 0.0));
        // clamp at mean +- 4.25 sd (as the kernels do), affine to 32 bits
        const double a = std::max(lo, mean - 4.25 * sd), b = std::min(hi, mean + 4.25 * sd);
        const double scale = 4294967295.0 / (b - a);
        for (int j = 0; j < n; ++j) {
            double t = (v[j] - a) * scale;
            t = t < 0 ? 0 : (t > 4294967295.0 ? 4294967295.0 : t);
            h[(size_t)s * 64 * K + j] = (unsigned)t;
        }
    }
    unsigned *dk, *dr;
    int* di;
    CK(hipMalloc(&dk, tot * 4));
    CK(hipMalloc(&dr, tot * 4));
    CK(hipMalloc(&di, nseg * 8));
    CK(hipMemcpy(dk, h.data(), tot * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)8 * kWsBytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rank_kernel<K, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    rank_kernel<K, T><<<(nseg + 7) / 8, 512, lds>>>(dk, dr, di, nseg, n, 1, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dr, tot * 4, hipMemcpyDeviceToHost));
    std::vector<int> info(2 * nseg);
    CK(hipMemcpy(info.data(), di, nseg * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, notok = 0, dup = 0;
    long cmaxsum = 0;
    int cmaxmax = 0;
    for (int s = 0; s < nseg; ++s) {
        const unsigned* k = &h[(size_t)s * 64 * K];
        std::vector<unsigned> srt(k, k + n);
        std::sort(srt.begin(), srt.end());
        bool has_dup = false;
        for (int j = 0; j + 1 < n; ++j) has_dup |= srt[j] == srt[j + 1];
        dup += has_dup;
        notok += info[2 * s + 1] == 0;
        cmaxsum += info[2 * s];
        cmaxmax = std::max(cmaxmax, info[2 * s]);
        if (info[2 * s] > sdwr::kMaxBin) continue;
        if (has_dup != (info[2 * s + 1] == 0)) ++bad;
        if (has_dup) continue;
        for (int j = 0; j < n; ++j) {
            const unsigned want = (unsigned)(std::lower_bound(srt.begin(), srt.end(), k[j]) - srt.begin());
            bad += out[(size_t)s * 64 * K + j] != want;
        }
    }
    printf("K=%d T=%d n=%d dist=%d: %zu errors; segments with equal keys %zu (flagged %zu); cmax mean %.2f max %d\n", K, T, n, dist, bad, dup,
           notok, (double)cmaxsum / nseg, cmaxmax);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long* dph;
    CK(hipMalloc(&dph, 48));
    CK(hipMemset(dph, 0, 48));
    rank_kernel<K, T><<<(nseg + 7) / 8, 512, lds>>>(dk, dr, di, nseg, n, reps, nullptr);
    CK(hipEventRecord(e0));
    rank_kernel<K, T><<<(nseg + 7) / 8, 512, lds>>>(dk, dr, di, nseg, n, reps, dph);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double sorts = (double)nseg * reps;
    printf("    %d segments x %d rankings in %.3f ms -> %.0f ns per wave-ranking per SIMD (1024 SIMDs)\n", nseg, reps, ms,
           ms * 1e6 / (sorts / 1024.0));
    unsigned long long ph[6];
    CK(hipMemcpy(ph, dph, 48, hipMemcpyDeviceToHost));
    printf("    wave clocks per ranking by phase A..F:");
    for (int k = 0; k < 6; ++k) printf(" %.0f", (double)ph[k] / sorts);
    printf("\n");
    CK(hipFree(dk));
    CK(hipFree(dr));
    CK(hipFree(di));
    return bad != 0;
}

int main(int argc, char** argv) {
    const int nseg = argc > 1 ? atoi(argv[1]) : 32768;
    const int reps = argc > 2 ? atoi(argv[2]) : 16;
    int bad = 0;
    for (int dist = 0; dist < 3; ++dist) {
        bad += run<20, 8>(nseg, 1240, reps, dist);
        if (dist < 2) bad += run<20, 4>(nseg, 1240, reps, dist);
    }
    bad += run<20, 8>(nseg, 1130, reps, 0);
    bad += run<20, 8>(nseg, 77, reps, 0);
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
