// LDS bisection of a sorted float64 series by many random queries, the way analog_f1_fused_kernel / analog_f1_mean3_kernel search
// (one 1 024-thread workgroup per CU, the series resident in LDS, 16 queries per thread, two queries in flight per thread):
//   A  ds_read_b64 of the values (what the kernels do)
//   B  the values kept as sortable 64-bit integers in the same array; the bisection reads only their upper words (ds_read_b32 at
//      byte offset 4), equal upper words are resolved by an exact forward scan on the full words
//   C  upper and lower words in two separate arrays (SoA), bisection on the array of upper words, same exact scan
// All three return the lower bound (number of values < q) of every query, checked against std::lower_bound on the host.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 csrc/microbench/search_test.hip -o csrc/microbench/search_test
//   csrc/microbench/search_test [n = 14600] [reps = 20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

constexpr int kQ = 16;  // queries per thread

__host__ __device__ inline unsigned long long sortable(double v) {
    unsigned long long b;
#ifdef __HIP_DEVICE_COMPILE__
    b = (unsigned long long)__double_as_longlong(v);
#else
    memcpy(&b, &v, 8);
#endif
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// strides of the branch-free bisection (the kernels shorten strides that are multiples of 16 doubles by one)
__device__ __forceinline__ int next_half(int& len) {
    int half = len >> 1;
    if ((half & 15) == 0) --half;
    len -= half;
    return half;
}

template <int MODE>
__global__ void __launch_bounds__(1024) search_kernel(const double* __restrict__ xs, int n, const double* __restrict__ q, int nq, int reps,
                                                      int* __restrict__ out, unsigned long long* __restrict__ clocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* buf = reinterpret_cast<double*>(smem);                              // A: n + 1 doubles
    unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);      // B: n + 1 sortable words
    unsigned* hi = reinterpret_cast<unsigned*>(smem);                           // C: n + 1 upper words, then n + 1 lower words
    unsigned* lo = hi + (n + 2);
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i <= n; i += nthr) {
        const double v = i < n ? xs[i] : __longlong_as_double(0x7ff0000000000000ll);
        if (MODE == 0) buf[i] = v;
        if (MODE == 1) key[i] = sortable(v);
        if (MODE == 2) {
            const unsigned long long s = sortable(v);
            hi[i] = (unsigned)(s >> 32);
            lo[i] = (unsigned)s;
        }
    }
    double qv[kQ];
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
        const int j = tid + i * nthr;
        qv[i] = j < nq ? q[(size_t)blockIdx.x * 0 + j] : 0.0;
    }
    __syncthreads();
    int res[kQ];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int i0 = 0; i0 < kQ; i0 += 2) {
            int pos[2] = {-1, -1};
            if (MODE == 0) {
                const double a = qv[i0], b = qv[i0 + 1];
#pragma unroll 1
                for (int len = n; len > 1;) {
                    const int half = next_half(len);
                    pos[0] += buf[pos[0] + half] < a ? half : 0;
                    pos[1] += buf[pos[1] + half] < b ? half : 0;
                }
                res[i0] = pos[0] + 1 + (buf[pos[0] + 1] < a ? 1 : 0);
                res[i0 + 1] = pos[1] + 1 + (buf[pos[1] + 1] < b ? 1 : 0);
            } else {
                const unsigned long long sa = sortable(qv[i0]), sb = sortable(qv[i0 + 1]);
                const unsigned ha = (unsigned)(sa >> 32), hb = (unsigned)(sb >> 32);
                const unsigned* hw = MODE == 1 ? reinterpret_cast<const unsigned*>(key) + 1 : hi;  // upper word of element i at hw[i * st]
                constexpr int st = MODE == 1 ? 2 : 1;
#pragma unroll 1
                for (int len = n; len > 1;) {
                    const int half = next_half(len);
                    pos[0] += hw[(pos[0] + half) * st] < ha ? half : 0;
                    pos[1] += hw[(pos[1] + half) * st] < hb ? half : 0;
                }
                int p0 = pos[0] + 1 + (hw[(pos[0] + 1) * st] < ha ? 1 : 0);
                int p1 = pos[1] + 1 + (hw[(pos[1] + 1) * st] < hb ? 1 : 0);
                // p = number of values whose upper word is below the query's; values with an equal upper word: exact scan
                if (MODE == 1) {
                    while (key[p0] < sa) ++p0;  // (key[n] = sortable(+inf): the scan stops)
                    while (key[p1] < sb) ++p1;
                } else {
                    const unsigned la = (unsigned)sa, lb = (unsigned)sb;
                    while (hi[p0] < ha || (hi[p0] == ha && lo[p0] < la)) ++p0;
                    while (hi[p1] < hb || (hi[p1] == hb && lo[p1] < lb)) ++p1;
                }
                res[i0] = p0;
                res[i0 + 1] = p1;
            }
        }
        if (rep + 1 < reps) {
#pragma unroll
            for (int i = 0; i < kQ; ++i) qv[i] += (double)(res[i] >> 20);  // (always 0) keeps the repetitions apart
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = tid + i * nthr;
            if (j < nq) out[j] = res[i];
        }
    }
}

// ---- window start of the k nearest values (analog_f1_mean3_kernel, generation 1) after the lower bound p:
//   R0  what the kernels do: 5 bisection steps over the k + 1 candidates p - k .. p on  (q - x[L])^2 > (q - x[L + k])^2
//   R1  fixed strides 16, 8, 4, 2, 1 from p - k on  x[L] + x[L + k] < 2 q  (one addition and one compare instead of two subtractions,
//       two multiplications and a compare; the kernels' exact separation check afterwards decides whether a window is accepted, so
//       the start only has to be right, not derived the same way)
__device__ __forceinline__ double sqd(double q, double x) {
    const double d = q - x;
    return d * d;
}
template <int R>
__global__ void __launch_bounds__(1024) refine_kernel(const double* __restrict__ xs, int n, const double* __restrict__ q, int nq, int k, int reps,
                                                      int* __restrict__ out, unsigned long long* __restrict__ clocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* buf = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i <= n; i += nthr) buf[i] = i < n ? xs[i] : __longlong_as_double(0x7ff0000000000000ll);
    double qv[kQ];
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
        const int j = tid + i * nthr;
        qv[i] = j < nq ? q[j] : 0.0;
    }
    __syncthreads();
    const int M = n - k > 0 ? n - k : 0;
    int nsteps = 0;
    while ((1 << nsteps) < (k + 1 < M + 1 ? k + 1 : M + 1)) ++nsteps;
    int res[kQ];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int i0 = 0; i0 < kQ; i0 += 2) {
            int pos[2] = {-1, -1};
            double a[2] = {qv[i0], qv[i0 + 1]};
#pragma unroll 1
            for (int len = n; len > 1;) {
                const int half = next_half(len);
#pragma unroll
                for (int j = 0; j < 2; ++j) pos[j] += buf[pos[j] + half] < a[j] ? half : 0;
            }
            int lo[2], hi[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = pos[j] + 1 + (buf[pos[j] + 1] < a[j] ? 1 : 0);
                lo[j] = p - k > 0 ? p - k : 0;
                hi[j] = p < M ? p : M;
            }
            if (R == 0) {
#pragma unroll 1
                for (int s = 0; s < nsteps; ++s) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int mid = (lo[j] + hi[j]) >> 1;
                        const bool act = lo[j] < hi[j];
                        const bool right = sqd(a[j], buf[mid]) > sqd(a[j], buf[mid + k]);
                        lo[j] = (act && right) ? mid + 1 : lo[j];
                        hi[j] = (act && !right) ? mid : hi[j];
                    }
                }
            } else {
                int ps[2] = {lo[0] - 1, lo[1] - 1};  // last start known to lie left of the window
                const double a2[2] = {a[0] + a[0], a[1] + a[1]};
#pragma unroll 1
                for (int half = 1 << (nsteps - 1); half >= 1; half >>= 1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int idx = ps[j] + half;
                        const int ic = idx < M ? idx : M;
                        const bool right = idx < hi[j] && (buf[ic] + buf[ic + k]) < a2[j];
                        ps[j] += right ? half : 0;
                    }
                }
                lo[0] = ps[0] + 1;
                lo[1] = ps[1] + 1;
            }
            res[i0] = lo[0];
            res[i0 + 1] = lo[1];
        }
        if (rep + 1 < reps) {
#pragma unroll
            for (int i = 0; i < kQ; ++i) qv[i] += (double)(res[i] >> 20);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = tid + i * nthr;
            if (j < nq) out[j] = res[i];
        }
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 14600, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nq = 1024 * kQ < n ? 1024 * kQ : n;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> g(0.0, 1.0);
    std::vector<double> xs(n), q(nq);
    for (auto& v : xs) v = g(rng);
    for (int i = 0; i < 40 && i + 1 < n; i += 2) xs[i + 1] = xs[i];                    // exact ties
    for (int i = 100; i < 140 && i + 1 < n; i += 2) xs[i + 1] = std::nextafter(xs[i], 10.0);  // values one ulp apart (equal upper words)
    std::sort(xs.begin(), xs.end());
    for (auto& v : q) v = 1.1 * g(rng);
    for (int i = 0; i < 64 && i < nq; ++i) q[i] = xs[(size_t)i * 97 % n];  // queries that hit values exactly
    q[nq - 1] = 100.0;
    q[nq - 2] = -100.0;
    std::vector<int> expect(nq);
    for (int i = 0; i < nq; ++i) expect[i] = (int)(std::lower_bound(xs.begin(), xs.end(), q[i]) - xs.begin());
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int nb = prop.multiProcessorCount;
    double *dxs, *dq;
    int* dout;
    unsigned long long* dclk;
    CHECK(hipMalloc(&dxs, sizeof(double) * n));
    CHECK(hipMalloc(&dq, sizeof(double) * nq));
    CHECK(hipMalloc(&dout, sizeof(int) * nq));
    CHECK(hipMalloc(&dclk, sizeof(unsigned long long) * nb));
    CHECK(hipMemcpy(dxs, xs.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dq, q.data(), sizeof(double) * nq, hipMemcpyHostToDevice));
    const size_t lds = sizeof(double) * (size_t)(n + 2) + 16;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&search_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&search_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&search_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char* names[3] = {"A  ds_read_b64 of the values", "B  upper words of sortable 64-bit keys (same array, ds_read_b32)",
                            "C  upper / lower words in separate arrays"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 3; ++mode) {
            CHECK(hipMemset(dout, 0xff, sizeof(int) * nq));
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(search_kernel<0>, dim3(nb), dim3(1024), lds, 0, dxs, n, dq, nq, reps, dout, dclk);
            if (mode == 1) hipLaunchKernelGGL(search_kernel<1>, dim3(nb), dim3(1024), lds, 0, dxs, n, dq, nq, reps, dout, dclk);
            if (mode == 2) hipLaunchKernelGGL(search_kernel<2>, dim3(nb), dim3(1024), lds, 0, dxs, n, dq, nq, reps, dout, dclk);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<int> got(nq);
            std::vector<unsigned long long> clk(nb);
            CHECK(hipMemcpy(got.data(), dout, sizeof(int) * nq, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(clk.data(), dclk, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < nq; ++i) bad += got[i] != expect[i];
            double mean = 0;
            for (auto c : clk) mean += (double)c;
            mean /= nb;
            if (round == 1)
                printf("%-70s %s  %9.0f clocks per pass over %d queries (wave 0 of a workgroup, mean of %d workgroups), kernel %.3f ms for %d passes\n",
                       names[mode], bad ? "WRONG" : "ok   ", mean / reps, nq, nb, ms, reps);
            if (bad) printf("   %d of %d lower bounds differ\n", bad, nq);
        }
    // ---- lower bound + window start (k = 30)
    const int k = 30;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<int> l0(nq), l1(nq);
    const char* rnames[2] = {"R0 lower bound + window start by squared distances (the kernels)", "R1 lower bound + window start by x[L] + x[L + k] < 2 q, fixed strides"};
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 2; ++mode) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(refine_kernel<0>, dim3(nb), dim3(1024), lds, 0, dxs, n, dq, nq, k, reps, dout, dclk);
            if (mode == 1) hipLaunchKernelGGL(refine_kernel<1>, dim3(nb), dim3(1024), lds, 0, dxs, n, dq, nq, k, reps, dout, dclk);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> clk(nb);
            CHECK(hipMemcpy((mode == 0 ? l0 : l1).data(), dout, sizeof(int) * nq, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(clk.data(), dclk, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost));
            double mean = 0;
            for (auto c : clk) mean += (double)c;
            mean /= nb;
            if (round == 1) printf("%-75s %9.0f clocks per pass, kernel %.3f ms for %d passes\n", rnames[mode], mean / reps, ms, reps);
        }
    int diff = 0;
    for (int i = 0; i < nq; ++i) diff += l0[i] != l1[i];
    printf("window starts that differ between R0 and R1: %d of %d (they go to the separation check / exact walk either way)\n", diff, nq);
    return 0;
}
