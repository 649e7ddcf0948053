// Microbenchmark: HBM bandwidth of the BCSD access pattern -- a workgroup reads all rows of one month
// (~1240 rows scattered in ~31-row runs over 14600) for W adjacent cells of a [T, C] f64 field.
// Prints GB/s per (W, mapping) so the kernel's tile width can be chosen from data, not guesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int W, int VEC>
__global__ void __launch_bounds__(512) tile_read(const double* __restrict__ x, int64_t ld, const int* __restrict__ order,
                                                 const int* __restrict__ goff, int G, int64_t C, double* sink, int mapping) {
    const int64_t ntiles = C / W;
    int64_t tile; int g;
    const int64_t b = blockIdx.x;
    if (mapping == 0) { tile = b % ntiles; g = (int)(b / ntiles); }            // month-major
    else if (mapping == 1) { g = (int)(b % G); tile = b / G; }                  // tile-major (12 months of a tile adjacent)
    else {                                                                      // XCD-aware: each XCD owns a contiguous tile range
        const int64_t per = (ntiles * G + 7) / 8; const int xcd = b % 8; const int64_t j = b / 8;
        const int64_t lin = xcd * per + j; if (lin >= ntiles * G) return;
        const int64_t tx = ntiles / 8;  // tiles per xcd (assume divisible)
        // within an XCD: month-major over its tile range
        const int64_t local = lin - xcd * per; tile = xcd * tx + local % tx; g = (int)(local / tx);
        if (g >= G) return;
    }
    const int beg = goff[g], n = goff[g + 1] - beg;
    constexpr int LW = W / VEC;           // lanes per row
    const int r0 = threadIdx.x / LW, cl = (threadIdx.x % LW) * VEC;
    const int64_t c0 = tile * W + cl;
    double acc = 0.0;
    for (int r = r0; r < n; r += 512 / LW) {
        const double* p = x + (int64_t)order[beg + r] * ld + c0;
        if (VEC == 1) acc += p[0];
        else { const double2 v = *reinterpret_cast<const double2*>(p); acc += v.x + v.y; }
    }
    if (acc == 1.2345678e300) sink[b] = acc;
}

__global__ void stream_copy(const double2* __restrict__ a, double2* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void stream_read(const double2* __restrict__ a, double* sink, int64_t n) {
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { double2 v = a[i]; acc += v.x + v.y; }
    if (acc == 1.2345678e300) sink[0] = acc;
}

template <int W, int VEC>
void run(const double* x, int64_t ld, const int* order, const int* goff, int G, int64_t C, int64_t T, double* sink, int mapping) {
    const int64_t nb = (C / W) * G;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((tile_read<W, VEC>), dim3((unsigned)nb), dim3(512), 0, 0, x, ld, order, goff, G, C, sink, mapping);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("tile_read W=%2d vec=%d mapping=%d : %8.3f ms  %8.1f GB/s\n", W, VEC, mapping, best, (double)T * C * 8 / best / 1e6);
}

int main(int argc, char** argv) {
    const int64_t T = 14600, C = argc > 1 ? atoll(argv[1]) : 100000; const int G = 12;
    // 40-year daily calendar from 1980-01-01
    std::vector<int> gid(T); { int y = 1980, m = 0, d = 0; const int dm[12] = {31,28,31,30,31,30,31,31,30,31,30,31};
        for (int64_t t = 0; t < T; ++t) { gid[t] = m; int len = dm[m] + ((m == 1 && (y % 4 == 0)) ? 1 : 0); if (++d == len) { d = 0; if (++m == 12) { m = 0; ++y; } } } }
    std::vector<int> off(G + 1, 0), order(T); for (auto g : gid) off[g + 1]++; for (int g = 0; g < G; ++g) off[g + 1] += off[g];
    { std::vector<int> cur(off.begin(), off.end() - 1); for (int64_t t = 0; t < T; ++t) order[cur[gid[t]]++] = (int)t; }
    double *x, *y2, *sink; int *dorder, *doff;
    CK(hipMalloc(&x, T * C * 8)); CK(hipMalloc(&y2, T * C * 8)); CK(hipMalloc(&sink, 8 * 4000000)); CK(hipMalloc(&dorder, T * 4)); CK(hipMalloc(&doff, (G + 1) * 4));
    CK(hipMemset(x, 0, T * C * 8)); CK(hipMemcpy(dorder, order.data(), T * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(doff, off.data(), (G + 1) * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
    const int64_t n2 = T * C / 2;
    for (int it = 0; it < 2; ++it) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(stream_copy, dim3(256 * 16), dim3(256), 0, 0, (const double2*)x, (double2*)y2, n2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    printf("stream_copy (r+w): %8.3f ms  %8.1f GB/s\n", ms, 2.0 * T * C * 8 / ms / 1e6);
    for (int it = 0; it < 2; ++it) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, 0, (const double2*)x, sink, n2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    printf("stream_read      : %8.3f ms  %8.1f GB/s\n", ms, 1.0 * T * C * 8 / ms / 1e6);
    for (int mapping = 0; mapping < 3; ++mapping) {
        run<4, 1>(x, C, dorder, doff, G, C, T, sink, mapping);
        run<4, 2>(x, C, dorder, doff, G, C, T, sink, mapping);
        run<8, 1>(x, C, dorder, doff, G, C, T, sink, mapping);
        run<8, 2>(x, C, dorder, doff, G, C, T, sink, mapping);
        run<16, 1>(x, C, dorder, doff, G, C, T, sink, mapping);
        run<16, 2>(x, C, dorder, doff, G, C, T, sink, mapping);
        run<32, 2>(x, C, dorder, doff, G, C, T, sink, mapping);
        run<64, 2>(x, C, dorder, doff, G, C, T, sink, mapping);
    }
    return 0;
}
