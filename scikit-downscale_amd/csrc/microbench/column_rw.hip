// Can per-cell workgroups read / write the columns of a time-major [T, C] field directly (8-byte accesses, stride C * 8)
// when the workgroups of 16 adjacent cells run at the same time on one XCD, so that every 128-byte line is fetched once
// and shared through that XCD's L2?  (The alternative is what sd_analog.hip does: cell-major staging copies.)
// Build: hipcc -O3 --offload-arch=gfx950 column_rw.hip -o column_rw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// workgroup b runs on XCD b % 8: XCD x owns cell groups of `grp` adjacent cells, walked group-fastest
__device__ __forceinline__ long cell_of(unsigned b, long C, int grp, int per_xcd_wg) {
    const int xcd = b & 7;
    const long j = b >> 3;                 // workgroup index within the XCD
    const long groups = C / grp / 8;       // groups per XCD (C a multiple of 8 * grp)
    (void)per_xcd_wg;
    return (xcd * groups + j / grp) * grp + j % grp;
}
template <int MODE>  // 0 read column -> sum, 1 write column, 2 read + write (column copy), 3 naive mapping read
__global__ void __launch_bounds__(1024) column_kernel(const double* __restrict__ src, double* __restrict__ dst, double* sums, long T, long C, int grp) {
    extern __shared__ double lds[];
    const long c = MODE == 3 ? (long)blockIdx.x : cell_of(blockIdx.x, C, grp, 0);
    if (c >= C) return;
    double s = 0.0;
    if (MODE == 0 || MODE == 2 || MODE == 3) {
        for (long t = threadIdx.x; t < T; t += 1024) {
            const double v = src[t * C + c];
            lds[t] = v;
            s += v;
        }
    }
    __syncthreads();
    if (MODE == 1 || MODE == 2) {
        for (long t = threadIdx.x; t < T; t += 1024) dst[t * C + c] = (MODE == 2 ? lds[t] : (double)t) + 1.0;
    }
    if (threadIdx.x == 0) sums[c] = s;
}
int main(int argc, char** argv) {
    const long T = 14600, C = argc > 1 ? atol(argv[1]) : 32768;
    double *a, *b, *sums;
    CHECK(hipMalloc(&a, sizeof(double) * T * C));
    CHECK(hipMalloc(&b, sizeof(double) * T * C));
    CHECK(hipMalloc(&sums, sizeof(double) * C));
    CHECK(hipMemset(a, 0, sizeof(double) * T * C));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds = sizeof(double) * T;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&column_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&column_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&column_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&column_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char* names[] = {"read column (XCD-grouped)", "write column (XCD-grouped)", "copy column (XCD-grouped)", "read column (cell = block id)"};
    for (int grp : {8, 16, 32}) {
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 3 && grp != 8) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(column_kernel<0>, dim3((unsigned)C), dim3(1024), lds, 0, a, b, sums, T, C, grp); break;
                    case 1: hipLaunchKernelGGL(column_kernel<1>, dim3((unsigned)C), dim3(1024), lds, 0, a, b, sums, T, C, grp); break;
                    case 2: hipLaunchKernelGGL(column_kernel<2>, dim3((unsigned)C), dim3(1024), lds, 0, a, b, sums, T, C, grp); break;
                    default: hipLaunchKernelGGL(column_kernel<3>, dim3((unsigned)C), dim3(1024), lds, 0, a, b, sums, T, C, grp); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double bytes = (double)T * C * 8 * (mode == 2 ? 2 : 1);
            printf("C=%ld grp=%2d %-32s %8.3f ms  %7.1f GB/s algorithmic\n", C, grp, names[mode], best, bytes / best / 1e6);
        }
    }
    return 0;
}
