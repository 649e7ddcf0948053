// Microbenchmark: streaming bandwidth of a copy kernel restricted to a subset of the CUs (hipExtStreamCreateWithCUMask), and which
// XCDs the masked stream's workgroups land on: can bandwidth-bound staging kernels run beside an LDS-bound kernel on disjoint CUs?
//   hipcc -O3 --offload-arch=gfx950 csrc/microbench/cu_mask_bw.hip -o csrc/microbench/cu_mask_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) copy_kernel(const double2* __restrict__ a, double2* __restrict__ b, int64_t n, unsigned* xcc_count) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 15u;  // HW_REG_XCC_ID, bits 3:0
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (8 << 6) | ((8 - 1) << 11));          // HW_REG_HW_ID bits 15:8: cu_id[3:0], sh_id, se_id[2:0]
        atomicAdd(&xcc_count[xcc], 1u);
        atomicOr(&xcc_count[16 + xcc * 8 + ((hw & 255u) >> 5)], 1u << (hw & 31u));  // per XCC: 256-bit set of (se, sh, cu) ids seen
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
    const int64_t n = (int64_t)1 << 28;  // 4 GB read + 4 GB written
    double2 *a, *b;
    unsigned* cnt;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&cnt, 4 * (16 + 64)));
    CK(hipMemset(a, 0, n * 16));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("CUs: %d\n", ncu);
    struct Case { const char* name; int stride, count, first; };
    const Case cases[] = {{"all", 1, ncu, 0}, {"first 64 bits", 1, 64, 0}, {"every 4th bit (64)", 4, 64, 0}, {"every 2nd bit (128)", 2, 128, 0},
                          {"first 128 bits", 1, 128, 0}, {"bits 0-7 of every 32 (64)", -32, 64, 0}, {"bits 0-11 of every 32 (96)", -3212, 96, 0}, {"every 8th bit (32)", 8, 32, 0}};
    for (const Case& c : cases) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        int set = 0;
        if (c.stride > 0) {
            for (int i = c.first; i < ncu && set < c.count; i += c.stride) { mask[i / 32] |= 1u << (i % 32); ++set; }
        } else {
            const int per = c.stride == -32 ? 8 : 12;
            for (int i = 0; i < ncu; ++i) if (i % 32 < per) { mask[i / 32] |= 1u << (i % 32); ++set; }
        }
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int it = 0; it < 3; ++it) {
            CK(hipMemsetAsync(cnt, 0, 4 * (16 + 64), s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(copy_kernel, dim3(ncu * 16), dim3(256), 0, s, (const double2*)a, b, n, cnt);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        unsigned h[16 + 64]; CK(hipMemcpy(h, cnt, 4 * (16 + 64), hipMemcpyDeviceToHost));
        printf("%-28s %3d CUs: %8.3f ms  %7.1f GB/s   workgroups per XCC:", c.name, set, best, 2.0 * n * 16 / best / 1e6);
        for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
        printf("   distinct CUs per XCC:");
        int tot = 0;
        for (int x = 0; x < 8; ++x) { int d = 0; for (int w = 0; w < 8; ++w) d += __builtin_popcount(h[16 + x * 8 + w]); printf(" %d", d); tot += d; }
        printf(" (total %d)\n", tot);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
