// Microbenchmark (round 6, the round-5 review's open question): one cross-lane compare-exchange stage of the register sort (sd_wsort.h)
// over KEYS keys per lane, with the partner lane 32 (or 16) lanes away,
//   (a) as the kernels do it: fetch the partner's key (ds_swizzle / v_mov_b32_dpp, here ds_bpermute for lane ^ 32 and row_ror:8 ...
//       -- the generic form: ds_bpermute_b32) and v_med3_u32(own, partner, sel): 2 instructions per key;
//   (b) with v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950): two key registers swap halves, then the pairs sit in the same
//       lane: v_min_u32 + v_max_u32, and a second swap puts minima / maxima back: 4 instructions per 2 keys.
// Saturated chip (4 waves per SIMD), the stage repeated in a long unrolled loop; reports cycles per key and stage.
// Build: hipcc -O3 --offload-arch=gfx950 permlane_swap.hip -o permlane_swap_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

constexpr int KEYS = 20, REPS = 64;

template <int MODE>
__global__ void __launch_bounds__(256) stage_kernel(unsigned* out, int iters) {
    unsigned k[KEYS];
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int i = 0; i < KEYS; ++i) k[i] = (threadIdx.x + 1u) * 2654435761u + (unsigned)i * 0x9e3779b9u + blockIdx.x;
    const unsigned sel = (lane & 32u) ? ~0u : 0u;  // upper half keeps the maximum
    const int partner_addr = (int)((lane ^ 32u) << 2);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REPS; ++r) {
            if (MODE == 0) {  // ds_bpermute + med3
#pragma unroll
                for (int i = 0; i < KEYS; ++i) {
                    const unsigned p = (unsigned)__builtin_amdgcn_ds_bpermute(partner_addr, (int)k[i]);
                    unsigned m;
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k[i]), "v"(p), "v"(sel));
                    k[i] = m;
                }
            } else if (MODE == 1) {  // row_mirror DPP (a within-row partner: the cheapest fetch the sort has) + med3
#pragma unroll
                for (int i = 0; i < KEYS; ++i) {
                    const unsigned p = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k[i], 0x140 /* row_mirror */, 0xf, 0xf, false);
                    unsigned m;
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k[i]), "v"(p), "v"(sel));
                    k[i] = m;
                }
            } else if (MODE == 2) {  // permlane32_swap, min, max, permlane32_swap: two keys at a time
#pragma unroll
                for (int i = 0; i < KEYS; i += 2) {
                    auto s = __builtin_amdgcn_permlane32_swap(k[i], k[i + 1], false, false);
                    unsigned mn, mx;
                    asm volatile("v_min_u32 %0, %1, %2" : "=v"(mn) : "v"(s[0]), "v"(s[1]));
                    asm volatile("v_max_u32 %0, %1, %2" : "=v"(mx) : "v"(s[0]), "v"(s[1]));
                    auto t = __builtin_amdgcn_permlane32_swap(mn, mx, false, false);
                    k[i] = t[0];
                    k[i + 1] = t[1];
                }
            } else if (MODE == 4) {  // ds_swizzle xor 4 + med3 (what the sort does for lane ^ 4)
#pragma unroll
                for (int i = 0; i < KEYS; ++i) {
                    const unsigned p = (unsigned)__builtin_amdgcn_ds_swizzle((int)k[i], 0x101F);
                    unsigned m;
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k[i]), "v"(p), "v"(sel));
                    k[i] = m;
                }
            } else if (MODE == 5) {  // lane ^ 4 with two masked DPP moves (row_shl:4 into banks 0 / 2, row_shr:4 into banks 1 / 3) + med3
#pragma unroll
                for (int i = 0; i < KEYS; ++i) {
                    int p = __builtin_amdgcn_update_dpp(0, (int)k[i], 0x104 /* row_shl:4 */, 0xf, 0x5, false);
                    p = __builtin_amdgcn_update_dpp(p, (int)k[i], 0x114 /* row_shr:4 */, 0xf, 0xa, false);
                    unsigned m;
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k[i]), "v"((unsigned)p), "v"(sel));
                    k[i] = m;
                }
            } else if (MODE == 6) {  // ds_swizzle xor 16 + med3 (what the sort does for lane ^ 16)
#pragma unroll
                for (int i = 0; i < KEYS; ++i) {
                    const unsigned p = (unsigned)__builtin_amdgcn_ds_swizzle((int)k[i], 0x401F);
                    unsigned m;
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k[i]), "v"(p), "v"(sel));
                    k[i] = m;
                }
            } else if (MODE == 3) {  // permlane16_swap (partner 16 lanes away)
#pragma unroll
                for (int i = 0; i < KEYS; i += 2) {
                    auto s = __builtin_amdgcn_permlane16_swap(k[i], k[i + 1], false, false);
                    unsigned mn, mx;
                    asm volatile("v_min_u32 %0, %1, %2" : "=v"(mn) : "v"(s[0]), "v"(s[1]));
                    asm volatile("v_max_u32 %0, %1, %2" : "=v"(mx) : "v"(s[0]), "v"(s[1]));
                    auto t = __builtin_amdgcn_permlane16_swap(mn, mx, false, false);
                    k[i] = t[0];
                    k[i + 1] = t[1];
                }
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < KEYS; ++i) acc ^= k[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, unsigned* out, int blocks, int iters, double ghz) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stage_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stage_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // waves per SIMD = 4 (blocks = 4 per CU of 4 waves); cycles per wave-level key-stage on one SIMD:
    const double stages_per_simd = 4.0 /* waves */ * (double)iters * REPS * KEYS;
    printf("%-44s %8.2f ms  %6.2f cycles per key and stage per SIMD (at %.1f GHz)\n", name, ms, ms * 1e-3 * ghz * 1e9 / stages_per_simd, ghz);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 4;  // 4 blocks of 4 waves per CU = 4 waves per SIMD
    const double ghz = prop.clockRate / 1e6;
    unsigned* out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(unsigned)));
    printf("%d CUs, %d blocks of 256 threads, %d keys per lane, clock %.2f GHz (nominal)\n", cus, blocks, KEYS, ghz);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("ds_bpermute_b32 + v_med3_u32", out, blocks, iters, ghz);
        run<1>("v_mov_b32_dpp row_mirror + v_med3_u32", out, blocks, iters, ghz);
        run<2>("v_permlane32_swap x2 + v_min/max_u32 (2 keys)", out, blocks, iters, ghz);
        run<3>("v_permlane16_swap x2 + v_min/max_u32 (2 keys)", out, blocks, iters, ghz);
        run<4>("ds_swizzle_b32 xor 4 + v_med3_u32", out, blocks, iters, ghz);
        run<5>("2 x v_mov_b32_dpp (row_shl/shr:4, masked) + med3", out, blocks, iters, ghz);
        run<6>("ds_swizzle_b32 xor 16 + v_med3_u32", out, blocks, iters, ghz);
    }
    return 0;
}
