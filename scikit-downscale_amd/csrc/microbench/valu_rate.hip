// Issue rate of the vector instructions the sort / search kernels are made of (cycles per wave instruction on one SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 4096;
constexpr int kChains = 8;  // independent dependency chains per thread

#define KERNEL(name, DECL, BODY)                                                         \
    __global__ void name(double* out, long long* cyc) {                                   \
        DECL;                                                                             \
        const long long t0 = clock64();                                                   \
        for (int it = 0; it < kIters; ++it) {                                             \
            BODY                                                                          \
        }                                                                                 \
        const long long t1 = clock64();                                                   \
        double s = 0;                                                                     \
        for (int c = 0; c < kChains; ++c) s += a[c];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                  \
    }

#define DECL_F64 double a[kChains], b = out[threadIdx.x]; for (int c = 0; c < kChains; ++c) a[c] = out[threadIdx.x + c]
#define ASM2(op) _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile(op " %0, %0, %1" : "+v"(a[c]) : "v"(b));

KERNEL(k_min_f64, DECL_F64, ASM2("v_min_f64"))
KERNEL(k_max_f64, DECL_F64, ASM2("v_max_f64"))
KERNEL(k_add_f64, DECL_F64, ASM2("v_add_f64"))
KERNEL(k_mul_f64, DECL_F64, ASM2("v_mul_f64"))
KERNEL(k_fma_f64, DECL_F64, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));)
KERNEL(k_cmp_f64, DECL_F64, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(a[c]), "v"(b) : "vcc");)

#define DECL_I32 int ai[kChains], bi = (int)out[threadIdx.x]; double a[kChains]; for (int c = 0; c < kChains; ++c) { ai[c] = (int)out[threadIdx.x + c]; a[c] = 0; }
#define FIN_I32 for (int c = 0; c < kChains; ++c) a[c] = ai[c];
KERNEL(k_add_u32, DECL_I32, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ai[c]) : "v"(bi)); if (it == kIters - 1) { FIN_I32 })
KERNEL(k_cndmask, DECL_I32, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ai[c]) : "v"(bi)); if (it == kIters - 1) { FIN_I32 })
KERNEL(k_min_u32, DECL_I32, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_min_u32 %0, %0, %1" : "+v"(ai[c]) : "v"(bi)); if (it == kIters - 1) { FIN_I32 })
KERNEL(k_lshl_add, DECL_I32, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(ai[c]) : "v"(bi)); if (it == kIters - 1) { FIN_I32 })
KERNEL(k_min_f32, DECL_I32, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_min_f32 %0, %0, %1" : "+v"(ai[c]) : "v"(bi)); if (it == kIters - 1) { FIN_I32 })
KERNEL(k_cmp_u64, DECL_F64, _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[c]), "v"(b) : "vcc");)

int main() {
    const int threads = 256, blocks = 1024;  // 4 waves per block: one per SIMD when one block sits on a CU
    double* out;
    long long* cyc;
    CHECK(hipMalloc(&out, sizeof(double) * (threads * blocks + 64)));
    CHECK(hipMemset(out, 0, sizeof(double) * (threads * blocks + 64)));
    CHECK(hipMalloc(&cyc, sizeof(long long) * blocks));
    struct { const char* name; void (*k)(double*, long long*); int per_iter; } ks[] = {
        {"v_min_f64", k_min_f64, kChains}, {"v_max_f64", k_max_f64, kChains}, {"v_add_f64", k_add_f64, kChains},
        {"v_mul_f64", k_mul_f64, kChains}, {"v_fma_f64", k_fma_f64, kChains}, {"v_cmp_le_f64", k_cmp_f64, kChains},
        {"v_add_u32", k_add_u32, kChains}, {"v_cndmask_b32", k_cndmask, kChains}, {"v_min_u32", k_min_u32, kChains},
        {"v_lshl_add_u32", k_lshl_add, kChains}, {"v_min_f32", k_min_f32, kChains}, {"v_cmp_lt_u64", k_cmp_u64, kChains}};
    for (auto& e : ks) {
        for (int waves = 1; waves <= 4; waves *= 2) {  // waves per SIMD: blocks of 256 threads, `waves` blocks per CU resident
            hipLaunchKernelGGL(e.k, dim3(256 * waves), dim3(threads), 0, 0, out, cyc);
            CHECK(hipDeviceSynchronize());
            std::vector<long long> h(256 * waves);
            CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= (double)h.size();
            // clock64 = s_memtime shader clock (100 MHz-based?) -- report relative to v_add_u32 as well
            printf("%-26s waves/SIMD=%d  ticks/iter/wave-instr = %.3f\n", e.name, waves, mean / kIters / e.per_iter * 1.0);
        }
    }
    return 0;
}
