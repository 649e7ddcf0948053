// Issue cost of the vector instructions the register wave sort is made of, measured by wall clock over a saturated chip:
// W waves per SIMD, every wave runs 16 independent chains of one instruction kind; reports ns and (at 2.4 GHz) cycles
// per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 8192;
constexpr int kCh = 16;

template <int OP>
__global__ void __launch_bounds__(256) kern(unsigned* out, int iters) {
    unsigned a[kCh];
    double d[kCh / 2];
    const unsigned b = out[threadIdx.x], c = out[threadIdx.x + 1];
    const double bd = (double)b;
#pragma unroll
    for (int i = 0; i < kCh; ++i) a[i] = out[threadIdx.x + i] + i;
#pragma unroll
    for (int i = 0; i < kCh / 2; ++i) d[i] = (double)a[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < kCh; ++i) {
                if (OP == 0) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 8 && i < kCh / 2) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
                if (OP == 9 && i < kCh / 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
                if (OP == 10 && i < kCh / 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(bd));
                if (OP == 11) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 12) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 13) asm volatile("v_min_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (OP == 14 && i < kCh / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(bd));
                if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
                if (OP == 16) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (OP == 18) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 19) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 20) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 21) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 22) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
                if (OP == 23) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
                if (OP == 24) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
                if (OP == 25) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 26) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == 27) asm volatile("v_bfe_u32 %0, %0, 3, 11" : "+v"(a[i]));
                if (OP == 28) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 29 && i < kCh / 2) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
                if (OP == 33 && i < kCh / 2) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c) : "s20", "s21");
                if (OP == 34 && i < kCh / 2) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(d[i]) : "v"(bd));
                if (OP == 35) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 36) asm volatile("v_lshl_or_b32 %0, %0, 11, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 37) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 38) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 39) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 40) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 41) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 42) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == 43) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 44 && i < kCh / 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
            }
            if (OP == 30) {  // the sort's cross stage pattern: 16 ds_swizzle then 16 med3
                unsigned t[kCh];
#pragma unroll
                for (int i = 0; i < kCh; ++i) asm volatile("ds_swizzle_b32 %0, %1 offset:0x101F" : "=v"(t[i]) : "v"(a[i]));
                asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
                for (int i = 0; i < kCh; ++i) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(t[i]), "v"(c));
            }
            if (OP == 31) {  // DPP move + med3
                unsigned t[kCh];
#pragma unroll
                for (int i = 0; i < kCh; ++i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t[i]) : "v"(a[i]));
#pragma unroll
                for (int i = 0; i < kCh; ++i) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(t[i]), "v"(c));
            }
            if (OP == 32) {  // compare-exchange pairs
#pragma unroll
                for (int i = 0; i < kCh; i += 2) {
                    unsigned lo, hi;
                    asm volatile("v_min_u32 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(a[i + 1]));
                    asm volatile("v_max_u32 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(a[i + 1]));
                    a[i] = lo;
                    a[i + 1] = hi;
                }
            }
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kCh; ++i) s += a[i];
    double sd = 0;
#pragma unroll
    for (int i = 0; i < kCh / 2; ++i) sd += d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (unsigned)sd;
}

struct Case { const char* name; void (*k)(unsigned*, int); int per_iter; };

int main() {
    unsigned* out;
    CK(hipMalloc(&out, 4 * (256 * 2048 * 2 + 64)));
    CK(hipMemset(out, 1, 4 * (256 * 2048 * 2 + 64)));
    Case cs[] = {
        {"v_min_u32", kern<0>, 64}, {"v_med3_u32", kern<1>, 64}, {"v_mov_b32_dpp quad_perm", kern<2>, 64},
        {"v_add_u32", kern<3>, 64}, {"v_min_f32", kern<4>, 64}, {"v_fma_f32", kern<5>, 64}, {"v_xor_b32", kern<6>, 64},
        {"v_lshl_add_u32", kern<7>, 64}, {"v_min_f64", kern<8>, 32}, {"v_add_f64", kern<9>, 32}, {"v_fma_f64", kern<10>, 32},
        {"v_mov_b32_dpp row_ror:8", kern<11>, 64}, {"v_max3_u32", kern<12>, 64}, {"v_min_u32_dpp", kern<13>, 64},
        {"v_pk_fma_f32", kern<14>, 32}, {"v_mov_b32", kern<15>, 64}, {"v_cmp_lt_u32", kern<16>, 64}, {"v_cndmask_b32", kern<17>, 64},
        {"v_pk_min_u16", kern<18>, 64}, {"v_min3_u32", kern<19>, 64}, {"v_and_or_b32", kern<20>, 64}, {"v_perm_b32", kern<21>, 64},
        {"v_cndmask_b32_e64 sgpr mask", kern<22>, 64}, {"v_cmp_lt_u32_e64 -> sgpr", kern<23>, 64}, {"v_cmp_e64 + v_cndmask (pair)", kern<24>, 64},
        {"v_and_b32", kern<25>, 64}, {"v_lshlrev_b32", kern<26>, 64}, {"v_bfe_u32", kern<27>, 64}, {"v_sub_u32", kern<28>, 64},
        {"v_cvt_u32_f64", kern<29>, 32}, {"v_mad_u64_u32", kern<33>, 32}, {"v_lshl_add_u64", kern<34>, 32}, {"v_or_b32", kern<35>, 64},
        {"v_lshl_or_b32", kern<36>, 64}, {"v_add3_u32", kern<37>, 64}, {"v_mul_f32", kern<38>, 64}, {"v_add_f32", kern<39>, 64},
        {"v_max_f32", kern<40>, 64}, {"v_mul_lo_u32", kern<41>, 64}, {"v_cmp vcc + v_addc (pair)", kern<42>, 64},
        {"v_mov_b32_dpp row_shr:1", kern<43>, 64}, {"v_mul_f64", kern<44>, 32},
        {"stage: 16 ds_swizzle + 16 med3 (32 instr)", kern<30>, 128}, {"stage: 16 mov_dpp + 16 med3 (32 instr)", kern<31>, 128},
        {"8 x (min_u32 + max_u32) on pairs", kern<32>, 64},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto& c : cs) {
        printf("%-44s", c.name);
        for (int w : {1, 2, 4, 8}) {
            const int blocks = 256 * w;  // 256-thread blocks = one wave per SIMD each
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 64);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, kIters);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = (double)w * kIters * c.per_iter;
            const double ns = ms * 1e6 / instr_per_simd;
            printf("  W=%d: %.3f ns (%.2f cyc@2.4)", w, ns, ns * 2.4);
        }
        printf("\n");
    }
    return 0;
}
