// Microbenchmark: the access pattern of the 'weight_analogs' window reads (analog_f1_mean_kernel): every lane reads k consecutive
// doubles of a 14 600-sample series, the 64 windows of a wave start at unrelated positions.  What bounds it: bytes, load
// instructions, or line requests (64 distinct 128-byte lines per wave instruction whatever its width)?
//   * 8-byte loads (k per query) against 16-byte loads (k / 2 per query), windows at random positions;
//   * the same with the windows of neighbouring lanes one sample apart (queries in value order): a wave instruction then touches
//     5 (9) lines instead of 64.
// One 1024-thread workgroup per CU walks over the cells, as the kernel does.
//   hipcc -O3 --offload-arch=gfx950 csrc/microbench/window_reads.hip -o csrc/microbench/window_reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));

template <int W, bool NEIGHBOURS>
__global__ void __launch_bounds__(1024) window_kernel(const double* __restrict__ yx, int n, int C, int Tq, int k, double* __restrict__ out,
                                                      long long* clocks) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const long long t0 = wall_clock64();
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const double* y = yx + (size_t)c * n;
        for (int tq = tid; tq < Tq; tq += nthr) {
            unsigned L;
            if (NEIGHBOURS) L = (unsigned)(((long long)tq * (n - k)) / Tq);
            else L = ((unsigned)(tq + 977 * c) * 2654435761u) % (unsigned)(n - k);
            const double* w = y + L;
            // 48 bytes per batch either way, written out (the compiler would merge adjacent 8-byte loads into 16-byte ones)
            double s = 0.0;
            for (int i = 0; i + 5 < k; i += 6) {
                if (W == 1) {
                    double a0, a1, a2, a3, a4, a5;
                    asm volatile(
                        "global_load_dwordx2 %0, %6, off\n\tglobal_load_dwordx2 %1, %6, off offset:8\n\tglobal_load_dwordx2 %2, %6, off offset:16\n\t"
                        "global_load_dwordx2 %3, %6, off offset:24\n\tglobal_load_dwordx2 %4, %6, off offset:32\n\tglobal_load_dwordx2 %5, %6, off offset:40\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5)
                        : "v"(w + i)
                        : "memory");
                    s += a0; s += a1; s += a2; s += a3; s += a4; s += a5;
                } else {
                    f64x2_a8 v0, v1, v2;
                    asm volatile(
                        "global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx4 %2, %3, off offset:32\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(v0), "=&v"(v1), "=&v"(v2)
                        : "v"(w + i)
                        : "memory");
                    s += v0.x; s += v0.y; s += v1.x; s += v1.y; s += v2.x; s += v2.y;
                }
            }
            out[(size_t)c * Tq + tq] = s;
        }
    }
    if (tid == 0) clocks[blockIdx.x] = wall_clock64() - t0;
}

template <int W, bool NB>
static void run(const char* name, const double* yx, int n, int C, int Tq, int k, double* out, long long* clk, int ncu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((window_kernel<W, NB>), dim3(ncu), dim3(1024), 0, 0, yx, n, C, Tq, k, out, clk);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double instr_per_cu = (double)C / ncu * (Tq / 64.0) * (W == 1 ? k : k / 2);  // wave load instructions per CU
    const double ns_per_instr = best * 1e6 / instr_per_cu;
    printf("%-44s %8.3f ms   %6.2f ms per 100 000 cells and window element   %6.1f ns per wave load   %5.2f lane-lines per ns and CU\n", name, best,
           best * (100000.0 / C) / k, ns_per_instr, 64.0 / ns_per_instr);
}

int main() {
    const int n = 14600, C = 4096, Tq = 14336, k = 30;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    double *yx, *out; long long* clk;
    CK(hipMalloc(&yx, (size_t)C * n * 8)); CK(hipMalloc(&out, (size_t)C * Tq * 8)); CK(hipMalloc(&clk, ncu * 8));
    CK(hipMemset(yx, 0, (size_t)C * n * 8));
    printf("CUs %d, %d cells x %d samples, %d queries per cell, windows of %d doubles (shader clock ~2.1 GHz under load)\n", ncu, C, n, Tq, k);
    run<1, false>("random windows, 8-byte loads", yx, n, C, Tq, k, out, clk, ncu);
    run<2, false>("random windows, 16-byte loads", yx, n, C, Tq, k, out, clk, ncu);
    run<1, true>("neighbouring windows, 8-byte loads", yx, n, C, Tq, k, out, clk, ncu);
    run<2, true>("neighbouring windows, 16-byte loads", yx, n, C, Tq, k, out, clk, ncu);
    return 0;
}
