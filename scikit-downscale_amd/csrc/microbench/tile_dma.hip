// Microbenchmark (round 6): how the tiles of the fused BCSD kernel can land in LDS.
//
// A workgroup = 8 adjacent cells x one month of a [T, C] float64 field (512 threads, 81.8 KB of LDS: two per CU), XCD-aware
// block -> (tile, month) map as in sd_wave.h.  Per item it reads three tiles (x_hist, x_fut, y_obs of the real kernel), sums
// every cell's column from LDS (one wave per cell, lane l owns samples 20 l .. 20 l + 19, like the kernel's K = 20 layout) and
// writes one tile out.  Two ways of getting a tile into LDS:
//
//   mode 0  REG  today's path: global_load_dwordx4 of 64-byte row fragments into registers (10 rows per thread in flight),
//                ds_write_b64 into per-cell rows (slot = 4 + j + j / 160), barrier, conflict-free column reads
//   mode 1  DMA  global_load_lds_dwordx4: a wave instruction lands 16 rows x 64 B = 1 KB lane-linear ([t][8 cells], time-major);
//                the base of every 16-row chunk is skewed by 16 B (chunk stride 1 040 B) so that the K-blocked column reads
//                (lane stride 20 rows) spread over the banks; no tile registers, no ds_write
//   mode 2  DMA2 as 1, but the second and third tile are requested while the column sums of the previous tile are taken
//                from a second landing buffer -- needs 2 x 81 KB: one workgroup per CU (what "loads in flight own no VGPRs" buys
//                when LDS is there to land in)
//   mode 3  LDS  column reads only (no global traffic): cost of the skewed time-major column reads against the cell-major rows
//                (mode 3: time-major skewed; mode 4: cell-major rows), 64 repetitions per item
//
// Prints ms and GB/s of the algorithmic bytes (3 reads + 1 write per sample) and checks the column sums.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e = (x);                                                                           \
        if (e != hipSuccess) {                                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);              \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)

constexpr int K = 20, kW = 8, kThreads = 512, kRowsPerPass = 128, NR = K / 2;
constexpr int kChunkRows = 16;
__constant__ int kChunkStride = 1024 + 16;  // bytes between chunk bases (set per run)
__constant__ int kLpc = 0;               // 1: lane-per-chunk layout (chunk l = rows 20 l .. 20 l + 15, chunks 64.. = the tails)
constexpr int kHeadBytes = 704;

struct P {
    const double* x[3];
    double* out;
    double* sums;  // [3][G][C]
    const int* ord;
    const int* off;
    int64_t ld, C, ntiles;
    int G;
};

// row held by slot S of chunk q
__device__ __forceinline__ int row_of_slot(int q, int S, int n) {
    if (!kLpc) return kChunkRows * q + S;
    const int nl = (n + 19) / 20;  // lanes with data: main chunks 0 .. nl - 1, tails behind them
    if (q < nl) return 20 * q + S;
    const int t = q - nl;
    const int l = (t & 7) + 8 * (S & 3) + 32 * (t >> 3);
    return 20 * l + 16 + (S >> 2);
}

__device__ __forceinline__ void xcd_tile_of_block(unsigned b, int64_t ntiles, int64_t* tile_id, int* gslot) {
    const int64_t tx = (ntiles + 7) / 8;
    const int xcd = (int)(b & 7u);
    const int64_t jb = (int64_t)(b >> 3);
    *tile_id = xcd * tx + jb % tx;
    *gslot = (int)(jb / tx);
}
__device__ __forceinline__ const double* row_of(const double* cp, int ti, int64_t ld) {
    const uint64_t off = (uint64_t)(uint32_t)ti * (uint64_t)(uint32_t)((uint32_t)ld * 8u);
    return reinterpret_cast<const double*>(reinterpret_cast<const char*>(cp) + off);
}
__device__ __forceinline__ int slot_of(int r) { return 4 + r + r / 160; }
typedef __attribute__((address_space(3))) const double lds_cdouble_t;
__device__ __forceinline__ double lds_f64(unsigned addr) { return *reinterpret_cast<lds_cdouble_t*>((uintptr_t)addr); }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- time-major skewed tile: addresses ---------------------------------------------------------------------------------
// row j of the tile sits at (j >> 4) * 1040 + (j & 15) * 64; lane l owns rows 20 l + i: 20 l = 16 (l + l / 4) + 4 (l & 3), so
// addr(i) = base2 + 64 i + 16 * ((4 p + i) >> 4), p = l & 3: per zone of four samples one of two per-lane bases
struct ColBases {
    unsigned z[5];
};
__device__ __forceinline__ ColBases col_bases(unsigned tile_b, int lane, int wave, int nl) {
    if (kLpc) {  // zones 0..3: the lane's own chunk; zone 4: its four tail rows
        ColBases b;
        const unsigned main = tile_b + (unsigned)lane * (unsigned)kChunkStride + 8u * (unsigned)wave;
        const unsigned tail = tile_b + (unsigned)(nl + (lane & 7) + 8 * (lane >> 5)) * (unsigned)kChunkStride + 64u * (unsigned)((lane >> 3) & 3) + 8u * (unsigned)wave;
#pragma unroll
        for (int z = 0; z < 4; ++z) b.z[z] = main;
        b.z[4] = tail - 64u * 16u;  // (col_sum_tm adds 64 i, i = 16..19: rows 16 + e sit at tail + 256 e)
        return b;
    }
    const int p = lane & 3;
    const unsigned base2 = tile_b + (unsigned)(lane + (lane >> 2)) * (unsigned)kChunkStride + 256u * (unsigned)p + 8u * (unsigned)wave;
    ColBases b;
#pragma unroll
    for (int z = 0; z < 5; ++z) b.z[z] = base2 + (p + z >= 4 ? 16u : 0u);
    return b;
}
__device__ __forceinline__ double col_sum_tm(const ColBases& b, int lane, int n) {
    double s = 0.0;
    if (K * lane < n) {
#pragma unroll
        for (int i = 0; i < K; ++i) s += lds_f64(b.z[i / 4] + (i >= 16 && kLpc ? 64u * 16u + 256u * (unsigned)(i - 16) : 64u * (unsigned)i));
    }
    return wave_sum(s);
}

// Row indices of a wave's chunks, loaded once per order table: element e = 16 k + i (k-th chunk of the wave, row i of it) sits in
// register e >> 6, lane e & 63 -- three registers for the 10 chunks a wave lands of a 1 240-row tile.  A request then takes its
// row index with one ds_bpermute (LDS crossbar, lgkmcnt): no ordinary global load -- whose wait would drain the DMA queue -- is
// needed while requests are in flight.
constexpr int kMaxChunksPerWave = 10, kIdxRegs = (16 * kMaxChunksPerWave + 63) / 64;
struct RowIdx {
    int v[kIdxRegs];
};
__device__ __forceinline__ RowIdx rows_of_wave(const int* __restrict__ ord, int n, int wave, int lane) {
    RowIdx t;
#pragma unroll
    for (int j = 0; j < kIdxRegs; ++j) {
        const int e = 64 * j + lane;
        const int r = row_of_slot(wave + kW * (e >> 4), e & 15, n);
        t.v[j] = ord[r < n ? r : n - 1];
    }
    // the values are "used" here, so the compiler waits for them here: a wait it placed later, between the requests, would count
    // only its own loads and drain the DMA queue with them
#pragma unroll
    for (int j = 0; j < kIdxRegs; ++j) asm volatile("" : "+v"(t.v[j]));
    return t;
}
// DMA of one tile: wave w lands chunks w, w + 8, ...; lane L of a chunk fetches 16 B of row 16 q + L / 4
__device__ __forceinline__ void tile_dma(const double* __restrict__ src, int64_t ld, const RowIdx& rows, int n, int64_t c0,
                                         char* tile, int wave, int lane) {
    const int nchunks = kLpc ? (n + 19) / 20 + 16 : (n + kChunkRows - 1) / kChunkRows;
    const char* colp = reinterpret_cast<const char*>(src + c0) + 16 * (lane & 3);
#pragma unroll
    for (int k = 0; k < kMaxChunksPerWave; ++k) {
        const int q = wave + kW * k;
        if (q < nchunks) {  // (wave-uniform)
            const int ti = __builtin_amdgcn_ds_bpermute(4 * (16 * (k & 3) + (lane >> 2)), rows.v[k >> 2]);
            const char* g = colp + (uint64_t)(uint32_t)ti * (uint64_t)(uint32_t)((uint32_t)ld * 8u);
            // inline asm, not __builtin_amdgcn_global_load_lds: the compiler treats a pending DMA as an LDS write and puts
            // s_waitcnt vmcnt(0) in front of the next DS instruction of any kind (the ds_bpermute of the next request here, the
            // ds_swizzle partner fetches of the sort in the real kernel); the waits are counted by hand instead
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(g), "s"(__builtin_amdgcn_readfirstlane((int)(lds_addr(tile) + (unsigned)(q * kChunkStride))))
                         : "memory");
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(kThreads, MODE == 2 ? 2 : 4) tile_kernel(const P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t tile_id;
    int g;
    xcd_tile_of_block(blockIdx.x, p.ntiles, &tile_id, &g);
    if (tile_id >= p.ntiles || g >= p.G) return;
    const int64_t c0 = tile_id * kW;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int beg = p.off[g], n = p.off[g + 1] - beg;
    const int* ord = p.ord + beg;
    char* tile = smem + kHeadBytes;

    if constexpr (MODE == 0) {
        constexpr int RS = 1258;
        double* rows = reinterpret_cast<double*>(tile);
        const int cp = tid & 3, rr = tid >> 2;
        for (int f = 0; f < 3; ++f) {
            double v0[NR], v1[NR];
            int ti[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int r = rr + k * kRowsPerPass;
                ti[k] = ord[r < n ? r : 0];
            }
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const double2 v = *reinterpret_cast<const double2*>(row_of(p.x[f] + c0 + 2 * cp, ti[k], p.ld));
                v0[k] = v.x;
                v1[k] = v.y;
            }
            double* d0 = rows + (2 * cp) * RS;
            double* d1 = d0 + RS;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int r = rr + k * kRowsPerPass;
                if (r < n) {
                    d0[slot_of(r)] = v0[k];
                    d1[slot_of(r)] = v1[k];
                }
            }
            __syncthreads();
            double s = 0.0;
            if (K * lane < n) {
                const double* ob = rows + wave * RS + 4 + K * lane + lane / 8;
#pragma unroll
                for (int i = 0; i < K; ++i) s += ob[i];
            }
            s = wave_sum(s);
            if (lane == 0) p.sums[((int64_t)f * p.G + g) * p.C + c0 + wave] = s;
            __syncthreads();
        }
        // the last tile goes out again
        const double* s0 = rows + (2 * cp) * RS;
        const double* s1 = s0 + RS;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rr + k * kRowsPerPass;
            if (r < n) *reinterpret_cast<double2*>(const_cast<double*>(row_of(p.out + c0 + 2 * cp, ord[r], p.ld))) = make_double2(s0[slot_of(r)], s1[slot_of(r)]);
        }
    } else if constexpr (MODE == 1 || MODE == 2) {
        const int kTileBytes = 78 * kChunkStride;
        char* buf[2] = {tile, MODE == 2 ? tile + kTileBytes : tile};
        const int nchunks = kLpc ? (n + 19) / 20 + 16 : (n + kChunkRows - 1) / kChunkRows;
        const RowIdx rows = rows_of_wave(ord, n, wave, lane);
        if (MODE == 2) tile_dma(p.x[0], p.ld, rows, n, c0, buf[0], wave, lane);
        for (int f = 0; f < 3; ++f) {
            char* cur = buf[f & 1];
            if (MODE == 1) {
                tile_dma(p.x[f], p.ld, rows, n, c0, cur, wave, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (f + 1 < 3) {
                    tile_dma(p.x[f + 1], p.ld, rows, n, c0, buf[(f + 1) & 1], wave, lane);
                    // this wave's pieces of tile f: everything but the (up to 10) requests of tile f + 1 issued just now
                    const int mine = (nchunks - wave + kW - 1) / kW;
                    if (mine >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
            const ColBases cb = col_bases(lds_addr(cur), lane, wave, (n + 19) / 20);
            const double s = col_sum_tm(cb, lane, n);
            if (lane == 0) p.sums[((int64_t)f * p.G + g) * p.C + c0 + wave] = s;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // the last tile goes out again: lane-linear 16-byte reads of the chunks, 64-byte row fragments to memory
        char* cur = buf[0];  // (f = 2 landed in buf[0] in both modes)
        char* colp = reinterpret_cast<char*>(p.out + c0) + 16 * (lane & 3);
#pragma unroll
        for (int k = 0; k < kMaxChunksPerWave; ++k) {
            const int q = wave + kW * k;
            const int r = row_of_slot(q, lane >> 2, n);
            const int ti = __builtin_amdgcn_ds_bpermute(4 * (16 * (k & 3) + (lane >> 2)), rows.v[k >> 2]);
            if (q < nchunks && r < n) {
                const double2 v = *reinterpret_cast<const double2*>(cur + q * kChunkStride + 16 * lane);
                *reinterpret_cast<double2*>(colp + (uint64_t)(uint32_t)ti * (uint64_t)(uint32_t)((uint32_t)p.ld * 8u)) = v;
            }
        }
    } else {  // LDS column reads only
        double acc = 0.0;
        if constexpr (MODE == 3) {
            const ColBases cb = col_bases(lds_addr(tile), lane, wave, (n + 19) / 20);
            for (int rep = 0; rep < 64; ++rep) {
                acc += col_sum_tm(cb, lane, n);
                asm volatile("" ::: "memory");
            }
        } else {
            const double* ob = reinterpret_cast<const double*>(tile) + wave * 1258 + 4 + K * lane + lane / 8;
            for (int rep = 0; rep < 64; ++rep) {
                double s = 0.0;
                if (K * lane < n) {
#pragma unroll
                    for (int i = 0; i < K; ++i) s += ob[i];
                }
                acc += wave_sum(s);
                asm volatile("" ::: "memory");
            }
        }
        if (acc == 1.2345e300) p.sums[c0 + wave] = acc;
    }
}

template <int MODE>
float run(const P& p, size_t lds, int64_t nblocks, const char* name, double bytes) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(tile_kernel<MODE>, dim3((unsigned)nblocks), dim3(kThreads), lds, 0, p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    if (bytes > 0) printf("%-44s %8.3f ms  %8.1f GB/s\n", name, best, bytes / best / 1e6);
    else printf("%-44s %8.3f ms\n", name, best);
    return best;
}

int main(int argc, char** argv) {
    const int64_t T = 14600, C = argc > 1 ? atoll(argv[1]) : 100000;
    const int G = 12;
    std::vector<int> gid(T);
    {
        int y = 1980, m = 0, d = 0;
        const int dm[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
        for (int64_t t = 0; t < T; ++t) {
            gid[t] = m;
            const int len = dm[m] + ((m == 1 && (y % 4 == 0)) ? 1 : 0);
            if (++d == len) {
                d = 0;
                if (++m == 12) {
                    m = 0;
                    ++y;
                }
            }
        }
    }
    std::vector<int> off(G + 1, 0), order(T);
    for (auto g : gid) off[g + 1]++;
    for (int g = 0; g < G; ++g) off[g + 1] += off[g];
    {
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (int64_t t = 0; t < T; ++t) order[cur[gid[t]]++] = (int)t;
    }
    P p{};
    double* x[3];
    int *dord, *doff;
    for (int f = 0; f < 3; ++f) CK(hipMalloc(&x[f], T * C * 8));
    CK(hipMalloc(&p.out, T * C * 8));
    CK(hipMalloc(&p.sums, 3 * G * C * 8));
    CK(hipMalloc(&dord, T * 4));
    CK(hipMalloc(&doff, (G + 1) * 4));
    CK(hipMemcpy(dord, order.data(), T * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(doff, off.data(), (G + 1) * 4, hipMemcpyHostToDevice));
    // x[f][t][c] = f * 1000 + (t % 977) + c % 61 / 64: exact in float64, sums exact
    {
        std::vector<double> h((size_t)T * 4096);
        for (int f = 0; f < 3; ++f) {
            for (int64_t t = 0; t < T; ++t)
                for (int64_t c = 0; c < 4096; ++c) h[t * 4096 + c] = f * 1000.0 + (double)(t % 977) + (double)(c % 61) / 64.0;
            for (int64_t cb = 0; cb < C; cb += 4096) {  // every 4 096-cell block gets the same pattern (C % 4096 handled by width)
                const int64_t w = cb + 4096 <= C ? 4096 : C - cb;
                CK(hipMemcpy2D(x[f] + cb, C * 8, h.data(), 4096 * 8, w * 8, T, hipMemcpyHostToDevice));
            }
            p.x[f] = x[f];
        }
    }
    p.ord = dord;
    p.off = doff;
    p.ld = C;
    p.C = C;
    p.ntiles = C / kW;
    p.G = G;
    const int64_t nblocks = 8 * ((p.ntiles + 7) / 8) * G;
    const size_t lds1 = kHeadBytes + 78 * kChunkStride;  // 81 824 B
    const size_t lds0 = kHeadBytes + 8 * 1258 * 8;       // 81 216 B
    const double bytes = 4.0 * T * C * 8;
    printf("cells %lld, blocks %lld, LDS %zu (REG) / %zu (DMA) bytes\n", (long long)C, (long long)nblocks, lds0, lds1);

    auto check = [&](const char* name) {
        std::vector<double> s((size_t)3 * G * C);
        CK(hipMemcpy(s.data(), p.sums, s.size() * 8, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int f = 0; f < 3; ++f)
            for (int g = 0; g < G; ++g) {
                double base = 0.0;
                for (int r = off[g]; r < off[g + 1]; ++r) base += f * 1000.0 + (double)(order[r] % 977);
                for (int64_t c = 0; c < C; ++c) {
                    const double e = base + (off[g + 1] - off[g]) * ((double)((c % 4096) % 61) / 64.0);
                    if (s[((size_t)f * G + g) * C + c] != e) ++bad;
                }
            }
        std::vector<double> o((size_t)C * 3);  // three rows of the output = three rows of tile 2
        int64_t badout = 0;
        const int64_t rows[3] = {0, 7777, T - 1};
        for (int k = 0; k < 3; ++k) {
            CK(hipMemcpy(o.data(), p.out + rows[k] * C, C * 8, hipMemcpyDeviceToHost));
            for (int64_t c = 0; c < C; ++c)
                if (o[c] != 2000.0 + (double)(rows[k] % 977) + (double)((c % 4096) % 61) / 64.0) ++badout;
        }
        printf("  check %-6s column sums wrong: %lld of %lld, output samples wrong: %lld\n", name, (long long)bad, (long long)(3 * G * C),
               (long long)badout);
        CK(hipMemset(p.sums, 0, 3 * G * C * 8));
        CK(hipMemset(p.out, 0, T * C * 8));
    };
    CK(hipMemset(p.sums, 0, 3 * G * C * 8));
    CK(hipMemset(p.out, 0, T * C * 8));
    run<0>(p, lds0, nblocks, "REG  regs -> ds_write rows, 2 WG/CU", bytes);
    check("REG");
    const float t4 = run<4>(p, lds0, nblocks, "LDS  column reads, cell-major rows x64", 0);
    const int strides[3] = {1040, 1032, 1040};
    const int lpcs[3] = {0, 1, 1};
    for (int v = 0; v < 3; ++v) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(kChunkStride), &strides[v], sizeof(int)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(kLpc), &lpcs[v], sizeof(int)));
        const size_t lds = kHeadBytes + 78 * (size_t)strides[v];
        printf("---- chunk stride %d, %s layout (LDS %zu B)\n", strides[v], lpcs[v] ? "lane-per-chunk" : "time-major", lds);
        run<1>(p, lds, nblocks, "DMA  global_load_lds, 2 WG/CU", bytes);
        check("DMA");
        if (v == 0) {
            run<2>(p, 2 * 78 * (size_t)strides[v] + kHeadBytes, nblocks, "DMA2 next tile in flight, 1 WG/CU", bytes);
            check("DMA2");
        }
        const float t3 = run<3>(p, lds, nblocks, "LDS  column reads x64", 0);
        printf("  per column pass (20 ds_read_b64 per lane, 8 waves): %.3f us (cell-major rows %.3f us) per workgroup-item\n",
               t3 * 1e3 / 64 / ((double)nblocks / 512), t4 * 1e3 / 64 / ((double)nblocks / 512));
    }
    return 0;
}
