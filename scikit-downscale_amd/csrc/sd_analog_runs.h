// sd_analog_runs.h -- part of the translation unit csrc/sd_analog.hip (included there, inside its unnamed namespace).
//
// Round 6: the queries of the one-feature predict kernels in VALUE ORDER (gard.py:273-364, 152-224: per query the k nearest
// training values -- the answer does not depend on the order the queries are asked in).
//
// The per-cell kernels (analog_f1_mean3_kernel, analog_f1_mean_kernel, analog_f1_window_kernel, analog_f1_fused_kernel) hand
// 64 consecutive positions of the cell-major query array to the 64 lanes of a wave.  In time order those are 64 unrelated
// values: every step of a lane's bisection through the sorted training values in LDS is a 64-lane read with unrelated
// addresses (9.5 clocks against 4 conflict-free: DESIGN 4.3), and the window of analog values a lane then reads from memory
// (weight_analogs, thresholds, AnalogRegression's residual path) is a line of its own (csrc/microbench/window_reads: 2.2 ms per
// window element and 100 000 cells against 0.43 ms when neighbouring lanes read neighbouring windows).  With the queries of a
// RUN of 1 024 consecutive time steps sorted by value, the 64 lanes hold 64 NEIGHBOURING values: the first steps of their
// bisections read the same words (broadcast), the last ones neighbouring words, and their windows overlap.
//
//   analog_query_runs_kernel     replaces the staging transpose of the queries: a 512-thread workgroup = 8 adjacent cells x one
//                                run, read as 64-byte row fragments of the time-major field (the geometry of the BCSD kernels),
//                                one wave per cell: 32-bit keys (21 bits of the value's position in the run's range above the
//                                10-bit time offset), the register sort of sd_wsort.h, values gathered through the tags, written
//                                cell-major in sorted order together with the 16-bit time offsets.  The order inside a run of
//                                equal 21-bit prefixes is arbitrary -- nothing depends on it.
//   analog_untranspose_runs_kernel  replaces the staging transpose of the outputs: the results of a run arrive in sorted order
//                                and are scattered to their time offsets inside the LDS tile that staged them anyway.
//
// Nothing else changes: the per-cell kernels index queries and results by POSITION, and position p of a cell's arrays now means
// "the p-th query of its run in value order" on both sides.  Same arithmetic per query: results are bit-identical to the
// time-ordered path (tests/test_gpu_analog.py).
#pragma once

constexpr int kRunK = 16;               // keys per lane of the run sort
constexpr int kRun = 64 * kRunK;        // queries per run: 1 024 consecutive time steps
constexpr int kRunRS = kRun + 2;        // LDS row stride in doubles (RS % 4 == 2: rows land 8 or 24 banks apart, like sd_bcsd_rs_row_stride)
constexpr int kRunRSQ = kRun + 32 + 2;  // the same for rows that hold sorted position p at slot p + p / 32 (see analog_query_runs_kernel)
__device__ __forceinline__ int run_slot(int p) { return p + (p >> 5); }
constexpr unsigned kRunQD = (1u << 21) - 2048u;

inline size_t query_runs_lds_bytes() { return sizeof(double) * ((size_t)sdw::kW * kRunRSQ + sdw::kHeadDoubles); }

__global__ void __launch_bounds__(sdw::kThreads, 4) analog_query_runs_kernel(const double* __restrict__ Xq, int64_t ld, int64_t Tq, int64_t C,
                                                                             int nruns, double* __restrict__ qs /* [C][Tq] */,
                                                                             unsigned short* __restrict__ qt /* [C][Tq] */, int32_t* status) {
    using namespace sdw;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NR = kRun / kRowsPerPass;  // rows a thread loads of the tile: 8
    double* const tile = reinterpret_cast<double*>(smem_raw) + kHeadDoubles;
    const int64_t ntiles = (C + kW - 1) / kW;
    int64_t tile_id;
    int q;
    xcd_tile_of_block(blockIdx.x, ntiles, &tile_id, &q);
    if (tile_id >= ntiles || q >= nruns) return;
    const int64_t c0 = tile_id * kW;
    const int64_t r0 = (int64_t)q * kRun;
    const int nq = (int)(Tq - r0 < kRun ? Tq - r0 : kRun);  // queries of this run (> 0)
    const int tid = tid_now();
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t cpair = c0 + 2 * cp;
    const bool vec = (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(Xq) & 15) == 0 && cpair + 1 < C;
    double x0[NR], x1[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = rr + k * kRowsPerPass;
        const double* px = Xq + (r0 + (r < nq ? r : 0)) * ld + cpair;
        if (vec) {
            const double2 v = *reinterpret_cast<const double2*>(px);
            x0[k] = v.x;
            x1[k] = v.y;
        } else {
            x0[k] = cpair < C ? px[0] : 0.0;
            x1[k] = cpair + 1 < C ? px[1] : 0.0;
        }
    }
    {
        double* d0 = tile + (2 * cp) * kRunRSQ;
        double* d1 = d0 + kRunRSQ;
        bool bad0 = false, bad1 = false;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rr + k * kRowsPerPass;
            if (r < nq) {
                bad0 |= !sd_finite(x0[k]);
                bad1 |= !sd_finite(x1[k]);
                d0[r] = x0[k];
                d1[r] = x1[k];
            }
        }
        // (what analog_transpose_kernel reports for a query series: a non-finite query flags its cell, gard.py:288 _validate_data)
        if (bad0 && cpair < C) atomicOr(&status[cpair], SDI_NONFINITE);
        if (bad1 && cpair + 1 < C) atomicOr(&status[cpair + 1], SDI_NONFINITE);
    }
    __syncthreads();
    const int64_t c = c0 + wave;
    if (c >= C) return;  // (no barrier below)
    double* const row = tile + wave * kRunRSQ;
    // ---- keys: lane l takes the time offsets l + 64 i (conflict-free reads); key = (q << 11) | offset, pads above every key ----
    double v[kRunK];
    double lo = __builtin_inf(), hi = -__builtin_inf();
#pragma unroll
    for (int i = 0; i < kRunK; ++i) {
        const int j = lane + i * kWave;
        const double x = row[j < nq ? j : 0];
        v[i] = sd_finite(x) ? x : 0.0;  // (a non-finite query sorts as 0; it is answered NaN from its exact value)
        const bool in = j < nq;
        lo = in && v[i] < lo ? v[i] : lo;
        hi = in && v[i] > hi ? v[i] : hi;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double a = __shfl_xor(lo, o, kWave), b = __shfl_xor(hi, o, kWave);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    const double sc = hi > lo ? (double)kRunQD / (hi - lo) : 0.0;
    const double off = -lo * sc;
    unsigned key[kRunK];
#pragma unroll
    for (int i = 0; i < kRunK; ++i) {
        const unsigned j = (unsigned)(lane + i * kWave);
        unsigned qq = (unsigned)__builtin_fma(v[i], sc, off);  // truncates, saturates
        qq = qq < kRunQD ? qq : kRunQD;
        key[i] = (int)j < nq ? (qq << 11) | j : ((kRunQD + 1u) << 11) | j;
    }
    sdws::wave_sort<kRunK>(key, lane, 64);  // lane l now owns the sorted positions 16 l .. 16 l + 15
    // ---- exact values through the tags; back into the row in sorted order; tags beside them ----
#pragma unroll
    for (int i = 0; i < kRunK; ++i) v[i] = row[key[i] & 1023u];
    wave_fence();  // every lane has read: the row is rewritten
    // (sorted position p at slot p + p / 32: the lanes' blocks of 16 start on different banks, and the lane-strided read below
    // stays conflict-free)
#pragma unroll
    for (int i = 0; i < kRunK; ++i) row[run_slot(kRunK * lane + i)] = v[i];
    wave_fence();
    double* dq = qs + c * Tq + r0;
#pragma unroll
    for (int i = 0; i < kRunK; ++i) {
        const int p = lane + i * kWave;
        if (p < nq) dq[p] = row[run_slot(p)];  // (the pads sorted behind the nq queries of a short last run)
    }
    // the lane's 16 tags are 32 consecutive bytes of the tag array: from the registers
    unsigned short* dt = qt + c * Tq + r0 + kRunK * lane;
    if (kRunK * lane + kRunK <= nq && (reinterpret_cast<uintptr_t>(dt) & 15) == 0) {
        typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
        u32x4 w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[h][e] = (key[8 * h + 2 * e] & 1023u) | ((key[8 * h + 2 * e + 1] & 1023u) << 16);
        reinterpret_cast<u32x4*>(dt)[0] = w[0];
        reinterpret_cast<u32x4*>(dt)[1] = w[1];
    } else {
#pragma unroll
        for (int i = 0; i < kRunK; ++i)
            if (kRunK * lane + i < nq) dt[i] = (unsigned short)(key[i] & 1023u);
    }
}

// cell-major staging [C][3][Tq] with the results of every run in its sorted order -> output field [Tq, 3, ld]: a workgroup
// stages one plane of one run of 8 adjacent cells, placing result p of the run at the time offset qt[p]
// prob_from_pred: as in analog_untranspose_kernel (the probability plane is derived from the predictions)
__global__ void __launch_bounds__(sdw::kThreads, 4) analog_untranspose_runs_kernel(const double* __restrict__ oc, const unsigned short* __restrict__ qt,
                                                                                   int64_t Tq, int64_t C, int nruns, double* __restrict__ out,
                                                                                   int64_t ld, int prob_from_pred) {
    using namespace sdw;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* const tile = reinterpret_cast<double*>(smem_raw) + kHeadDoubles;
    const int64_t ntiles = (C + kW - 1) / kW;
    int64_t tile_id;
    int q;
    xcd_tile_of_block(blockIdx.x, ntiles, &tile_id, &q);
    if (tile_id >= ntiles || q >= nruns) return;
    const int64_t c0 = tile_id * kW;
    const int64_t r0 = (int64_t)q * kRun;
    const int nq = (int)(Tq - r0 < kRun ? Tq - r0 : kRun);
    const int tid = tid_now();
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
    const int64_t c = c0 + wave;
    const int cp = tid & 3, rr = tid >> 2;
    const int64_t cpair = c0 + 2 * cp;
    const bool vec = (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && cpair + 1 < C;
    double* const row = tile + wave * kRunRS;
    const double* s0 = tile + (2 * cp) * kRunRS;
    const double* s1 = s0 + kRunRS;
    // (one plane per workgroup -- grid y --: looping over the planes inside one workgroup, the time offsets read once, was
    // measured slower, 18.5 against 14.9 ms per 100 000 cells: two more barriers per plane and a third of the workgroups)
    const int j = prob_from_pred ? 2 * (int)blockIdx.y : (int)blockIdx.y;  // 2 planes (pred [+ prob], err) or all 3
    if (c < C) {
        const double* src = oc + (c * 3 + j) * Tq + r0;
        const unsigned short* st = qt + c * Tq + r0;
        double v[kRunK];
        int t[kRunK];
#pragma unroll
        for (int i = 0; i < kRunK; ++i) {
            const int p = lane + i * kWave;
            v[i] = p < nq ? src[p] : 0.0;
            t[i] = p < nq ? (int)st[p] : -1;
        }
#pragma unroll
        for (int i = 0; i < kRunK; ++i)
            if (t[i] >= 0) row[t[i]] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRun / kRowsPerPass; ++k) {
        const int r = rr + k * kRowsPerPass;
        if (r < nq) {
            const double a = s0[r], b = s1[r];
            double* p = out + ((r0 + r) * 3 + j) * ld + cpair;
            if (vec) {
                *reinterpret_cast<double2*>(p) = make_double2(a, b);
                if (prob_from_pred && j == 0) *reinterpret_cast<double2*>(p + ld) = make_double2(a != a ? a : 1.0, b != b ? b : 1.0);
            } else {
                if (cpair < C) {
                    p[0] = a;
                    if (prob_from_pred && j == 0) p[ld] = a != a ? a : 1.0;
                }
                if (cpair + 1 < C) {
                    p[1] = b;
                    if (prob_from_pred && j == 0) p[ld + 1] = b != b ? b : 1.0;
                }
            }
        }
    }
}

// the two launches; the staging arrays of a cell chunk start at cell 0 of the chunk
inline int launch_query_runs(sd_ctx* ctx, const double* Xq, int64_t ld, int64_t Tq, int64_t C, double* qs, unsigned short* qt, int32_t* status) {
    const int nruns = (int)((Tq + kRun - 1) / kRun);
    const size_t lds = query_runs_lds_bytes();
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_query_runs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t ntiles = (C + sdw::kW - 1) / sdw::kW, tx = (ntiles + 7) / 8;
    const int64_t nblocks = 8 * tx * nruns;
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "analog predict: grid too large");
    SD_LAUNCH(ctx, "analog_query_runs_kernel", analog_query_runs_kernel, dim3((unsigned)nblocks), dim3(sdw::kThreads), lds, Xq, ld, Tq, C, nruns, qs, qt, status);
    return SD_OK;
}
inline int launch_untranspose_runs(sd_ctx* ctx, const double* oc, const unsigned short* qt, int64_t Tq, int64_t C, double* out, int64_t ld, int skip_prob) {
    const int nruns = (int)((Tq + kRun - 1) / kRun);
    const size_t lds = sizeof(double) * ((size_t)sdw::kW * kRunRS + sdw::kHeadDoubles);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&analog_untranspose_runs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t ntiles = (C + sdw::kW - 1) / sdw::kW, tx = (ntiles + 7) / 8;
    const int64_t nblocks = 8 * tx * nruns;
    SD_CHECK_ARG(nblocks < ((int64_t)1 << 31), "analog predict: grid too large");
    SD_LAUNCH(ctx, "analog_untranspose_runs_kernel", analog_untranspose_runs_kernel, dim3((unsigned)nblocks, skip_prob ? 2u : 3u), dim3(sdw::kThreads), lds, oc, qt,
              Tq, C, nruns, out, ld, skip_prob);
    return SD_OK;
}
// value-ordered runs pay from a few runs on (the window kernel's value-range passes and every per-cell kernel take them)
inline bool query_runs_apply(int64_t Tq) { return Tq >= 2 * kRun && sd_dev_env("SD_ANALOG_NORUNS") == nullptr; }
