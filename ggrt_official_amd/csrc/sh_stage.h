// sh_stage.h — cooperative, coalesced staging of a block's SH rows into LDS (preprocess_fwd / preprocess_bwd).
//
// A per-Gaussian SH row is 12·M bytes; read lane-per-Gaussian it would touch 64 different cache lines per load
// instruction.  The block's rows are therefore copied with float4 loads along the rows and read back from LDS
// with an odd row stride (conflict-free).  EVERY global load of a thread is issued before its first use —
// unconditional, clamped addresses, the bounds check only guards the LDS store: with the check around the load
// the compiler serialises the round trips, and 12 loads taken 4 at a time are 3 × ≈2 µs per block
// (measured at C3: preprocess_fwd 0.094 → 0.079 ms).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ggr {

// rows: `row` floats apart in global memory, the first `copy_row` floats of each are needed; LDS stride `stride`
// (== row when `flat`).  Must be called by all 256 threads of the block; ends WITHOUT a barrier.
__device__ __forceinline__ void stage_sh_rows(float* __restrict__ sh_lds, const float* __restrict__ shs, size_t g0,
                                              int nG, size_t row, int copy_row, int stride, bool flat) {
    const int tid = threadIdx.x;
    if (flat) {
        // odd row length (GGRt: 3·M = 75 floats): the block's rows are ONE contiguous, 16-B aligned region (g0 is
        // a multiple of the block size) → flat float4 copy, the odd stride is the row length itself
        const size_t total = (size_t)nG * row;
        const float* src = shs + g0 * row;
        const int n4 = (int)(total >> 2);
        // (GGRt's rows: 256 × 75 floats = 4800 float4 = 18.75 per thread: ONE trip with 19 loads in flight; with 10 per
        // trip the second trip waited for the first)
        constexpr int U = 19;
        for (int j0 = 0; j0 < n4; j0 += U * 256) {
            float4 v[U];
#pragma unroll
            for (int it = 0; it < U; it++) v[it] = ggr_ld_f4(src + 4 * (size_t)min(j0 + it * 256 + tid, n4 - 1));
#pragma unroll
            for (int it = 0; it < U; it++) {
                const int j = j0 + it * 256 + tid;
                if (j < n4) reinterpret_cast<float4*>(sh_lds)[j] = v[it];
            }
        }
        for (int j = (n4 << 2) + tid; j < (int)total; j += 256) sh_lds[j] = ggr_ld(src + j);
    } else if ((row & 3) == 0 && (copy_row & 3) == 0) {
        // 16-B aligned rows: float4 loads, repacked to the odd LDS stride with scalar stores
        const int q_per = copy_row >> 2;
        const int total4 = nG * q_per;
        for (int j0 = 0; j0 < total4; j0 += 12 * 256) {
            float4 v[12];
            int gg[12], qq[12];
#pragma unroll
            for (int it = 0; it < 12; it++) {
                const int j = min(j0 + it * 256 + tid, total4 - 1);
                gg[it] = j / q_per;
                qq[it] = j - gg[it] * q_per;
                v[it] = ggr_ld_f4(shs + (g0 + gg[it]) * row + 4 * qq[it]);
            }
#pragma unroll
            for (int it = 0; it < 12; it++) {
                if (j0 + it * 256 + tid < total4) {
                    float* d = sh_lds + gg[it] * stride + 4 * qq[it];
                    d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
                }
            }
        }
    } else {
        // rows not 16-B aligned and not flat-copyable: one wave per row, lanes along the row (contiguous 4-B
        // loads, no per-element div/mod)
        const int wv = tid >> 6, ln = tid & 63;
#pragma unroll 8
        for (int g = wv; g < nG; g += 4)
            for (int k = ln; k < copy_row; k += 64) sh_lds[g * stride + k] = ggr_ld(shs + (g0 + g) * row + k);
    }
}

// Compact variant: only the 3K floats a row contributes are kept in LDS (stride 3K|1) although the row holds 3M > 3K
// (GGRt: M = 25, K = 16 → 49 instead of 75 floats per Gaussian = 3 instead of 2 resident blocks per CU).  One wave
// per row, lanes along the row (4-B loads, 256-B coalesced segments), eight rows = 16 loads in flight.
// k-major rows keep columns < 3K; channel-major rows keep (col mod M) < K, stored as [c][K].
__device__ __forceinline__ void stage_sh_rows_compact(float* __restrict__ sh_lds, const float* __restrict__ shs,
                                                      size_t g0, int nG, int M, int K, int stride, bool channel_major) {
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const int row = 3 * M;
    // this lane's two columns and where (if anywhere) they go
    int col[2] = {ln, ln + 64}, dst[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (col[h] >= row) dst[h] = -1;
        else if (channel_major) { const int c = col[h] / M, k = col[h] - c * M; dst[h] = k < K ? c * K + k : -1; }
        else dst[h] = col[h] < 3 * K ? col[h] : -1;
    }
    const int c0 = min(col[0], row - 1), c1 = min(col[1], row - 1);
    for (int gb = wv * 8; gb < nG; gb += 32) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const size_t g = g0 + (size_t)min(gb + r, nG - 1);
            v[2 * r] = ggr_ld(shs + g * row + c0);
            v[2 * r + 1] = ggr_ld(shs + g * row + c1);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (gb + r < nG) {
                if (dst[0] >= 0) sh_lds[(gb + r) * stride + dst[0]] = v[2 * r];
                if (dst[1] >= 0) sh_lds[(gb + r) * stride + dst[1]] = v[2 * r + 1];
            }
        }
    }
}

// Inverse of the compact staging for the gradient rows: dL/dSH row g = the 3K compact LDS values at their columns,
// zero everywhere else (coefficients k ≥ K get no gradient), written with coalesced 4-B stores along the row.
__device__ __forceinline__ void write_sh_rows_compact(float* __restrict__ dL_dsh, const float* __restrict__ sh_lds,
                                                      size_t g0, int nG, int M, int K, int stride, bool channel_major) {
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const int row = 3 * M;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int col = ln + 64 * h;
        if (col >= row) continue;
        int dst;
        if (channel_major) { const int c = col / M, k = col - c * M; dst = k < K ? c * K + k : -1; }
        else dst = col < 3 * K ? col : -1;
#pragma unroll 8
        for (int g = wv; g < nG; g += 4)
            ggr_st(dL_dsh + (g0 + g) * row + col, dst >= 0 ? sh_lds[g * stride + dst] : 0.f);
    }
}

// ---- one third of every row at a time (preprocess_fwd / preprocess_bwd, one view, degree 3 / 4) ----------------------
// A whole 75-float row per Gaussian is 76.8 KB of LDS per 256-thread block = 2 blocks per CU; a third (KC floats: one
// colour channel of channel-major rows, floats [J·KC, (J+1)·KC) of k-major ones) is 25.6 KB.  Per trip the block copies
// RPI = ⌊256 / KC⌋ rows' thirds, thread t float t mod KC of row t / KC (KC = 25: 250 of the 256 threads), so addresses
// and LDS slots advance by constants from trip to trip and only the loaded values are carried (tools/sh_stage_bench.hip).
template <int KC>
struct ShThirds {
    static constexpr int STRIDE = KC | 1, RPI = 256 / KC, ITS = (256 + RPI - 1) / RPI;
    const float* src;   // the block's first row
    int g, k, last;     // this thread's row inside a trip, its float inside the third, the block's last row
    uint32_t row;       // floats per row in global memory
    int step;           // floats from one third to the next inside a row
    __device__ __forceinline__ void init(const float* block_rows, int nG, int row_floats, int third_step) {
        src = block_rows; row = (uint32_t)row_floats; step = third_step; last = nG - 1;
        g = (int)threadIdx.x / KC; k = (int)threadIdx.x - g * KC;
    }
    // request third J (uniform base + 32-bit lane offset: one address register per load; clamped rows: every load
    // is issued, rows past the block's last are not kept)
    __device__ __forceinline__ void load(float (&v)[ITS], int J) const {
        const float* s = src + J * step;
#pragma unroll
        // (plain loads on purpose: the three thirds of a row share cache lines — loaded non-temporally, every third
        //  fetched its lines from HBM again: preprocess_fwd 0.063 → 0.085 ms at C3)
        for (int it = 0; it < ITS; it++) v[it] = s[(uint32_t)min(g + it * RPI, last) * row + (uint32_t)k];
    }
    // the loaded third into LDS, row r at r·STRIDE
    __device__ __forceinline__ void store(float* lds, const float (&v)[ITS]) const {
#pragma unroll
        for (int it = 0; it < ITS; it++)
            if (g < RPI && g + it * RPI < 256) lds[(g + it * RPI) * STRIDE + k] = v[it];
    }
};

}  // namespace ggr
