// binning.hip — tile binning for gfx950: depth pre-sort, offsets scan, pair emission, tile sort, ranges.
//
// Replaces the scan / duplicateWithKeys / 64-bit radix sort / identifyTileRanges stages of the
// rasterizer behind reference cuda_splatting.py:114-125.
//
// MI355X-first reformulation (identical resulting order, ~5× less sort traffic):
//   the reference sorts N = Σ tiles_touched pairs by the 64-bit key (tile << 32 | depth_bits) — ≥6 radix
//   passes over N·12 B.  Here the P Gaussians are first sorted by (depth_bits) with a stable sort
//   (ties keep ascending id, 4 passes over P·8 B, P ≪ N), pairs are emitted in that order, and a
//   stable sort by tile id alone (⌈log2 tiles⌉ ≤ 16 bits → 2 passes over N·8 B) yields exactly the
//   (tile, depth, id) order of the 64-bit sort.
//
// The radix sort is hand-written for wave64: per 8-bit digit pass a histogram kernel, a row-scan
// kernel, and a scatter kernel that ranks stably with ballot-based digit matching (8 ballots per
// 64 keys) and per-wave digit counters in LDS.
#include "ggr_common.h"

namespace ggr {

// ---------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------
// block b owns keys [b*4096, (b+1)*4096); wave w of the block owns a contiguous 1024-key slice,
// round r of the wave covers 64 consecutive keys → order inside the block is (wave, round, lane).

__global__ void __launch_bounds__(GGR_SORT_THREADS)
radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n, int shift, uint32_t nblocks,
                  uint32_t* __restrict__ block_hist /*[256][nblocks]*/, uint32_t* __restrict__ totals /*[256]*/) {
    __shared__ uint32_t h[GGR_RADIX];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * GGR_SORT_TILE;
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int r = 0; r < GGR_SORT_ITEMS; r++) {
        const size_t idx = base + (size_t)wave * (64 * GGR_SORT_ITEMS) + r * 64 + lane;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & (GGR_RADIX - 1)], 1u);
    }
    __syncthreads();
    const uint32_t c = h[tid];
    block_hist[(size_t)tid * nblocks + blockIdx.x] = c;
    if (c) atomicAdd(&totals[tid], c);
}

// one block per digit: exclusive scan of that digit's row over blocks, plus the digit's global base
__global__ void __launch_bounds__(256)
radix_scan_kernel(uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ totals, uint32_t nblocks) {
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x;
    const int digit = blockIdx.x;
    // digit base = Σ totals[d' < digit]
    sh[tid] = tid < digit ? totals[tid] : 0u;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) carry = sh[0];
    __syncthreads();
    uint32_t* row = block_hist + (size_t)digit * nblocks;
    for (uint32_t start = 0; start < nblocks; start += 256) {
        const uint32_t i = start + tid;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        __syncthreads();
        sh[tid] = v;
        __syncthreads();
        // Hillis–Steele inclusive scan over 256
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t t = tid >= off ? sh[tid - off] : 0u;
            __syncthreads();
            sh[tid] += t;
            __syncthreads();
        }
        const uint32_t incl = sh[tid];
        const uint32_t c = carry;
        if (i < nblocks) row[i] = c + incl - v;
        __syncthreads();
        if (tid == 255) carry = c + incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(GGR_SORT_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int shift,
                     uint32_t nblocks, const uint32_t* __restrict__ block_offs /*[256][nblocks]*/) {
    __shared__ uint32_t wcount[4][GGR_RADIX];  // per-wave running digit counters, then per-wave bases
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int w = 0; w < 4; w++) wcount[w][tid] = 0;
    __syncthreads();

    const size_t base = (size_t)blockIdx.x * GGR_SORT_TILE + (size_t)wave * (64 * GGR_SORT_ITEMS);
    uint32_t key[GGR_SORT_ITEMS], val[GGR_SORT_ITEMS], rank[GGR_SORT_ITEMS];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < GGR_SORT_ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[r] = valid ? vals_in[idx] : 0u;
    }
    volatile uint32_t* wc = wcount[wave];
#pragma unroll
    for (int r = 0; r < GGR_SORT_ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (key[r] >> shift) & (GGR_RADIX - 1);
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < GGR_RADIX_BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        // m = valid lanes of this round holding the same digit (meaningful only when valid)
        const uint32_t before = (uint32_t)__popcll(m & lt_mask);
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t prev = 0;
        if (valid) prev = wc[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) wc[d] = prev + cnt;
        __builtin_amdgcn_wave_barrier();
        rank[r] = prev + before;
    }
    __syncthreads();
    // per-digit: turn the 4 per-wave counts into bases (global block offset + prefix over waves)
    {
        const uint32_t g = block_offs[(size_t)tid * nblocks + blockIdx.x];
        const uint32_t c0 = wcount[0][tid], c1 = wcount[1][tid], c2 = wcount[2][tid];
        __syncthreads();
        wcount[0][tid] = g;
        wcount[1][tid] = g + c0;
        wcount[2][tid] = g + c0 + c1;
        wcount[3][tid] = g + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GGR_SORT_ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        if (idx < n) {
            const uint32_t d = (key[r] >> shift) & (GGR_RADIX - 1);
            const uint32_t pos = wcount[wave][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = val[r];
        }
    }
}

void radix_sort_pairs(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                      uint32_t* hist, size_t n, int nbits, uint32_t** keys_out, uint32_t** vals_out,
                      hipStream_t s) {
    uint32_t *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
    if (n > 0) {
        const uint32_t nblocks = (uint32_t)ggr_sort_blocks(n);
        uint32_t* totals = hist + (size_t)nblocks * GGR_RADIX;
        for (int shift = 0; shift < nbits; shift += GGR_RADIX_BITS) {
            hipMemsetAsync(totals, 0, GGR_RADIX * sizeof(uint32_t), s);
            hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(GGR_SORT_THREADS), 0, s, kin, n, shift,
                               nblocks, hist, totals);
            hipLaunchKernelGGL(radix_scan_kernel, dim3(GGR_RADIX), dim3(256), 0, s, hist, totals, nblocks);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblocks), dim3(GGR_SORT_THREADS), 0, s, kin, vin,
                               kout, vout, n, shift, nblocks, hist);
            uint32_t* t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
    }
    *keys_out = kin;
    *vals_out = vin;
}

__global__ void iota_kernel(uint32_t* v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
void launch_iota(uint32_t* v, size_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n);
}

// ---------------------------------------------------------------------------------------------
// offsets scan (inclusive) of tiles_touched gathered in depth order
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// phase 0: block sums; phase 1 (single block): exclusive scan of block sums; phase 2: final scan
__global__ void __launch_bounds__(256)
scan_block_sums_kernel(const uint32_t* __restrict__ tt, const uint32_t* __restrict__ order, size_t n,
                       uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t ws[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * GGR_SORT_TILE;
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < GGR_SORT_ITEMS; r++) {
        const size_t idx = base + r * 256 + tid;
        if (idx < n) acc += tt[order[idx]];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) ws[wave] = acc;
    __syncthreads();
    if (tid == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(256)
scan_of_sums_kernel(uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < nblocks; start += 256) {
        const uint32_t i = start + tid;
        const uint32_t v = i < nblocks ? block_sums[i] : 0u;
        sh[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t t = tid >= off ? sh[tid - off] : 0u;
            __syncthreads();
            sh[tid] += t;
            __syncthreads();
        }
        const uint32_t incl = sh[tid];
        const uint32_t c = carry;
        if (i < nblocks) block_sums[i] = c + incl - v;
        __syncthreads();
        if (tid == 255) carry = c + incl;
        __syncthreads();
    }
    if (tid == 0) *total_out = carry;
}

__global__ void __launch_bounds__(256)
scan_final_kernel(const uint32_t* __restrict__ tt, const uint32_t* __restrict__ order, size_t n,
                  const uint32_t* __restrict__ block_excl, uint32_t* __restrict__ offsets) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * GGR_SORT_TILE;
    if (tid == 0) running = block_excl[blockIdx.x];
    __syncthreads();
    for (int r = 0; r < GGR_SORT_ITEMS; r++) {
        const size_t idx = base + r * 256 + tid;
        const uint32_t v = idx < n ? tt[order[idx]] : 0u;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        uint32_t pre = running;
        for (int w = 0; w < wave; w++) pre += ws[w];
        if (idx < n) offsets[idx] = pre + incl;
        __syncthreads();
        if (tid == 0) running += ws[0] + ws[1] + ws[2] + ws[3];
        __syncthreads();
    }
}

void launch_scan_tiles(const uint32_t* tiles_touched, const uint32_t* order, uint32_t* offsets,
                       uint32_t* scan_tmp, uint32_t* total_out, size_t P, hipStream_t s) {
    if (P == 0) {
        hipMemsetAsync(total_out, 0, 4, s);
        return;
    }
    const uint32_t nblocks = (uint32_t)ggr_sort_blocks(P);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblocks), dim3(256), 0, s, tiles_touched, order, P, scan_tmp);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(256), 0, s, scan_tmp, nblocks, total_out);
    hipLaunchKernelGGL(scan_final_kernel, dim3(nblocks), dim3(256), 0, s, tiles_touched, order, P, scan_tmp, offsets);
}

// ---------------------------------------------------------------------------------------------
// pair emission: for the i-th Gaussian in depth order write (tile id, Gaussian id) for every tile
// of its rect (row-major: y outer, x inner — the emission order of Appendix A.2)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
emit_pairs_kernel(size_t P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                  const uint32_t* __restrict__ tiles_touched, const uint2* __restrict__ rect, int grid_x,
                  uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = order[i];
    const uint32_t cnt = tiles_touched[g];
    if (cnt == 0) return;
    uint32_t off = offsets[i] - cnt;
    const uint2 rc = rect[g];
    const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.x >> 16, x1 = rc.y & 0xFFFFu, y1 = rc.y >> 16;
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0; x < x1; x++) {
            keys[off] = y * (uint32_t)grid_x + x;
            vals[off] = g;
            off++;
        }
}

void launch_emit_pairs(size_t P, const uint32_t* order, const uint32_t* offsets,
                       const uint32_t* tiles_touched, const uint2* rect, int grid_x, uint32_t* keys,
                       uint32_t* vals, hipStream_t s) {
    if (P == 0) return;
    hipLaunchKernelGGL(emit_pairs_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, order, offsets,
                       tiles_touched, rect, grid_x, keys, vals);
}

__global__ void __launch_bounds__(256)
tile_ranges_kernel(const uint32_t* __restrict__ keys, size_t N, uint2* __restrict__ ranges) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const uint32_t t = keys[k];
    if (k == 0) ranges[t].x = 0;
    else {
        const uint32_t p = keys[k - 1];
        if (p != t) { ranges[p].y = (uint32_t)k; ranges[t].x = (uint32_t)k; }
    }
    if (k == N - 1) ranges[t].y = (uint32_t)N;
}

void launch_tile_ranges(const uint32_t* keys_sorted, size_t N, uint2* ranges, size_t tiles, hipStream_t s) {
    hipMemsetAsync(ranges, 0, tiles * sizeof(uint2), s);
    if (N == 0) return;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, keys_sorted, N, ranges);
}

}  // namespace ggr
