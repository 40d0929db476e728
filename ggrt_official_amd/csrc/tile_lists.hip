// tile_lists.hip — builds the per-tile, depth-ordered Gaussian lists with a STABLE COUNTING SORT BY TILE
// that never materialises the unsorted (tile, id) pairs (gfx950).
//
// Replaces scan + duplicateWithKeys + the 64-bit radix sort + identifyTileRanges of the rasterizer behind
// reference cuda_splatting.py:114-125.  Same resulting lists, bit for bit:
// within a tile, entries are ordered by (depth bits, Gaussian id).
//
// Input: `order[P]` = Gaussian ids sorted by (depth bits, id) (culled ones last, they touch no tile) and
// each Gaussian's tile rect.  The position of entry (g, t) in the final list is
//        tile_start[t] + #{ Gaussians before g in `order` whose rect contains t },
// which is computed without any sort over the N = Σ tiles_touched entries:
//
//   K1 count    chunk c = 1024 consecutive positions of `order`; per (chunk, tile band) one workgroup
//               histograms the chunk's rects into LDS (ds_add, order irrelevant) → table[c][t]
//   K2a/b/c     exclusive prefix of table over chunks per tile (grouped: G groups of chunks so that the
//               scan has T·G-way parallelism), exclusive scan over tiles → tile_start, ranges, N;
//               table[c][t] becomes the ABSOLUTE list position of chunk c's first entry for tile t
//   K3 scatter  per (chunk, tile band) ONE wave walks its chunk's Gaussians in order; the band's cursors
//               live in LDS (initialised from table[c][·]); for each Gaussian its lanes (one per tile of the
//               rect) do `pos = ds_add_rtn(cursor[tile], 1)` — distinct tiles within a Gaussian, program
//               order across Gaussians, LDS executes a wave's operations in order ⇒ stable — and store the
//               id at point_list[pos].
//
// HBM traffic: table (chunks·T·4 B, 32 MB at C3) written once, read/written once, read once; rects read
// twice per band; N·4 B of ids written.  ≈ 0.25 GB instead of ≈ 0.9 GB for emit + 2 radix passes, and 5
// launches instead of 12.  A tile band is ≤ 4096 tiles (16 KB of LDS) so any image size works.
#include "ggr_common.h"

namespace ggr {

__device__ __forceinline__ void unpack_rect(uint2 rc, uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1) {
    x0 = rc.x & 0xFFFFu; y0 = rc.x >> 16; x1 = rc.y & 0xFFFFu; y1 = rc.y >> 16;
}

// ---- K1 -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bin_count_kernel(uint32_t P, const uint32_t* __restrict__ order, const uint2* __restrict__ rect, uint32_t T,
                 uint32_t band_tiles, uint32_t grid_x, uint32_t* __restrict__ table) {
    extern __shared__ uint32_t hist[];  // [band_tiles]
    const uint32_t chunk = blockIdx.x, band = blockIdx.y, tid = threadIdx.x;
    const uint32_t lo = band * band_tiles, hi = min(T, lo + band_tiles);
    for (uint32_t i = tid; i < hi - lo; i += 256) hist[i] = 0;
    __syncthreads();
    const uint32_t base = chunk * GGR_BIN_CHUNK;
    for (uint32_t k = tid; k < GGR_BIN_CHUNK; k += 256) {
        const uint32_t i = base + k;
        if (i >= P) break;
        uint32_t x0, y0, x1, y1;
        unpack_rect(rect[order[i]], x0, y0, x1, y1);
        if (x1 <= x0 || y1 <= y0) continue;
        if ((y1 - 1) * grid_x + x1 - 1 < lo || y0 * grid_x + x0 >= hi) continue;
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) {
                const uint32_t t = y * grid_x + x;
                if (t >= lo && t < hi) atomicAdd(&hist[t - lo], 1u);
            }
    }
    __syncthreads();
    for (uint32_t i = tid; i < hi - lo; i += 256) table[(size_t)chunk * T + lo + i] = hist[i];
}

// ---- K2a: per (tile, group of chunks) sum ---------------------------------------------------------
__global__ void __launch_bounds__(256)
bin_group_sum_kernel(const uint32_t* __restrict__ table, uint32_t T, uint32_t nchunks, uint32_t chunks_per_group,
                     uint32_t* __restrict__ gsum /*[G][T]*/) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (t >= T) return;
    const uint32_t c0 = g * chunks_per_group, c1 = min(nchunks, c0 + chunks_per_group);
    uint32_t s = 0;
    for (uint32_t c = c0; c < c1; c++) s += table[(size_t)c * T + t];
    gsum[(size_t)g * T + t] = s;
}

// ---- K2b: ONE block: exclusive scan over groups per tile, exclusive scan over tiles -----------------
__global__ void __launch_bounds__(1024)
bin_tile_scan_kernel(uint32_t* __restrict__ gsum, uint32_t T, uint32_t G, uint32_t* __restrict__ tile_start,
                     uint2* __restrict__ ranges, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < T; t0 += 1024) {
        const uint32_t t = t0 + tid;
        uint32_t run = 0;
        if (t < T)
            for (uint32_t g = 0; g < G; g++) {
                const uint32_t v = gsum[(size_t)g * T + t];
                gsum[(size_t)g * T + t] = run;  // exclusive prefix over groups
                run += v;
            }
        sh[tid] = run;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t v = tid >= off ? sh[tid - off] : 0u;
            __syncthreads();
            sh[tid] += v;
            __syncthreads();
        }
        const uint32_t incl = sh[tid], c = carry;
        if (t < T) {
            const uint32_t start = c + incl - run;
            tile_start[t] = start;
            ranges[t] = run ? make_uint2(start, start + run) : make_uint2(0u, 0u);
        }
        __syncthreads();
        if (tid == 1023) carry = c + incl;
        __syncthreads();
    }
    if (tid == 0) *total_out = carry;
}

// ---- K2c: table[c][t] ← absolute position of chunk c's first entry in tile t's list ------------------
__global__ void __launch_bounds__(256)
bin_group_prefix_kernel(uint32_t* __restrict__ table, uint32_t T, uint32_t nchunks, uint32_t chunks_per_group,
                        const uint32_t* __restrict__ gsum, const uint32_t* __restrict__ tile_start) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (t >= T) return;
    const uint32_t c0 = g * chunks_per_group, c1 = min(nchunks, c0 + chunks_per_group);
    uint32_t run = tile_start[t] + gsum[(size_t)g * T + t];
    for (uint32_t c = c0; c < c1; c++) {
        const uint32_t v = table[(size_t)c * T + t];
        table[(size_t)c * T + t] = run;
        run += v;
    }
}

// ---- K3 ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
bin_scatter_kernel(uint32_t P, const uint32_t* __restrict__ order, const uint2* __restrict__ rect, uint32_t T,
                   uint32_t band_tiles, uint32_t grid_x, const uint32_t* __restrict__ table,
                   uint32_t* __restrict__ point_list) {
    extern __shared__ uint32_t cursor[];  // [band_tiles]: next free list position per tile of the band
    const uint32_t chunk = blockIdx.x, band = blockIdx.y, lane = threadIdx.x;
    const uint32_t lo = band * band_tiles, hi = min(T, lo + band_tiles);
    for (uint32_t i = lane; i < hi - lo; i += 64) cursor[i] = table[(size_t)chunk * T + lo + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t base = chunk * GGR_BIN_CHUNK;
    const uint32_t end = min(P, base + GGR_BIN_CHUNK);
    for (uint32_t b0 = base; b0 < end; b0 += 64) {
        const uint32_t i = b0 + lane;
        uint32_t g = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        if (i < end) {
            g = order[i];
            unpack_rect(rect[g], x0, y0, x1, y1);
        }
        const uint32_t w = x1 > x0 ? x1 - x0 : 0, h = y1 > y0 ? y1 - y0 : 0;
        const uint32_t n = w * h;
        const bool hit = n > 0 && (y1 - 1) * grid_x + x1 - 1 >= lo && y0 * grid_x + x0 < hi;
        uint64_t mask = __ballot(hit);
        while (mask) {  // Gaussians of this 64-batch that touch the band, in order
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint32_t gj = __builtin_amdgcn_readlane(g, j);
            const uint32_t xj = __builtin_amdgcn_readlane(x0, j), yj = __builtin_amdgcn_readlane(y0, j);
            const uint32_t wj = __builtin_amdgcn_readlane(w, j), nj = __builtin_amdgcn_readlane(n, j);
            const float inv_w = 1.0f / (float)wj;
            for (uint32_t l0 = 0; l0 < nj; l0 += 64) {
                const uint32_t l = l0 + lane;
                if (l < nj) {
                    // row / column of the l-th tile of the rect: (l + ½)/w is ≥ ½/w away from any integer, far
                    // more than the fp32 error of the product for every l < 2^16
                    const uint32_t ly = (uint32_t)(((float)l + 0.5f) * inv_w);
                    const uint32_t lx = l - ly * wj;
                    const uint32_t t = (yj + ly) * grid_x + xj + lx;
                    if (t >= lo && t < hi) {
                        const uint32_t pos = atomicAdd(&cursor[t - lo], 1u);
                        point_list[pos] = gj;
                    }
                }
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------
TileListPlan plan_tile_lists(size_t P, size_t T) {
    TileListPlan p;
    p.nchunks = (uint32_t)((P + GGR_BIN_CHUNK - 1) / GGR_BIN_CHUNK);
    if (p.nchunks == 0) p.nchunks = 1;
    p.band_tiles = (uint32_t)(T < 4096 ? (T ? T : 1) : 4096);
    p.nbands = (uint32_t)((T + p.band_tiles - 1) / p.band_tiles);
    if (p.nbands == 0) p.nbands = 1;
    p.groups = p.nchunks < 32 ? p.nchunks : 32;
    p.chunks_per_group = (p.nchunks + p.groups - 1) / p.groups;
    p.groups = (p.nchunks + p.chunks_per_group - 1) / p.chunks_per_group;
    const size_t Tp = T ? T : 1;
    p.table_words = (size_t)p.nchunks * Tp;
    p.gsum_words = (size_t)p.groups * Tp;
    p.work_bytes = ggr_align(p.table_words * 4) + ggr_align(p.gsum_words * 4) + ggr_align(Tp * 4);
    return p;
}

void launch_tile_list_count(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                            const uint2* rect, void* work, uint2* ranges, uint32_t* total_out, hipStream_t s) {
    uint32_t* table = (uint32_t*)work;
    uint32_t* gsum = (uint32_t*)((char*)work + ggr_align(pl.table_words * 4));
    uint32_t* tile_start = (uint32_t*)((char*)gsum + ggr_align(pl.gsum_words * 4));
    if (T == 0) {
        (void)hipMemsetAsync(total_out, 0, 4, s);
        return;
    }
    hipLaunchKernelGGL(bin_count_kernel, dim3(pl.nchunks, pl.nbands), dim3(256), pl.band_tiles * 4, s, (uint32_t)P,
                       order, rect, (uint32_t)T, pl.band_tiles, (uint32_t)grid_x, table);
    const unsigned tb = (unsigned)((T + 255) / 256);
    hipLaunchKernelGGL(bin_group_sum_kernel, dim3(tb, pl.groups), dim3(256), 0, s, table, (uint32_t)T, pl.nchunks,
                       pl.chunks_per_group, gsum);
    hipLaunchKernelGGL(bin_tile_scan_kernel, dim3(1), dim3(1024), 0, s, gsum, (uint32_t)T, pl.groups, tile_start,
                       ranges, total_out);
    hipLaunchKernelGGL(bin_group_prefix_kernel, dim3(tb, pl.groups), dim3(256), 0, s, table, (uint32_t)T, pl.nchunks,
                       pl.chunks_per_group, gsum, tile_start);
}

void launch_tile_list_scatter(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                              const uint2* rect, const void* work, uint32_t* point_list, hipStream_t s) {
    if (T == 0 || P == 0) return;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(pl.nchunks, pl.nbands), dim3(64), pl.band_tiles * 4, s, (uint32_t)P,
                       order, rect, (uint32_t)T, pl.band_tiles, (uint32_t)grid_x, (const uint32_t*)work, point_list);
}

}  // namespace ggr
