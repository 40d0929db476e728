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
    // all of this thread's rects first (independent loads in flight together — the kernel is otherwise a
    // chain of serial ≈1–2 µs round trips; PMC showed 59 % of the wave time waiting)
    uint2 rcs[GGR_BIN_CHUNK / 256];
#pragma unroll
    for (uint32_t q = 0; q < GGR_BIN_CHUNK / 256; q++) {
        const uint32_t i = base + tid + q * 256;
        rcs[q] = i < P ? rect[i] : make_uint2(0u, 0u);  // rect_sorted: already in depth order
    }
#pragma unroll
    for (uint32_t q = 0; q < GGR_BIN_CHUNK / 256; q++) {
        uint32_t x0, y0, x1, y1;
        unpack_rect(rcs[q], x0, y0, x1, y1);
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
                     uint32_t* __restrict__ gsum /*[G][T]*/, uint32_t* __restrict__ total /*[T], zeroed*/) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (t >= T) return;
    const uint32_t c0 = g * chunks_per_group, c1 = min(nchunks, c0 + chunks_per_group);
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t c = c0; c < c1; c++) s += table[(size_t)c * T + t];
    gsum[(size_t)g * T + t] = s;
    if (s) atomicAdd(&total[t], s);  // G atomics per tile at most
}

// ---- K2b: ONE block: exclusive scan of the per-tile totals → tile_start, ranges, N --------------------
__global__ void __launch_bounds__(1024)
bin_tile_scan_kernel(uint32_t T, uint32_t* __restrict__ tile_start /*in: totals, out: starts*/,
                     uint2* __restrict__ ranges, uint32_t* __restrict__ total_out /*[0] = N, [1] = overflow*/,
                     uint32_t capacity /*entries the caller's list buffer holds (sync-free mode); ~0u = exact*/) {
    // each thread owns E CONSECUTIVE tiles (E = ⌈T/1024⌉ rounded up to a multiple of 8, ≤ 64 per slab): local
    // sums, ONE block-wide scan of the 1024 partials, then the running starts — instead of T/1024 sequential
    // 1024-wide scans (16 µs → a few µs at 8160 tiles)
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    constexpr uint32_t E = 8;
    for (uint32_t t0 = 0; t0 < T; t0 += 1024 * E) {
        const uint32_t first = t0 + tid * E;
        uint32_t cnt[E], local = 0;
#pragma unroll
        for (uint32_t e = 0; e < E; e++) {
            cnt[e] = first + e < T ? tile_start[first + e] : 0u;
            local += cnt[e];
        }
        sh[tid] = local;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t v = tid >= off ? sh[tid - off] : 0u;
            __syncthreads();
            sh[tid] += v;
            __syncthreads();
        }
        const uint32_t incl = sh[tid], c = carry;
        uint32_t start = c + incl - local;
#pragma unroll
        for (uint32_t e = 0; e < E; e++) {
            if (first + e < T) {
                tile_start[first + e] = start;
                // sync-free mode: a list that does not fit the caller's buffer is cut at its end (the overflow
                // flag tells the caller that this frame is incomplete) — nothing ever reads or writes beyond it
                const uint32_t rs = min(start, capacity), re = min(start + cnt[e], capacity);
                ranges[first + e] = re > rs ? make_uint2(rs, re) : make_uint2(0u, 0u);
            }
            start += cnt[e];
        }
        __syncthreads();
        if (tid == 1023) carry = c + incl;
        __syncthreads();
    }
    if (tid == 0) {
        total_out[0] = carry;
        total_out[1] = carry > capacity ? 1u : 0u;
    }
}

// ---- K2c: table[c][t] ← absolute position of chunk c's first entry in tile t's list ------------------
__global__ void __launch_bounds__(256)
bin_group_prefix_kernel(uint32_t* __restrict__ table, uint32_t T, uint32_t nchunks, uint32_t chunks_per_group,
                        const uint32_t* __restrict__ gsum, const uint32_t* __restrict__ tile_start) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (t >= T) return;
    const uint32_t c0 = g * chunks_per_group, c1 = min(nchunks, c0 + chunks_per_group);
    uint32_t run = tile_start[t];
#pragma unroll 8
    for (uint32_t gg = 0; gg < g; gg++) run += gsum[(size_t)gg * T + t];  // exclusive prefix over groups
    // counts first (independent loads, 8 in flight), then the running positions
    for (uint32_t cb = c0; cb < c1; cb += 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = cb + u < c1 ? table[(size_t)(cb + u) * T + t] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
            if (cb + u < c1) {
                table[(size_t)(cb + u) * T + t] = run;
                run += v[u];
            }
    }
}

// ---- K3 ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
bin_scatter_kernel(uint32_t P, const uint32_t* __restrict__ order, const uint2* __restrict__ rect, uint32_t T,
                   uint32_t band_tiles, uint32_t grid_x, const uint32_t* __restrict__ table,
                   uint32_t* __restrict__ point_list, uint32_t capacity /*entries in point_list; ~0u = exact size*/,
                   uint32_t nbands_total) {
    // LDS: cursor[band_tiles] (next free list position per tile of the band) + the compacted list of this
    // chunk's Gaussians that touch the band: id, packed origin (x0 | y0<<16), packed size (w | h<<16)
    extern __shared__ uint32_t lds[];
    uint32_t* cursor = lds;
    uint32_t* l_id = lds + band_tiles;
    uint32_t* l_xy = l_id + GGR_BIN_CHUNK;
    uint32_t* l_wh = l_xy + GGR_BIN_CHUNK;
    // XCD-affine work order.  Workgroup b runs on XCD b mod 8, and every XCD has its own L2: when the waves that
    // append to one tile list sit on different XCDs, each L2 writes back its own partial copy of every 64-B list
    // line (measured: 366 MB written for 43 MB of ids).  So a band is given to ONE XCD — band = xcd + 8·k — and the
    // chunks of a band follow each other on it; the ≤ 7 padding bands exit at once.
    const uint32_t lane = threadIdx.x;
    const uint32_t bands8 = (nbands_total + 7u) >> 3;             // bands per XCD
    const uint32_t xcd = blockIdx.x & 7u, r = blockIdx.x >> 3;
    const uint32_t band = xcd + 8u * (r % bands8), chunk = r / bands8;
    if (band >= nbands_total || band * band_tiles >= T) return;
    const uint32_t lo = band * band_tiles, hi = min(T, lo + band_tiles);
    const uint32_t base = chunk * GGR_BIN_CHUNK;
    const uint32_t end = min(P, base + GGR_BIN_CHUNK);
    // ALL global loads of this wave are issued before anything waits on them (PMC: with the loads inside the
    // loops, 52 % of a wave's life was spent in ≈19 serial 1–2 µs round trips): the band's start positions
    // (≤ 1024 tiles = 16 per lane) and the chunk's 1024 (id, rect) pairs (16 per lane).
    constexpr int NB = GGR_BIN_CHUNK / 64;
    uint32_t cur0[16], gq[NB];
    uint2 rq[NB];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t i = lane + 64 * q;
        cur0[q] = i < hi - lo ? table[(size_t)chunk * T + lo + i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const uint32_t i = base + 64 * q + lane;
        gq[q] = i < end ? order[i] : 0u;
        rq[q] = i < end ? rect[i] : make_uint2(0u, 0u);  // rect is already in depth order (rect_sorted)
    }
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t i = lane + 64 * q;
        if (i < hi - lo) cursor[i] = cur0[q];
    }
    // phase A: keep (in order) the Gaussians whose rect can touch the band
    uint32_t nh = 0;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const uint32_t g = gq[q];
        uint32_t x0, y0, x1, y1;
        unpack_rect(rq[q], x0, y0, x1, y1);
        const uint32_t w = x1 > x0 ? x1 - x0 : 0, h = y1 > y0 ? y1 - y0 : 0;
        const bool hit = w * h > 0 && (y1 - 1) * grid_x + x1 - 1 >= lo && y0 * grid_x + x0 < hi;
        const uint64_t mk = __ballot(hit);
        if (hit) {
            const uint32_t p = nh + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            l_id[p] = g;
            l_xy[p] = y0 * grid_x + x0 - lo;  // first tile of the rect RELATIVE to the band (may wrap below 0)
            l_wh[p] = w | (h << 16);
        }
        nh += (uint32_t)__popcll(mk);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // phase B: one Gaussian per step, lanes = tiles of its rect.  Distinct tiles within a step, program order
    // across steps, and LDS executes one wave's operations in order ⇒ the lists come out stable.
    // Every step is a ds_add_rtn → global store chain whose latency (not its issue cost) bounds the walk, so
    // four steps are software-pipelined: four independent ds_add_rtn back to back (still in program order),
    // then the four stores.
    constexpr int U = 4;
    for (uint32_t k0 = 0; k0 < nh; k0 += U) {
        uint32_t cg[U], bt[U], wj[U], nj[U], pos[U];
        bool in[U];
        float inv_w[U];
        const uint32_t band_n = hi - lo;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = k0 + u < nh ? k0 + u : k0;  // tail: re-read entry k0, masked out below
            cg[u] = l_id[k];
            bt[u] = l_xy[k];
            const uint32_t cwh = l_wh[k];
            wj[u] = cwh & 0xFFFFu;
            nj[u] = k0 + u < nh ? wj[u] * (cwh >> 16) : 0u;
            // v_rcp_f32 (1 ulp) is enough: (l + ½)/w is ≥ ½/w away from any integer, i.e. a relative margin
            // of ½/(l + ½) ≥ 7.6e-6 for l < 2^16 against ≈ 2.5e-7 of rcp + multiply rounding
            inv_w[u] = __builtin_amdgcn_rcpf((float)wj[u]);
        }
        if (max(max(nj[0], nj[1]), max(nj[2], nj[3])) <= 64u) {
            // all four rects fit one step each: pipeline them
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t ly = (uint32_t)(((float)lane + 0.5f) * inv_w[u]);  // row of the lane-th tile
                const uint32_t lx = lane - ly * wj[u];
                const uint32_t tr = bt[u] + ly * grid_x + lx;  // tile index relative to the band
                in[u] = lane < nj[u] && tr < band_n;           // (unsigned: also rejects tiles before the band)
                pos[u] = in[u] ? atomicAdd(&cursor[tr], 1u) : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (in[u] && pos[u] < capacity) point_list[pos[u]] = cg[u];
        } else {
            // a rect of more than 64 tiles needs several steps; all of them must precede the next entry's
            // (a later entry may share one of the tail tiles) → walk this group strictly one entry at a time
#pragma unroll
            for (int u = 0; u < U; u++) {
                for (uint32_t l0 = 0; l0 < nj[u]; l0 += 64) {
                    const uint32_t l = l0 + lane;
                    if (l < nj[u]) {
                        const uint32_t ly = (uint32_t)(((float)l + 0.5f) * inv_w[u]);
                        const uint32_t lx = l - ly * wj[u];
                        const uint32_t tr = bt[u] + ly * grid_x + lx;
                        if (tr < band_n) {
                            const uint32_t p = atomicAdd(&cursor[tr], 1u);
                            if (p < capacity) point_list[p] = cg[u];
                        }
                    }
                }
            }
        }
    }
}

// rect_sorted[i] = rect[order[i]]: lets K1 / K3 stream the rects instead of chasing order[] → rect[]
__global__ void __launch_bounds__(256)
gather_rect_kernel(uint32_t P, const uint32_t* __restrict__ order, const uint2* __restrict__ rect,
                   uint2* __restrict__ rect_sorted, uint32_t* __restrict__ tile_total, uint32_t T) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    // also clears K2a's per-tile totals (saves a fill launch)
    for (uint32_t t = i; t < T; t += gridDim.x * 256) tile_total[t] = 0u;
    if (i < P) rect_sorted[i] = rect[order[i]];
}

// ---- host side -------------------------------------------------------------------------------------
TileListPlan plan_tile_lists(size_t P, size_t T) {
    TileListPlan p;
    p.nchunks = (uint32_t)((P + GGR_BIN_CHUNK - 1) / GGR_BIN_CHUNK);
    if (p.nchunks == 0) p.nchunks = 1;
    // count: wide bands (few re-reads of the rects, order-free LDS adds); scatter: narrow bands (many short,
    // independent in-order walks instead of few long ones — every step waits for a ds_add_rtn)
    p.band_tiles = (uint32_t)(T < 4096 ? (T ? T : 1) : 4096);
    p.nbands = (uint32_t)((T + p.band_tiles - 1) / p.band_tiles);
    if (p.nbands == 0) p.nbands = 1;
    {   // aim for ≥ ~8192 independent walks (chunks × bands); a band is 64 … 1024 tiles.
        // The band COUNT is a multiple of 8: bin_scatter gives each band to one XCD (see there), so equal
        // counts per XCD keep the eight of them balanced.  (An earlier attempt at the same idea changed the
        // band sizes at the same time and showed no gain; with the sizes kept and only the block → (band, chunk)
        // map changed, C3's scatter went 0.164 → 0.115 ms.)
        const size_t Tn = T ? T : 1;
        const size_t want_bands = (8192 + p.nchunks - 1) / p.nchunks;
        size_t nb = 8 * ((want_bands + 4) / 8);                  // nearest multiple of 8 …
        if (nb < 8) nb = 8;
        while ((Tn + nb - 1) / nb > 1024) nb += 8;                // … with bands of at most 1024 tiles
        while (nb > 8 && (Tn + nb - 1) / nb < 64) nb -= 8;        // … and of at least 64 where the image allows
        p.sband_tiles = (uint32_t)((Tn + nb - 1) / nb);
        p.nsbands = (uint32_t)nb;                                // (trailing bands may be empty: they exit at once)
    }
    p.groups = p.nchunks < 32 ? p.nchunks : 32;
    p.chunks_per_group = (p.nchunks + p.groups - 1) / p.groups;
    p.groups = (p.nchunks + p.chunks_per_group - 1) / p.chunks_per_group;
    const size_t Tp = T ? T : 1;
    p.table_words = (size_t)p.nchunks * Tp;
    p.gsum_words = (size_t)p.groups * Tp;
    p.work_bytes = ggr_align(p.table_words * 4) + ggr_align(p.gsum_words * 4) + ggr_align(Tp * 4) +
                   ggr_align((P ? P : 1) * sizeof(uint2));
    return p;
}

namespace {
struct WorkArea {
    uint32_t* table;
    uint32_t* gsum;
    uint32_t* tile_start;
    uint2* rect_sorted;
};
WorkArea carve_work(const TileListPlan& pl, void* work, size_t T) {
    WorkArea w;
    char* p = (char*)work;
    w.table = (uint32_t*)p; p += ggr_align(pl.table_words * 4);
    w.gsum = (uint32_t*)p; p += ggr_align(pl.gsum_words * 4);
    w.tile_start = (uint32_t*)p; p += ggr_align((T ? T : 1) * 4);
    w.rect_sorted = (uint2*)p;
    return w;
}
}  // namespace

void tile_list_gather_targets(const TileListPlan& pl, void* work, size_t T, uint2** rect_sorted,
                              uint32_t** zero_area, uint32_t* zero_words) {
    const WorkArea w = carve_work(pl, work, T);
    *rect_sorted = w.rect_sorted;
    *zero_area = w.tile_start;
    *zero_words = (uint32_t)T;
}

void launch_tile_list_count(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                            const uint2* rect, void* work, uint2* ranges, uint32_t* total_out, uint32_t capacity,
                            hipStream_t s, bool rects_gathered) {
    const WorkArea w = carve_work(pl, work, T);
    if (T == 0 || P == 0) {
        (void)hipMemsetAsync(total_out, 0, 8, s);
        if (T) (void)hipMemsetAsync(ranges, 0, T * sizeof(uint2), s);
        return;
    }
    if (!rects_gathered)  // (ggr_forward: the depth sort's last pass has done both jobs already)
        hipLaunchKernelGGL(gather_rect_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, (uint32_t)P, order,
                           rect, w.rect_sorted, w.tile_start, (uint32_t)T);
    hipLaunchKernelGGL(bin_count_kernel, dim3(pl.nchunks, pl.nbands), dim3(256), pl.band_tiles * 4, s, (uint32_t)P,
                       order, w.rect_sorted, (uint32_t)T, pl.band_tiles, (uint32_t)grid_x, w.table);
    const unsigned tb = (unsigned)((T + 255) / 256);
    hipLaunchKernelGGL(bin_group_sum_kernel, dim3(tb, pl.groups), dim3(256), 0, s, w.table, (uint32_t)T, pl.nchunks,
                       pl.chunks_per_group, w.gsum, w.tile_start);
    hipLaunchKernelGGL(bin_tile_scan_kernel, dim3(1), dim3(1024), 0, s, (uint32_t)T, w.tile_start, ranges, total_out,
                       capacity);
    hipLaunchKernelGGL(bin_group_prefix_kernel, dim3(tb, pl.groups), dim3(256), 0, s, w.table, (uint32_t)T, pl.nchunks,
                       pl.chunks_per_group, w.gsum, w.tile_start);
}

void launch_tile_list_scatter(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                              const uint2* rect, const void* work, uint32_t* point_list, uint32_t capacity,
                              hipStream_t s) {
    (void)rect;
    if (T == 0 || P == 0) return;
    const WorkArea w = carve_work(pl, (void*)work, T);
    const size_t lds = ((size_t)pl.sband_tiles + 3 * GGR_BIN_CHUNK) * 4;
    const uint32_t bands8 = (pl.nsbands + 7u) / 8u;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(pl.nchunks * bands8 * 8u), dim3(64), lds, s, (uint32_t)P, order,
                       w.rect_sorted, (uint32_t)T, pl.sband_tiles, (uint32_t)grid_x, w.table, point_list, capacity,
                       pl.nsbands);
}

}  // namespace ggr
