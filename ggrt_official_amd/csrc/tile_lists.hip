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
//   K1 count    chunk c = 1024 consecutive positions of `order`; a workgroup takes 4 consecutive chunks of one band of
//               tile rows: per chunk the rects' corner deltas go into an LDS grid (4 order-free ds_add per
//               Gaussian) whose 2-D prefix sum is the per-tile count → table[c][t] = entries of the workgroup's
//               earlier chunks, wsum[w][t] = entries of all its chunks (+ per-group and per-tile totals by atomics)
//   K2          exclusive prefix of wsum over the workgroups per tile (grouped: 8 groups so that the scan has
//               T·8-way parallelism), exclusive scan over tiles → ranges, N (formed by every block for its own
//               256 tiles — no launch of its own); table[c][t] + wsum[c/4][t] is then the ABSOLUTE list
//               position of chunk c's first entry for tile t
//   K3 scatter  per (chunk, tile band) ONE wave walks its chunk's (Gaussian, tile) pairs in order, 64 consecutive
//               pairs ("slots") per step; the band's cursors live in LDS (initialised from table[c][·]).  Lanes
//               of a step that hold the same tile are matched through a per-tile lane-mask word (atomic OR,
//               order-free); the lowest of them reserves `count` positions with one ds_add_rtn, the others take
//               base + rank.  Program order across steps + in-order LDS ⇒ stable lists; each id is stored at
//               point_list[pos].  One XCD owns a band (its list lines are then merged in ONE L2).
//
// HBM traffic: table (chunks·T·4 B, 32 MB at C3) written once, read once; wsum (8 MB) written, read + rewritten,
// read; rects read once per count band and once per scatter band; N·4 B of ids written.  A count band is ≤ 4608
// tiles (18 KB of LDS), a scatter band ≤ 512, so any image size works.
#include "ggr_common.h"
#include <stdlib.h>
#include <algorithm>

namespace ggr {

__device__ __forceinline__ void unpack_rect(uint2 rc, uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1) {
    x0 = rc.x & 0xFFFFu; y0 = rc.x >> 16; x1 = rc.y & 0xFFFFu; y1 = rc.y >> 16;
}

// inclusive wave64 scans with DPP: four row_shr steps inside each 16-lane row, then row_bcast:15 / :31 carry the
// row totals on (lanes without a source keep the identity 0)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
    v += dpp_u32<0x111, 0xf>(v);  // row_shr:1
    v += dpp_u32<0x112, 0xf>(v);  // row_shr:2
    v += dpp_u32<0x114, 0xf>(v);  // row_shr:4
    v += dpp_u32<0x118, 0xf>(v);  // row_shr:8
    v += dpp_u32<0x142, 0xa>(v);  // row_bcast:15 → rows 1, 3
    v += dpp_u32<0x143, 0xc>(v);  // row_bcast:31 → rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
    v = max(v, dpp_u32<0x111, 0xf>(v));
    v = max(v, dpp_u32<0x112, 0xf>(v));
    v = max(v, dpp_u32<0x114, 0xf>(v));
    v = max(v, dpp_u32<0x118, 0xf>(v));
    v = max(v, dpp_u32<0x142, 0xa>(v));
    v = max(v, dpp_u32<0x143, 0xc>(v));
    return v;
}

// ---- K1: per-(chunk, tile) counts from RECT CORNERS + a 2-D prefix sum -------------------------------------------
// The number of a chunk's rects that cover tile (y, x) is the 2-D inclusive prefix sum of the corner deltas
// +1 at (y0, x0), −1 at (y0, x1), −1 at (y1, x0), +1 at (y1, x1): 4 order-free LDS adds per Gaussian and one scan of
// the band's tile grid per chunk, instead of visiting every (Gaussian, tile) pair (round 2: the scatter's slot walk
// with a ds_add per pair, ≈ 17 k wave instructions per chunk; this: ≈ 1 k).  A workgroup handles GGR_COUNT_CPG
// consecutive chunks of one band of tile ROWS and keeps, per tile, the running count over its chunks in registers:
//   table[c][t]  = entries of the workgroup's EARLIER chunks in tile t        (exclusive prefix inside the workgroup)
//   wsum[w][t]   = entries of all its chunks                                  (K2 turns it into the absolute base)
// so that K2 sweeps the 8 MB of wsum instead of the 32 MB table (round 2: 27 + 8 + 15.5 µs → 21 + 5.4 + 11.9 µs at C3).
// GGR_COUNT_SLOTS (ggr_common.h) = (row, 64-tile piece) pairs per wave: bounds the band, = registers for the running counts
__global__ void __launch_bounds__(256)
bin_count_kernel(uint32_t P, const uint2* __restrict__ rect /*in depth order*/, uint32_t T, uint32_t grid_x,
                 uint32_t rows_total, uint32_t band_rows, uint32_t col_w, uint32_t nchunks, uint32_t* __restrict__ table,
                 uint32_t* __restrict__ wsum) {
    extern __shared__ uint32_t grid[];  // [band_rows][gw] corner deltas → column prefixes
    GGR_CRITICAL_PRIO();
    const uint32_t w = blockIdx.x, band = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t r0 = band * band_rows, r1 = min(rows_total, r0 + band_rows), nr = r1 - r0;
    // the band's window of tile columns [c0, c1): the whole row (blockIdx.z = 0, col_w ≥ grid_x) unless a row has more
    // 64-tile pieces than a count wave has slots — then the row is cut into windows of col_w tiles, each counted like a
    // band of its own: a rect is clipped to the window exactly as it is clipped to the band's rows
    const uint32_t c0 = blockIdx.z * col_w, c1 = min(grid_x, c0 + col_w), gw = c1 - c0;
    const uint32_t cells = nr * gw, tile0 = r0 * grid_x + c0;
    const uint32_t pieces = (gw + 63u) >> 6;
    // slot k of this wave = (row wave + 4·(k / pieces), tiles [64·(k mod pieces), +64) of it); lane = tile inside the piece
    uint32_t run[GGR_COUNT_SLOTS];
#pragma unroll
    for (int k = 0; k < GGR_COUNT_SLOTS; k++) run[k] = 0u;
    constexpr uint32_t PER_THREAD = GGR_BIN_CHUNK / 256;
    const uint32_t c_first = w * GGR_COUNT_CPG, c_end = min(nchunks, c_first + GGR_COUNT_CPG);
    uint2 rc[PER_THREAD];
#pragma unroll
    for (uint32_t q = 0; q < PER_THREAD; q++) {
        const uint32_t i = c_first * GGR_BIN_CHUNK + q * 256 + tid;
        rc[q] = i < P ? rect[i] : make_uint2(0u, 0u);
    }
    for (uint32_t c = c_first; c < c_end; c++) {
        for (uint32_t i = tid; i < cells; i += 256) grid[i] = 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PER_THREAD; q++) {
            uint32_t x0, y0, x1, y1;
            unpack_rect(rc[q], x0, y0, x1, y1);
            const uint32_t ya = max(y0, r0), yb = min(y1, r1);
            const uint32_t xa = max(x0, c0), xb = min(x1, c1);
            if (xb > xa && yb > ya) {  // (corners on the band's far edges would only feed cells outside it)
                uint32_t* top = grid + (ya - r0) * gw;
                atomicAdd(top + (xa - c0), 1u);
                if (xb < c1) atomicAdd(top + (xb - c0), 0xFFFFFFFFu);
                if (yb < r1) {
                    uint32_t* bot = grid + (yb - r0) * gw;
                    atomicAdd(bot + (xa - c0), 0xFFFFFFFFu);
                    if (xb < c1) atomicAdd(bot + (xb - c0), 1u);
                }
            }
        }
        // the next chunk's rects travel while this one is scanned
        if (c + 1 < c_end) {
#pragma unroll
            for (uint32_t q = 0; q < PER_THREAD; q++) {
                const uint32_t i = (c + 1) * GGR_BIN_CHUNK + q * 256 + tid;
                rc[q] = i < P ? rect[i] : make_uint2(0u, 0u);
            }
        }
        __syncthreads();
        // prefix along y, in place: one thread per column (consecutive lanes, consecutive words), 8 rows' loads in
        // flight at a time (one LDS round trip per 8 rows instead of one per row)
        for (uint32_t x = tid; x < gw; x += 256) {
            uint32_t acc = 0u;
            for (uint32_t rb = 0; rb < nr; rb += 8) {
                uint32_t v[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) v[u] = rb + u < nr ? grid[(rb + u) * gw + x] : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    acc += v[u];
                    if (rb + u < nr) grid[(rb + u) * gw + x] = acc;
                }
            }
        }
        __syncthreads();
        // prefix along x straight into the table: a wave per row, 64 tiles per piece, every slot's loads issued first
        uint32_t* trow = table + (size_t)c * T + tile0;
        uint32_t v[GGR_COUNT_SLOTS];
#pragma unroll
        for (int k = 0; k < GGR_COUNT_SLOTS; k++) {
            const uint32_t row = wave + 4u * ((uint32_t)k / pieces), x = (((uint32_t)k % pieces) << 6) + lane;
            v[k] = (row < nr && x < gw) ? grid[row * gw + x] : 0u;
        }
        uint32_t carry = 0u;
#pragma unroll
        for (int k = 0; k < GGR_COUNT_SLOTS; k++) {
            const uint32_t piece = (uint32_t)k % pieces;
            const uint32_t row = wave + 4u * ((uint32_t)k / pieces), x = (piece << 6) + lane;
            if (piece == 0u) carry = 0u;
            const uint32_t incl = wave_scan_add(v[k]) + carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (row < nr && x < gw) {
                trow[row * grid_x + x] = run[k];
                run[k] += incl;
            }
        }
        __syncthreads();  // (the next chunk clears the grid)
    }
    const size_t wo = (size_t)w * T + tile0;
#pragma unroll
    for (int k = 0; k < GGR_COUNT_SLOTS; k++) {
        const uint32_t row = wave + 4u * ((uint32_t)k / pieces), x = (((uint32_t)k % pieces) << 6) + lane;
        if (row < nr && x < gw) {
            const uint32_t i = row * grid_x + x;
            wsum[wo + i] = run[k];
        }
    }
}

// ---- K2a: per (tile, group of count workgroups) sum; per-tile totals ----------------------------------------------
// (Folded into K1 as one atomic pair per workgroup and tile it cost 17 µs of atomic line transactions at C3 — 4 M lanes,
//  0.25 M lines — against 7 µs for this launch, which reads the 8 MB of wsum once.)
__global__ void __launch_bounds__(256)
bin_group_sum_kernel(const uint32_t* __restrict__ wsum, uint32_t T, uint32_t nw, uint32_t wpg,
                     uint32_t* __restrict__ gsum /*[G][T]*/, uint32_t* __restrict__ total /*[T], zeroed*/) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (t >= T) return;
    const uint32_t w0 = g * wpg, w1 = min(nw, w0 + wpg);
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t w = w0; w < w1; w++) s += wsum[(size_t)w * T + t];
    gsum[(size_t)g * T + t] = s;
    if (s) atomicAdd(&total[t], s);  // G atomics per tile at most
}

// ---- K2b: wsum[w][t] ← absolute list position of workgroup w's first entry in tile t; tile ranges; N -------------------
// Every block forms the start of ITS 256 tiles itself: Σ totals of all tiles before them (≤ 32 KB from L2, 32 loads per
// thread in flight) + a block scan of its own (a single-block scan launch cost ≈ 11 µs for ≈ 3 µs of work).  The blocks
// of group 0 write the tile ranges; block (0, 0) — dispatched first — is the one that owns the LAST tiles, so N reaches
// the host's pinned word while the rest of the launch is still running.
__global__ void __launch_bounds__(256)
bin_group_prefix_kernel(uint32_t* __restrict__ wsum, uint32_t T, uint32_t nw, uint32_t wpg,
                        const uint32_t* __restrict__ gsum, const uint32_t* __restrict__ total /*[T] per-tile totals (K1)*/,
                        uint2* __restrict__ ranges, uint32_t* __restrict__ total_out /*[0] = N, [1] = overflow | fault*/,
                        uint32_t capacity /*entries the caller's list buffer holds (sync-free mode); ~0u = exact*/,
                        uint32_t* __restrict__ host_total /*pinned host word that also receives N, or NULL*/,
                        const uint32_t* __restrict__ sort_fault /*the depth sort's look-back timeout word, or NULL*/,
                        int ranges_only /*1: only (re)write the tile ranges — the repair of a hinted forward whose guess
                                          did not hold (api.hip): wsum already holds positions and is left alone*/,
                        uint32_t list_limit /*per-tile depth sort (tile_sort.hip): the longest list it can take; a longer one
                                              raises bit 3 of the status word.  ~0u: no limit (global depth sort)*/) {
    __shared__ uint32_t w_part[4], w_own[4], w_max[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t bx = gridDim.x - 1u - blockIdx.x, g = blockIdx.y;
    const uint32_t first = bx * 256u, t = first + tid;
    // totals of all tiles in front of this block's (independent loads, 8 in flight per trip)
    uint32_t part = 0, longest = 0;   // (longest: only the block that owns the last tiles has seen every tile)
    for (uint32_t i0 = 0; i0 < first; i0 += 8 * 256) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) { const uint32_t i = i0 + u * 256 + tid; v[u] = i < first ? total[i] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) { part += v[u]; longest = max(longest, v[u]); }
    }
    const uint32_t own = t < T ? total[t] : 0u;
    const uint32_t incl = wave_scan_add(own), psum = wave_scan_add(part);
    const uint32_t lmax = wave_scan_max(max(longest, own));
    if (lane == 63u) { w_own[wave] = incl; w_part[wave] = psum; w_max[wave] = lmax; }
    __syncthreads();
    uint32_t base = w_part[0] + w_part[1] + w_part[2] + w_part[3];
    uint32_t before = 0;
#pragma unroll
    for (uint32_t ww = 0; ww < 4; ww++) before += ww < wave ? w_own[ww] : 0u;
    const uint32_t start = base + before + incl - own;  // tile t's list starts here
    if (g == 0 && t < T) {
        // sync-free mode: a list that does not fit the caller's buffer is cut at its end (the overflow
        // flag tells the caller that this frame is incomplete) — nothing ever reads or writes beyond it
        const uint32_t rs = min(start, capacity), re = min(start + own, capacity);
        ranges[t] = re > rs ? make_uint2(rs, re) : make_uint2(0u, 0u);
    }
    if (ranges_only) return;
    if (g == 0 && bx == gridDim.x - 1u && tid == 0) {
        const uint32_t n_all = base + w_own[0] + w_own[1] + w_own[2] + w_own[3];
        // a look-back spin of the depth sort that ran into its bound leaves a mis-sorted order behind: the frame
        // must not be used.  Status word: bit 0 = list overflow, bit 1 = look-back spin ran into its bound, bit 2 = a
        // key beyond the sort's 30 bits (the sort's fault word shifted up by one); the exact mode's host word
        // carries GGR_HOST_FAULT_SPIN / _RANGE instead of N (ggr_forward fails with GGR_E_HIP / GGR_E_LIMIT on them).
        const uint32_t fault_word = sort_fault ? *sort_fault : 0u;
        const uint32_t fault = fault_word & 3u;
        // (the depth sort's bucket form left a bucket unsorted: the lists are complete but not in depth order everywhere — the
        //  host sorts again in three passes; status bit 4, and bit 31 of the host's second word)
        const uint32_t bucket = (fault_word & GGR_FAULT_BUCKET) ? 1u : 0u;
        // … or sorted them all, but many of them the slow way (depths concentrated in a small part of the frame's range):
        // advisory — status bit 5, bit 30 of the host's second word
        const uint32_t slow = (sort_fault && sort_fault[GGR_HIST_MSD_BIG - GGR_HIST_FAULT] >= GGR_MSD_BIG_MANY) ? 1u : 0u;
        // the frame's longest tile list: what the per-tile depth sort must be able to hold (status bit 3: it cannot; the
        // host's second pinned word carries the length itself, written BEFORE the release store of N)
        const uint32_t n_longest = max(max(w_max[0], w_max[1]), max(w_max[2], w_max[3]));
        total_out[0] = n_all;
        total_out[1] = (n_all > capacity ? 1u : 0u) | (fault << 1) | (n_longest > list_limit ? 8u : 0u) | (bucket << 4) | (slow << 5);
        total_out[2] = n_longest;
        total_out[3] = 0u;   // (the per-tile depth sort counts the entries of its slow route here: tile_sort.h)
        if (host_total) {
            host_total[1] = n_longest | (bucket << 31) | (slow << 30);
            __hip_atomic_store(host_total, (fault & 1u) ? GGR_HOST_FAULT_SPIN : fault ? GGR_HOST_FAULT_RANGE : n_all,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (t >= T) return;
    const uint32_t w0 = g * wpg, w1 = min(nw, w0 + wpg);
    uint32_t run = start;
#pragma unroll 8
    for (uint32_t gg = 0; gg < g; gg++) run += gsum[(size_t)gg * T + t];  // exclusive prefix over groups
    // counts first (independent loads, 8 in flight), then the running positions
    for (uint32_t wb = w0; wb < w1; wb += 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = wb + u < w1 ? wsum[(size_t)(wb + u) * T + t] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
            if (wb + u < w1) {
                wsum[(size_t)(wb + u) * T + t] = run;
                run += v[u];
            }
    }
}

// ---- K3 ------------------------------------------------------------------------------------------
// PAIRS (id order, for the per-tile depth sort — tile_sort.hip): the chunk is a run of Gaussian ids (no `order`), and every
// list entry is written as (id, depth key) into `pair_list` — the sort then reads a tile's entries in one coalesced sweep
// instead of gathering a key per id (7.9 M random 4-B reads at C3: 16 of its 68 µs)
template <bool PAIRS>
__global__ void __launch_bounds__(64)
bin_scatter_kernel(uint32_t P, const uint32_t* __restrict__ order, const uint2* __restrict__ rect, uint32_t T,
                   uint32_t band_tiles, uint32_t grid_x, const uint32_t* __restrict__ table,
                   const uint32_t* __restrict__ wsum /*[workgroup of K1][T]: absolute base of the chunk's count workgroup*/,
                   uint32_t* __restrict__ point_list, uint32_t capacity /*entries in point_list; ~0u = exact size*/,
                   uint32_t nbands_total, const uint32_t* __restrict__ keys /*PAIRS: depth key per Gaussian*/,
                   uint2* __restrict__ pair_list /*PAIRS: [capacity] (id, key) instead of point_list*/) {
    // LDS: cursor[band_tiles] (next free list position per tile of the band) + the compacted list of this
    // chunk's Gaussians that touch the band: id, first tile relative to the band, rect width, first slot;
    // mark[64]: scratch of one step
    extern __shared__ uint32_t lds[];
    unsigned long long* same = (unsigned long long*)lds;  // [band_tiles] lane masks: who holds tile t in this step
    uint32_t* cursor = lds + 2 * band_tiles;
    uint32_t* l_id = cursor + band_tiles;
#ifndef GGR_SCATTER_PARTS
#define GGR_SCATTER_PARTS 4
#endif
    constexpr uint32_t HALF = GGR_BIN_CHUNK / GGR_SCATTER_PARTS;  // the chunk is walked in parts (4: 4 KB of list instead of 16):
    uint32_t* l_xy = l_id + HALF;                 // more resident waves per CU for a walk whose steps are chains
    uint32_t* l_wh = l_xy + HALF;                 // of LDS round trips (1 / 2 / 4 parts at C3: 0.092 / 0.075 / 0.072 ms)
    uint32_t* l_pre = l_wh + HALF;
    uint32_t* mark = l_pre + HALF;
    uint32_t* l_key = mark + 64;                  // (PAIRS only: the launch sizes the LDS for it)
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
    // (≤ 512 tiles = 8 per lane) and the chunk's 1024 (id, rect) pairs (16 per lane).
    constexpr int NB = GGR_BIN_CHUNK / 64;
    uint32_t cur0[8], gq[NB];   // (PAIRS: gq holds the KEYS — the id of position i is i)
    uint2 rq[NB];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t i = lane + 64 * q;
        cur0[q] = i < hi - lo ? table[(size_t)chunk * T + lo + i] + wsum[(size_t)(chunk / GGR_COUNT_CPG) * T + lo + i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const uint32_t i = base + 64 * q + lane;
        gq[q] = i < end ? (PAIRS ? keys[i] : order[i]) : 0u;
        rq[q] = i < end ? rect[i] : make_uint2(0u, 0u);  // rect is already in walk order (rect_sorted, or the rects themselves)
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t i = lane + 64 * q;
        if (i < hi - lo) { cursor[i] = cur0[q]; same[i] = 0ull; }
    }
    // phase A: keep (in order) the Gaussians whose rect can touch the band, with their rows clipped to the band's,
    // and number the (Gaussian, tile) pairs of the whole chunk consecutively: pair j of hit k is SLOT pre[k] + j
    const uint32_t band_n = hi - lo;
    const uint32_t row_lo = lo / grid_x, row_hi = (hi - 1) / grid_x + 1;
#pragma unroll
  for (int half = 0; half < GGR_SCATTER_PARTS; half++) {
    uint32_t nh = 0, S = 0;
#pragma unroll
    for (int qq = 0; qq < NB / GGR_SCATTER_PARTS; qq++) {
        const int q = half * (NB / GGR_SCATTER_PARTS) + qq;
        const uint32_t g = PAIRS ? base + 64u * (uint32_t)q + lane : gq[q];
        uint32_t x0, y0, x1, y1;
        unpack_rect(rq[q], x0, y0, x1, y1);
        const uint32_t ya = max(y0, row_lo), yb = min(y1, row_hi);
        const uint32_t w = x1 > x0 ? x1 - x0 : 0, h = yb > ya ? yb - ya : 0;
        const uint32_t n = w * h;
        const bool hit = n > 0 && (yb - 1) * grid_x + x1 - 1 >= lo && ya * grid_x + x0 < hi;
        const uint32_t incl = wave_scan_add(hit ? n : 0u);
        const uint64_t mk = __ballot(hit);
        if (hit) {
            const uint32_t p = nh + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            l_id[p] = g;
            if (PAIRS) l_key[p] = gq[q];
            l_xy[p] = ya * grid_x + x0 - lo;  // first tile of the clipped rect RELATIVE to the band (may wrap below 0)
            l_wh[p] = w;
            l_pre[p] = S + incl - n;
        }
        nh += (uint32_t)__popcll(mk);
        S += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // phase B: 64 consecutive slots per step, whichever Gaussians they belong to (a rect of several hundred tiles
    // simply spans several steps).  The order inside a tile's list is (Gaussian, i.e. slot) order:
    //   * steps run in program order and LDS executes one wave's operations in order;
    //   * inside a step, lanes holding the SAME tile are found with a ballot match over the tile index bits
    //     (the radix sort's idiom): the lowest of them reserves `count` list positions with ONE ds_add_rtn and
    //     the others take base + (number of matching lanes below them) — no reliance on how the LDS orders
    //     conflicting atomics.
    // ≈ S/64 steps (19 at C3) instead of one step per Gaussian (≈ 170), each with two dependent LDS round trips.
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t carry1 = 0;  // (hit that owns the previous step's last slot) + 1;  0 = none yet
    for (uint32_t s0 = 0; s0 < S; s0 += 64) {
        const uint32_t s = s0 + lane;
        // owners: the ≤ 64 hits that START inside this step mark their first lane, a running maximum spreads them
        mark[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t kc = carry1 + lane;  // candidate: the (lane+1)-th hit after the carried one
        if (kc < nh) {
            const uint32_t pk = l_pre[kc];
            if (pk < s0 + 64u) mark[pk - s0] = kc + 1u;  // pk ≥ s0: the carried hit owns slot s0 - 1
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t own1 = mark[lane];
        if (lane == 0) own1 = max(own1, carry1);
        own1 = wave_scan_max(own1);
        carry1 = (uint32_t)__builtin_amdgcn_readlane((int)own1, 63);
        const bool act = s < S;
        const uint32_t k = own1 - 1u;  // (own1 ≥ 1: slot 0 is the start of hit 0)
        const uint32_t j = s - l_pre[k], w = l_wh[k], cg = l_id[k], bt = l_xy[k];
        // v_rcp_f32 (1 ulp) is enough: (j + ½)/w is ≥ ½/w away from any integer, i.e. a relative margin of
        // ½/(j + ½) ≥ 7.6e-6 for j < 2^16 (a clipped rect has at most band + 2 rows of tiles) against ≈ 2.5e-7
        const uint32_t ly = (uint32_t)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)w));
        const uint32_t lx = j - ly * w;
        const uint32_t tr = bt + ly * grid_x + lx;       // tile index relative to the band
        const bool in = act && tr < band_n;               // (unsigned: also rejects tiles before the band)
        // m = the lanes of this step that hold the same tile: an atomic OR of lane bits into the tile's mask word
        // (order-free), read back, cleared again — three LDS operations instead of a 10-round ballot match
        // (≈ 80 VALU instructions in a loop that is VALU-bound: 172 steps per SIMD at C3)
        uint64_t m = 0ull;
        if (in) atomicOr(&same[tr], 1ull << lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (in) m = same[tr];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (in) same[tr] = 0ull;
        const uint32_t before = (uint32_t)__popcll(m & lt_mask);
        uint32_t p0 = 0u;
        if (in && before == 0u) p0 = atomicAdd(&cursor[tr], (uint32_t)__popcll(m));
        const int leader = in ? (int)__builtin_ctzll(m) : (int)lane;
        p0 = (uint32_t)__shfl((int)p0, leader);
        const uint32_t pos = p0 + before;
        if (PAIRS) { if (in && pos < capacity) pair_list[pos] = make_uint2(cg, l_key[k]); }
        else if (in && pos < capacity) point_list[pos] = cg;
    }
    // the next half's list overwrites this one: order this half's LDS reads before those writes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
    // scatter: narrow bands (many short, independent in-order walks instead of few long ones — every step waits for
    // a ds_add_rtn); the count kernel's bands are whole tile rows, chosen at launch (it needs grid_x)
    {   // aim for ≥ ~8192 independent walks (chunks × bands); a band is 64 … 512 tiles.
        // The band COUNT is a multiple of 8: bin_scatter gives each band to one XCD (see there), so equal
        // counts per XCD keep the eight of them balanced.  (An earlier attempt at the same idea changed the
        // band sizes at the same time and showed no gain; with the sizes kept and only the block → (band, chunk)
        // map changed, C3's scatter went 0.164 → 0.115 ms.)
        const size_t Tn = T ? T : 1;
        const size_t want_bands = (8192 + p.nchunks - 1) / p.nchunks;
        size_t nb = 8 * ((want_bands + 4) / 8);                  // nearest multiple of 8 …
        if (nb < 8) nb = 8;
        while ((Tn + nb - 1) / nb > 512) nb += 8;                 // … with bands of at most 512 tiles: a 1080p frame then
        // has 16 bands, two per XCD, interleaved — with 8 bands of 1020 tiles a frame whose upper half is empty left
        // four XCDs without work (scatter 0.079 → 0.166 ms); the uniform frame is as fast either way (0.079 / 0.077)
        while (nb > 8 && (Tn + nb - 1) / nb < 64) nb -= 8;        // … and of at least 64 where the image allows
#ifdef GGR_DEV_SCATTER_BANDS  // dev builds only (GGR_EXTRA_HIPCC_FLAGS=-DGGR_DEV_SCATTER_BANDS=24): force the band count
        { const size_t v = GGR_DEV_SCATTER_BANDS; if (v >= 8 && v % 8 == 0 && (Tn + v - 1) / v <= 512) nb = v; }
#endif
        p.sband_tiles = (uint32_t)((Tn + nb - 1) / nb);
        p.nsbands = (uint32_t)nb;                                // (trailing bands may be empty: they exit at once)
    }
    p.nw = (p.nchunks + GGR_COUNT_CPG - 1) / GGR_COUNT_CPG;
    p.wpg = (p.nw + GGR_COUNT_GROUPS - 1) / GGR_COUNT_GROUPS;
    p.groups = (p.nw + p.wpg - 1) / p.wpg;
    const size_t Tp = T ? T : 1;
    p.table_words = (size_t)p.nchunks * Tp;
    p.wsum_words = (size_t)p.nw * Tp;
    p.gsum_words = (size_t)p.groups * Tp;
    // [table | wsum | gsum, total (cleared together by the depth sort's last pass) | rect_sorted]
    p.work_bytes = ggr_align(p.table_words * 4) + ggr_align(p.wsum_words * 4) + ggr_align((p.gsum_words + Tp) * 4) +
                   ggr_align((P ? P : 1) * sizeof(uint2));
    return p;
}

namespace {
struct WorkArea {
    uint32_t* table;
    uint32_t* wsum;
    uint32_t* gsum;   // [groups][T], then …
    uint32_t* total;  // … [T]: both accumulated with atomics by K1, so both start from zero
    uint2* rect_sorted;
};
WorkArea carve_work(const TileListPlan& pl, void* work, size_t T) {
    WorkArea w;
    char* p = (char*)work;
    w.table = (uint32_t*)p; p += ggr_align(pl.table_words * 4);
    w.wsum = (uint32_t*)p; p += ggr_align(pl.wsum_words * 4);
    w.gsum = (uint32_t*)p; w.total = w.gsum + pl.gsum_words; p += ggr_align((pl.gsum_words + (T ? T : 1)) * 4);
    w.rect_sorted = (uint2*)p;
    return w;
}
}  // namespace

void tile_list_gather_targets(const TileListPlan& pl, void* work, size_t T, uint2** rect_sorted,
                              uint32_t** zero_area, uint32_t* zero_words) {
    const WorkArea w = carve_work(pl, work, T);
    *rect_sorted = w.rect_sorted;
    *zero_area = w.gsum;
    *zero_words = (uint32_t)(pl.gsum_words + T);
}

void launch_tile_list_count(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                            const uint2* rect, void* work, uint2* ranges, uint32_t* total_out, uint32_t capacity,
                            hipStream_t s, bool rects_gathered, uint32_t* host_total, hipEvent_t after_scan,
                            const uint32_t* sort_fault, bool id_order, uint32_t list_limit) {
    const WorkArea w = carve_work(pl, work, T);
    if (T == 0 || P == 0) {
        (void)hipMemsetAsync(total_out, 0, 8, s);
        if (T) (void)hipMemsetAsync(ranges, 0, T * sizeof(uint2), s);
        return;
    }
    // id_order (per-tile depth sort, tile_sort.hip): the chunks are runs of Gaussian ids — the rects are read where preprocess_fwd
    // left them, and it has cleared the totals
    const uint2* rect_walk = id_order ? rect : w.rect_sorted;
    if (!rects_gathered && !id_order)  // (ggr_forward: the depth sort's last pass has done both jobs already)
        hipLaunchKernelGGL(gather_rect_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, (uint32_t)P, order,
                           rect, w.rect_sorted, w.gsum, (uint32_t)(pl.gsum_words + T));
    // count bands: whole tile rows; a wave holds the running counts of its (row, 64-tile piece) slots in registers, so a
    // band has at most 4·⌊GGR_COUNT_SLOTS / pieces⌋ rows — and few enough that the launch has ≳ 1000 workgroups (the
    // kernel is a chain of LDS round trips and barriers: it needs several workgroups per CU).  1080p, 1 M Gaussians:
    // 245 workgroups of 4 chunks × 4 bands of 17 rows.
    const uint32_t gx = (uint32_t)grid_x, rows = (uint32_t)(T / gx);
    // A row of more than 64·GGR_COUNT_SLOTS tiles (12 288 px) does not fit a count wave's slots: it is cut into windows of
    // equal width (a multiple of 64 tiles, ≤ 768), each counted like a band of its own (blockIdx.z).  The reference has no
    // width limit; until round 5 this build refused such frames with GGR_E_LIMIT.
    const uint32_t pieces_row = (gx + 63u) / 64u;
    const uint32_t ncols = (pieces_row + GGR_COUNT_SLOTS - 1) / GGR_COUNT_SLOTS;
    const uint32_t col_w = ncols == 1 ? gx : 64u * ((pieces_row + ncols - 1) / ncols);
    const uint32_t pieces = (std::min(col_w, gx) + 63u) / 64u;
    const uint32_t max_rows = 4u * (GGR_COUNT_SLOTS / pieces);
    uint32_t nbands = (rows + max_rows - 1) / max_rows;
    const uint32_t want = (1000u + pl.nw * ncols - 1) / (pl.nw * ncols);
    if (nbands < want) nbands = want < rows ? want : rows;
    const uint32_t band_rows = (rows + nbands - 1) / nbands;
    nbands = (rows + band_rows - 1) / band_rows;
    hipLaunchKernelGGL(bin_count_kernel, dim3(pl.nw, nbands, ncols), dim3(256), (size_t)band_rows * std::min(col_w, gx) * 4, s,
                       (uint32_t)P, rect_walk, (uint32_t)T, gx, rows, band_rows, col_w, pl.nchunks, w.table, w.wsum);
    const unsigned tb = (unsigned)((T + 255) / 256);
    hipLaunchKernelGGL(bin_group_sum_kernel, dim3(tb, pl.groups), dim3(256), 0, s, w.wsum, (uint32_t)T, pl.nw, pl.wpg,
                       w.gsum, w.total);
    hipLaunchKernelGGL(bin_group_prefix_kernel, dim3(tb, pl.groups), dim3(256), 0, s, w.wsum, (uint32_t)T, pl.nw,
                       pl.wpg, w.gsum, w.total, ranges, total_out, capacity, host_total, sort_fault, 0, list_limit);
    if (after_scan) (void)hipEventRecord(after_scan, s);  // (N is in the host word long before: written by the launch's first block)
}

// the tile ranges once more, uncut: after a hinted forward's guess did not hold they are cut at the guessed capacity
void launch_tile_list_ranges(const TileListPlan& pl, size_t T, void* work, uint2* ranges, hipStream_t s) {
    if (T == 0) return;
    const WorkArea w = carve_work(pl, work, T);
    hipLaunchKernelGGL(bin_group_prefix_kernel, dim3((unsigned)((T + 255) / 256), 1), dim3(256), 0, s, w.wsum, (uint32_t)T,
                       pl.nw, pl.wpg, w.gsum, w.total, ranges, (uint32_t*)nullptr, 0xFFFFFFFFu, (uint32_t*)nullptr,
                       (const uint32_t*)nullptr, 1, 0xFFFFFFFFu);
}

void launch_tile_list_scatter(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                              const uint2* rect, const void* work, uint32_t* point_list, uint32_t capacity,
                              hipStream_t s, const uint32_t* keys, uint2* pair_list) {
    if (T == 0 || P == 0) return;
    const WorkArea w = carve_work(pl, (void*)work, T);
    const uint2* rect_walk = order ? w.rect_sorted : rect;   // (order == NULL: id order, the rects as preprocess_fwd left them)
    const bool pairs = order == nullptr;
    const size_t lds = (3 * (size_t)pl.sband_tiles + (pairs ? 5 : 4) * (GGR_BIN_CHUNK / GGR_SCATTER_PARTS) + 64) * 4;
    const uint32_t bands8 = (pl.nsbands + 7u) / 8u;
    if (pairs)
        hipLaunchKernelGGL(bin_scatter_kernel<true>, dim3(pl.nchunks * bands8 * 8u), dim3(64), lds, s, (uint32_t)P, order,
                           rect_walk, (uint32_t)T, pl.sband_tiles, (uint32_t)grid_x, w.table, w.wsum, point_list, capacity,
                           pl.nsbands, keys, pair_list);
    else
        hipLaunchKernelGGL(bin_scatter_kernel<false>, dim3(pl.nchunks * bands8 * 8u), dim3(64), lds, s, (uint32_t)P, order,
                           rect_walk, (uint32_t)T, pl.sband_tiles, (uint32_t)grid_x, w.table, w.wsum, point_list, capacity,
                           pl.nsbands, (const uint32_t*)nullptr, (uint2*)nullptr);
}

}  // namespace ggr
