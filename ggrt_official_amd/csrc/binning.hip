// binning.hip — the depth pre-sort of the tile binning for gfx950 (stable u32-key radix sort, three passes).
//
// Part of what replaces the scan / duplicateWithKeys / 64-bit radix sort / identifyTileRanges stages of
// the rasterizer behind reference cuda_splatting.py:114-125.
//
// MI355X-first reformulation (identical resulting lists, far less traffic):
//   the reference sorts N = Σ tiles_touched pairs by the 64-bit key (tile << 32 | depth_bits) — ≥6 radix
//   passes over N·12 B.  Here only the P Gaussians (P ≪ N) are sorted, by depth bits, with the stable
//   sort below (ties keep ascending id); tile_lists.hip then builds the per-tile lists from that order with
//   a stable counting sort by tile that never materialises the N pairs.
//
// Keys arrive as (depth bits − bits of 0.2f) (ggr_common.h GGR_KEY_BASE): the order is the same, the constant
// high part of the float bits is gone, and so a real scene's keys have ≈ 27 significant bits — THREE passes of
// ⌈bits/3⌉-bit digits instead of four 8-bit ones.  The digit width is found on the device (no host sync): every
// preprocess block leaves the maximum of its keys, the histogram kernel reduces them and publishes the width.
//
// The sort is hand-written for wave64: one upfront histogram of the three digits, then per digit ONE kernel
// ("onesweep") that ranks stably with ballot-based digit matching + per-wave digit counters in LDS, scans the
// digit totals itself (no separate scan launch) and obtains its tile's global offsets by decoupled look-back.
#include "ggr_common.h"
#include <algorithm>

namespace ggr {

// tile t owns keys [t*4096, (t+1)*4096); wave w of the workgroup owns a contiguous 512-key slice,
// round r of the wave covers 64 consecutive keys → order inside the tile is (wave, round, lane).
//
// Look-back protocol (guide §6 G16, recipe R2 — "the data IS the flag"): a status word is
// (flag << 30) | count with flag 1 = tile aggregate, 2 = inclusive prefix; it is written with ONE
// relaxed agent-scope atomic store (write-through, leaves the XCD's L2) and polled with relaxed
// agent-scope atomic loads (bypass the reader's L1), so no fence is needed and no ordering between
// different words is assumed.  Tiles take their index from an atomic ticket, so a tile only ever
// waits for tiles whose workgroups are already running: no dispatch-order assumption.
// While all tiles of a sort are resident at once (≤ 384 tiles: they run in lockstep and every tile starts far from the
// inclusive frontier) the look-back is a three-level tree of span-8 / span-64 aggregates — three dependent round trips
// for every tile (see the kernel); beyond that a tile walks back one predecessor per trip until it meets an inclusive
// prefix (measured, tools/sort_bench.hip: 4 M keys 0.313 ms with 1 per trip / 0.335 with 8 — in the pipelined regime
// the extra polls only add traffic).

#define GGR_FLAG_AGG 1u
#define GGR_FLAG_INCL 2u
#define GGR_COUNT_MASK 0x3FFFFFFFu
#define GGR_SPIN_LIMIT (1u << 22)
#ifndef GGR_LOOKBACK
#define GGR_LOOKBACK 8
#endif
// (GGR_FAULT_SPIN / _RANGE / _BUCKET: ggr_common.h)

#ifdef GGR_SORT_PROBE  // dev build (tools/sort_bench.hip): per-tile phase timestamps, 100 MHz constant clock
__device__ unsigned long long ggr_probe[3][8][2048];
#define PROBE(k) do { if (threadIdx.x == 0 && tile < 2048) ggr_probe[pass][k][tile] = wall_clock64(); } while (0)
#else
#define PROBE(k) do { } while (0)
#endif

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

// one maximum per 256 keys (what preprocess_fwd leaves behind; only tools/sort_bench.hip launches this)
__global__ void __launch_bounds__(GGR_PRE_THREADS)
radix_block_max_kernel(const uint32_t* __restrict__ keys, size_t n, uint32_t* __restrict__ block_max) {
    __shared__ uint32_t wm[GGR_PRE_THREADS / 64];
    const size_t i = (size_t)blockIdx.x * GGR_PRE_THREADS + threadIdx.x;
    const uint32_t m = wave_max_u32(i < n ? keys[i] : 0u);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) block_max[blockIdx.x] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}

// digit totals of all three passes in one read of the keys.  Every block first reduces the producer blocks' key
// maxima (n/256 words, one round trip, in flight together with its keys) to the digit width.
#ifndef GGR_HIST_THREADS
#define GGR_HIST_THREADS 1024
#endif
#ifndef GGR_HIST_ITEMS
#define GGR_HIST_ITEMS 8
#endif
__global__ void __launch_bounds__(GGR_HIST_THREADS)
radix_global_hist_kernel(const uint32_t* __restrict__ keys, size_t n /*keys per segment*/, uint32_t blocks_per_seg,
                         uint32_t* __restrict__ hist, const uint32_t* __restrict__ block_max, uint32_t nmax) {
    __shared__ uint32_t h[GGR_SORT_PASSES][GGR_SORT_MAX_BINS];
    __shared__ uint32_t wm[GGR_HIST_THREADS / 64];
    GGR_CRITICAL_PRIO();
    const int tid = threadIdx.x;
    // this block's keys: all loads issued before anything waits (the kernel is latency-bound)
    uint32_t ks[GGR_HIST_ITEMS];
    const uint32_t seg = blockIdx.x / blocks_per_seg, bseg = blockIdx.x - seg * blocks_per_seg;
    keys += (size_t)seg * n;
    const size_t base = (size_t)bseg * (GGR_HIST_THREADS * GGR_HIST_ITEMS);
#pragma unroll
    for (int u = 0; u < GGR_HIST_ITEMS; u++) {
        const size_t idx = base + (size_t)u * GGR_HIST_THREADS + tid;
        ks[u] = idx < n ? keys[idx] : 0xFFFFFFFFu;   // (0xFFFFFFFF marks "no key": real keys are < 2^31)
    }
    uint32_t m = 0;
    for (uint32_t i0 = 0; i0 < nmax; i0 += 4 * GGR_HIST_THREADS) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = block_max[min(i0 + u * GGR_HIST_THREADS + tid, nmax - 1)];
        m = max(max(m, v[0]), max(v[1], max(v[2], v[3])));
    }
    for (int x = tid; x < GGR_SORT_PASSES * GGR_SORT_MAX_BINS; x += GGR_HIST_THREADS) (&h[0][0])[x] = 0;
    m = wave_max_u32(m);
    if ((tid & 63) == 0) wm[tid >> 6] = m;
    __syncthreads();
    m = wave_max_u32(wm[tid & (GGR_HIST_THREADS / 64 - 1)]);  // (the first lanes hold the wave maxima, the others repeat them)
    const uint32_t bits = m ? 32u - (uint32_t)__builtin_clz(m) : 1u;
    uint32_t w = (bits + 2u) / 3u;
    if (w > GGR_SORT_MAX_BITS) {
        w = GGR_SORT_MAX_BITS;
        if (tid == 0 && blockIdx.x == 0) atomicOr(&hist[GGR_HIST_FAULT], GGR_FAULT_RANGE);
    }
    if (tid == 0 && blockIdx.x == 0) hist[GGR_HIST_PARAMS] = w;   // (read by the pass kernels: later launches)
    const uint32_t mask = (1u << w) - 1u;
#pragma unroll
    for (int u = 0; u < GGR_HIST_ITEMS; u++) {
        if (ks[u] != 0xFFFFFFFFu) {
            atomicAdd(&h[0][ks[u] & mask], 1u);
            atomicAdd(&h[1][(ks[u] >> w) & mask], 1u);
            atomicAdd(&h[2][(ks[u] >> (2u * w)) & mask], 1u);
        }
    }
    __syncthreads();
    const uint32_t bins = 1u << w;
    for (uint32_t x = tid; x < GGR_SORT_PASSES * bins; x += GGR_HIST_THREADS) {
        const uint32_t p = x >> w, d = x & mask;
        const uint32_t c = h[p][d];
        if (c) atomicAdd(&hist[GGR_HIST_TOTALS + (seg * GGR_SORT_PASSES + p) * GGR_SORT_MAX_BINS + d], c);
    }
}

// ---- the BUCKET form (round 6) -------------------------------------------------------------------------------------------------
// Three dependent onesweep passes are three look-back chains and three launch floors for 8 MB of keys (78-105 µs per million keys,
// 0.06 of the HBM roofline: NOTES r5).  The bucket form keeps ONE of them: a stable partition of the keys into <= 1024 depth
// buckets per segment, then every bucket sorted by (key, id) in LDS by one workgroup — the per-tile sort's routine
// (tile_sort.h), whose fast route ranks an entry among the few members of its sub-bucket.  The buckets are made EQUALLY FULL,
// whatever the depth distribution: this kernel takes a 4096-bin histogram of the frame's own key range [kmin, kmax] (from the
// preprocess blocks' maxima and minima; the culled Gaussians — key 0 — are counted apart), and every block of the partition
// pass turns it into splitters: consecutive fine bins are merged into a bucket until it holds `target` keys.  A bucket that
// still exceeds what a workgroup sorts (GGR_TSORT_CAP_LARGE keys inside ONE fine bin = 1/4096 of the frame's depth range) is
// copied out as it is when all its keys are equal (a plane of constant depth: the stable partition left it in id order) and
// raises GGR_FAULT_BUCKET otherwise: the caller sorts again with the three-pass form.
__global__ void __launch_bounds__(GGR_HIST_THREADS)
msd_hist_kernel(const uint32_t* __restrict__ keys, size_t n /*keys per segment*/, uint32_t blocks_per_seg,
                uint32_t* __restrict__ hist, const uint32_t* __restrict__ block_max, const uint32_t* __restrict__ block_min,
                uint32_t nmax, uint32_t* __restrict__ fine /*[segments][GGR_MSD_FINE_WORDS], zeroed*/) {
    __shared__ uint32_t h[GGR_MSD_FINE + 1];
    __shared__ uint32_t wm[GGR_HIST_THREADS / 64], wn[GGR_HIST_THREADS / 64];
    GGR_CRITICAL_PRIO();
    const int tid = threadIdx.x;
    uint32_t ks[GGR_HIST_ITEMS];
    const uint32_t seg = blockIdx.x / blocks_per_seg, bseg = blockIdx.x - seg * blocks_per_seg;
    keys += (size_t)seg * n;
    const size_t base = (size_t)bseg * (GGR_HIST_THREADS * GGR_HIST_ITEMS);
#pragma unroll
    for (int u = 0; u < GGR_HIST_ITEMS; u++) {
        const size_t idx = base + (size_t)u * GGR_HIST_THREADS + tid;
        ks[u] = idx < n ? keys[idx] : 0xFFFFFFFFu;   // (0xFFFFFFFF marks "no key": real keys are < 2^31)
    }
    uint32_t m = 0, mn = 0xFFFFFFFFu;
    for (uint32_t i0 = 0; i0 < nmax; i0 += 4 * GGR_HIST_THREADS) {
        uint32_t v[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = min(i0 + u * GGR_HIST_THREADS + tid, nmax - 1);
            v[u] = block_max[i];
            w[u] = block_min[i];
        }
        m = max(max(m, v[0]), max(v[1], max(v[2], v[3])));
        mn = min(min(mn, w[0]), min(w[1], min(w[2], w[3])));
    }
    for (int x = tid; x < GGR_MSD_FINE + 1; x += GGR_HIST_THREADS) h[x] = 0;
    m = wave_max_u32(m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    if ((tid & 63) == 0) { wm[tid >> 6] = m; wn[tid >> 6] = mn; }
    __syncthreads();
    m = wave_max_u32(wm[tid & (GGR_HIST_THREADS / 64 - 1)]);
    mn = wn[tid & (GGR_HIST_THREADS / 64 - 1)];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    // (the blocks leave (smallest visible key) − 1, ~0 when they saw none)
    const uint32_t kmin = mn == 0xFFFFFFFFu ? 1u : mn + 1u;
    const uint32_t range = m >= kmin ? m - kmin : 0u;
    const uint32_t nb = range ? 32u - (uint32_t)__builtin_clz(range) : 0u;
    const uint32_t sh = nb > 12u ? nb - 12u : 0u;       // (range >> sh) < 4096
    if (tid == 0 && blockIdx.x == 0) {
        hist[GGR_HIST_PARAMS] = GGR_SORT_MAX_BITS;          // the partition pass ranks on a 10-bit "digit": the bucket
        hist[GGR_HIST_MSD_KMIN] = kmin;
        hist[GGR_HIST_MSD_SHIFT] = sh;
    }
#pragma unroll
    for (int u = 0; u < GGR_HIST_ITEMS; u++)
        if (ks[u] != 0xFFFFFFFFu) atomicAdd(&h[ks[u] == 0u ? (uint32_t)GGR_MSD_FINE : min((ks[u] - kmin) >> sh, (uint32_t)GGR_MSD_FINE - 1u)], 1u);
    __syncthreads();
    for (uint32_t x = tid; x < GGR_MSD_FINE + 1; x += GGR_HIST_THREADS) {
        const uint32_t c = h[x];
        if (c) atomicAdd(&fine[(size_t)seg * GGR_MSD_FINE_WORDS + x], c);
    }
}

// GATHER (last pass of the depth sort only): every pair also carries an 8-byte payload looked up by its value,
// gather_dst[final position] = gather_src[val] — the tile rect of the Gaussian, so that the tile-list kernels can
// stream the rects in depth order without a separate gather launch; the pass also clears `zero_area`.
// ITEMS keys per thread (8 … 16): a sort of a little more than 256 tiles of 4096 keys (GGRt's LLFF eval frame:
// 1 146 880 Gaussians = 281 tiles on 256 CUs) runs with larger tiles instead of doubling up on 25 CUs.
// MSD (the bucket form's partition pass; pass = 0, GATHER = false): the "digit" of a key is its BUCKET — 0 for a culled Gaussian
// (key 0), else 1 + (keys in the fine bins before its own) / target, from the fine histogram `msd_fine` that every block scans
// for itself (16 KB from L2: no launch of its own, and the look-up table is wanted in LDS anyway); the pairs leave as
// (id, key) records in `pair_out`, the form the bucket sort reads, and the tile with ticket 0 writes the buckets' ranges.
template <bool GATHER, int ITEMS, bool MSD = false>
__global__ void __launch_bounds__(GGR_SORT_THREADS)
radix_onesweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n /*keys per segment*/,
                      int pass, uint32_t ntiles /*per segment*/, uint32_t nseg, int tree_lookback, uint32_t* __restrict__ hist,
                      const uint2* __restrict__ gather_src,
                      uint2* __restrict__ gather_dst, uint32_t* __restrict__ zero_area, uint32_t zero_words,
                      const uint32_t* __restrict__ msd_fine = nullptr, uint2* __restrict__ msd_ranges = nullptr,
                      uint2* __restrict__ pair_out = nullptr, uint32_t msd_target = 1u /*2^32 / (keys per bucket)*/) {
    constexpr int NW = GGR_SORT_THREADS / 64;
    constexpr int DPT = GGR_SORT_MAX_BINS / GGR_SORT_THREADS;  // digits per thread at the widest digit
    __shared__ uint16_t wcount[NW][GGR_SORT_MAX_BINS];  // per-wave digit counters (a wave holds 512 keys), later per-wave prefixes
    __shared__ uint32_t dbase[GGR_SORT_MAX_BINS];       // global start of this tile's run of every digit
    __shared__ uint32_t texcl[GGR_SORT_MAX_BINS];       // start of that run inside the tile's locally sorted order
    __shared__ uint2 sorted[(GGR_SORT_THREADS * ITEMS)];             // the tile's (key, val) pairs — then its payloads — in output order
    __shared__ uint32_t wsum[NW];
    __shared__ uint32_t tile_sh;
    __shared__ uint16_t lut[MSD ? GGR_MSD_FINE : 2];   // MSD: fine bin → bucket
    GGR_CRITICAL_PRIO();
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // segments are dealt round robin to the workgroups, so that all of them progress together
    const uint32_t seg = blockIdx.x % nseg;
    const bool vals_in_null = vals_in == nullptr;
    if (tid == 0) tile_sh = atomicAdd(&hist[GGR_HIST_TICKETS + pass * GGR_SORT_MAX_SEGMENTS + seg], 1u);
    {
        const size_t so = (size_t)seg * n;
        keys_in += so; vals_out += so;
        if (!MSD) keys_out += so;   // (MSD: no keys_out — the pairs leave as records)
        if (!vals_in_null) vals_in += so;
        if (GATHER || (MSD && gather_dst)) gather_dst += so;
        if (MSD) pair_out += so;
    }
    const uint32_t w = MSD ? (uint32_t)GGR_SORT_MAX_BITS : hist[GGR_HIST_PARAMS];   // bits per digit (block-uniform)
    const uint32_t bins = 1u << w, mask = bins - 1u;
    const uint32_t shift = (uint32_t)pass * w;
    const uint32_t msd_kmin = MSD ? hist[GGR_HIST_MSD_KMIN] : 0u, msd_shift = MSD ? hist[GGR_HIST_MSD_SHIFT] : 0u;
    // the digit of a key (MSD: its bucket; a slot beyond the segment's end never uses it)
    auto digit_of = [&](uint32_t k) -> uint32_t {
        if (MSD) return (k == 0u || k == 0xFFFFFFFFu) ? 0u : (uint32_t)lut[min((k - msd_kmin) >> msd_shift, (uint32_t)GGR_MSD_FINE - 1u)];
        return (k >> shift) & mask;
    };
    // this pass's digit totals, requested now (they are final: the histogram kernel has ended)
    uint32_t tot[DPT];
    if (!MSD) {
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t d = tid + q * GGR_SORT_THREADS;
            tot[q] = d < bins ? hist[GGR_HIST_TOTALS + (seg * GGR_SORT_PASSES + pass) * GGR_SORT_MAX_BINS + d] : 0u;
        }
    }
    for (uint32_t x = tid; x < NW * GGR_SORT_MAX_BINS / 2; x += GGR_SORT_THREADS)
        reinterpret_cast<uint32_t*>(&wcount[0][0])[x] = 0u;
    if (MSD) {
        // splitters: thread t owns fine bins [FPT·t, FPT·(t+1)); bucket of a bin = 1 + (visible keys in the bins before it) / target
        constexpr int FPT = GGR_MSD_FINE / GGR_SORT_THREADS;
        const uint32_t* fseg = msd_fine + (size_t)seg * GGR_MSD_FINE_WORDS;
        uint32_t c[FPT], tsum = 0u;
#pragma unroll
        for (int j = 0; j < FPT; j++) { c[j] = fseg[FPT * tid + j]; tsum += c[j]; }
        const uint32_t culled = fseg[GGR_MSD_FINE];
        uint32_t incl = tsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        for (uint32_t x = tid; x < GGR_SORT_MAX_BINS; x += GGR_SORT_THREADS) texcl[x] = 0u;   // (bucket totals, until the look-back)
        __syncthreads();
        uint32_t e = incl - tsum;
#pragma unroll
        for (int ww = 0; ww < NW; ww++) e += ww < wave ? wsum[ww] : 0u;
#pragma unroll
        for (int j = 0; j < FPT; j++) {
            // (any non-decreasing function of e will do — every block computes the same one: a multiply-high, not a division)
            const uint32_t b = min(1u + __umulhi(e, msd_target), (uint32_t)GGR_SORT_MAX_BINS - 1u);
            lut[FPT * tid + j] = (uint16_t)b;
            if (c[j]) atomicAdd(&texcl[b], c[j]);
            e += c[j];
        }
        if (tid == 0) texcl[0] = culled;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DPT; q++) tot[q] = texcl[tid + q * GGR_SORT_THREADS];
    }
    __syncthreads();
    const uint32_t tile = tile_sh;
    PROBE(0);
    const bool tree = tree_lookback != 0;  // (uniform; decided by the host together with the size of the status area)
    const size_t level_words = (size_t)ntiles << GGR_SORT_MAX_BITS;
    uint32_t* status = hist + ggr_sort_status_base(nseg) + ((size_t)pass * nseg + seg) * level_words * (tree ? GGR_SORT_LEVELS : 1);

    const size_t base = (size_t)tile * (GGR_SORT_THREADS * ITEMS) + (size_t)wave * (64 * ITEMS);
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        // (vals_in == nullptr: the first pass of a sort whose values are the identity — the index in the whole array)
        val[r] = !valid ? 0u : vals_in_null ? (uint32_t)((size_t)seg * n + idx) : vals_in[idx];
    }
    uint2 pay[GATHER ? ITEMS : 1];
    if (GATHER) {
        for (uint32_t wz = blockIdx.x * GGR_SORT_THREADS + tid; wz < zero_words; wz += gridDim.x * GGR_SORT_THREADS)
            zero_area[wz] = 0u;
        // issued now, consumed after the ranking and the look-back: the random 8-B reads hide behind them
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const size_t idx = base + r * 64 + lane;
            pay[r] = idx < n ? gather_src[val[r]] : make_uint2(0u, 0u);
        }
    }
    // exclusive scan of the digit totals → global digit bases (was a separate one-block launch): thread d owns
    // digits d and d + 512; wave scan + the 8 wave sums, the second half continues the first
    {
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            if ((uint32_t)q * GGR_SORT_THREADS < bins) {  // (block-uniform)
                uint32_t incl = tot[q];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
                    if (lane >= off) incl += t;
                }
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                uint32_t before = run;
#pragma unroll
                for (int ww = 0; ww < NW; ww++) {
                    const uint32_t sw = wsum[ww];
                    if (ww < wave) before += sw;
                    run += sw;
                }
                const uint32_t d = tid + q * GGR_SORT_THREADS;
                if (d < bins) dbase[d] = before + incl - tot[q];
                __syncthreads();
            }
        }
    }
    if (MSD && tile == 0u) {   // the buckets' ranges in the whole array (what the bucket-sort launch walks)
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t d = tid + q * GGR_SORT_THREADS, st = (uint32_t)((size_t)seg * n) + dbase[d];
            msd_ranges[(size_t)seg * GGR_SORT_MAX_BINS + d] = (tot[q] && d != 0u) ? make_uint2(st, st + tot[q]) : make_uint2(0u, 0u);
        }
    }
    PROBE(1);
    // wave-private counters: plain LDS accesses, ordered by wavefront-scope fences (LDS executes a wave's
    // operations in order; a `volatile` pointer here compiles to flat_load/flat_store + s_waitcnt vmcnt(0))
    uint16_t* wc = wcount[wave];
    // (a) the match masks of all eight rounds first — ballots only, the rounds are independent of each other —
    // (b) then the chain through the wave's counters: two dependent LDS operations per round and nothing else
    // (with the ballots inside the chain a round cost ≈ 0.4 µs: tools/sort_bench.hip -DGGR_SORT_PROBE)
    uint32_t before[ITEMS], cnt[ITEMS];
    uint32_t dig[MSD ? ITEMS : 1];   // MSD: a key's bucket costs an LDS look-up — once per key, not once per use
    if (MSD) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) dig[MSD ? r : 0] = digit_of(key[r]);
    }
    auto digit_r = [&](int r) -> uint32_t { return MSD ? dig[MSD ? r : 0] : digit_of(key[r]); };
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        const uint32_t d = digit_r(r);
        uint64_t m = __ballot(idx < n);
#pragma unroll
        for (int b = 0; b < GGR_SORT_MAX_BITS; b++) {
            if ((uint32_t)b < w) {  // (uniform)
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
        }
        // m = valid lanes of this round holding the same digit
        before[r] = (uint32_t)__popcll(m & lt_mask);
        cnt[r] = (uint32_t)__popcll(m);
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = digit_r(r);
        uint32_t prev = 0;
        if (valid) prev = wc[d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && before[r] == 0) wc[d] = (uint16_t)(prev + cnt[r]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rank[r] = prev + before[r];
    }
    PROBE(2);
    __syncthreads();
    PROBE(3);
    // thread d owns digits d (and d + 512): publish the tile aggregate, look back, publish the inclusive prefix;
    // then dbase[d] = global start of this tile's run of digit d, wcount[w][d] = wave w's offset inside that run
    uint32_t dtot[DPT];
#pragma unroll
    for (int q = 0; q < DPT; q++) {
        const uint32_t d = tid + q * GGR_SORT_THREADS;
        dtot[q] = 0;
        if (d < bins) {
            uint32_t cw[NW], total = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { cw[ww] = wcount[ww][d]; total += cw[ww]; }
            uint32_t* mine = status + (size_t)tile * bins + d;
            uint32_t excl = 0;
            if (tree) {
                // ---- tree look-back (all tiles of the sort resident at once: they reach this point within ≈ 3 µs of each
                // other, and walking back 8 predecessors per trip costs the last tiles ≈ 6 round trips —
                // profiles/r02_sort_phase_probe.txt).  Three levels of status words, each (1 ≪ 30) | count:
                //   L0[t] = tile t's count;  L1[t] = Σ L0 over tiles max(0, t−7) … t;  L2[t] = Σ L0 over max(0, t−63) … t.
                // A tile sums L0 of its 7 predecessors, publishes L1, sums L1 of t−8, t−16, … t−56, publishes L2, sums L2 of
                // t−64, t−128, …: three dependent round trips whatever the tile.  It only ever waits for lower tickets.
                constexpr uint32_t FLAG = 1u << 30;
                uint32_t* const l0 = status + d;
                uint32_t* const l1 = l0 + level_words;
                uint32_t* const l2 = l1 + level_words;
                __hip_atomic_store(l0 + (size_t)tile * bins, FLAG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t spins = 0;
                bool fault = false;
                // sum of `cnt` (≤ 7) flagged words src[(t0 − step·j)·bins], j = 0 … cnt−1: all loads of a trip in flight together
                auto gather7 = [&](const uint32_t* src, int t0, int step, int cnt) -> uint32_t {
                    uint32_t got = 0, sum = 0;
                    const uint32_t want = (1u << cnt) - 1u;
                    while (got != want) {
                        uint32_t v[7];
#pragma unroll
                        for (int j = 0; j < 7; j++)
                            v[j] = (j < cnt && !((got >> j) & 1u))
                                       ? __hip_atomic_load(src + (size_t)(t0 - step * j) * bins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                        uint32_t fresh = 0;
#pragma unroll
                        for (int j = 0; j < 7; j++)
                            if (v[j] >> 30) { sum += v[j] & GGR_COUNT_MASK; fresh |= 1u << j; }
                        got |= fresh;
                        if (got != want && !fresh) {
                            if (++spins > GGR_SPIN_LIMIT) { fault = true; break; }  // never hang
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    return sum;
                };
                if (tile > 0) {
                    const int t = (int)tile;
                    // (after a timeout the words are still published — the sort's result is void, GGR_E_HIP, but nobody
                    //  else should wait for this tile — and no further level is awaited)
                    uint32_t s = total + gather7(l0, t - 1, 1, min(7, t));                    // tiles max(0, t−7) … t
                    __hip_atomic_store(l1 + (size_t)tile * bins, FLAG | (s & GGR_COUNT_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t >= 8 && !fault) s += gather7(l1, t - 8, 8, min(7, t / 8));          // … max(0, t−63) … t
                    __hip_atomic_store(l2 + (size_t)tile * bins, FLAG | (s & GGR_COUNT_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t >= 64 && !fault) s += gather7(l2, t - 64, 64, min(7, t / 64));      // everything before
                    excl = s - total;
                    if (fault) atomicOr(&hist[GGR_HIST_FAULT], GGR_FAULT_SPIN);
                } else {
                    __hip_atomic_store(l1 + (size_t)tile * bins, FLAG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(l2 + (size_t)tile * bins, FLAG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
            __hip_atomic_store(mine, ((tile == 0 ? GGR_FLAG_INCL : GGR_FLAG_AGG) << 30) | total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            if (tile > 0) {
                int t = (int)tile - 1;
                uint32_t spins = 0;
                bool done = false;
                const int depth = 1;  // (pipelined regime: more polls per trip only add traffic — 4 M keys 0.313 ms with 1, 0.335 with 8)
                while (!done) {
                    uint32_t v[GGR_LOOKBACK];
#pragma unroll
                    for (int j = 0; j < GGR_LOOKBACK; j++)
                        v[j] = j < depth ? __hip_atomic_load(status + (size_t)max(t - j, 0) * bins + d, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    int used = 0;
#pragma unroll
                    for (int j = 0; j < GGR_LOOKBACK; j++) {
                        if (!done && used == j && j < depth && t - j >= 0) {
                            const uint32_t flag = v[j] >> 30;
                            if (flag != 0) {
                                excl += v[j] & GGR_COUNT_MASK;
                                used = j + 1;
                                if (flag == GGR_FLAG_INCL || t - j == 0) done = true;
                            }
                        }
                    }
                    t -= used;
                    if (!done && used == 0) {
                        if (++spins > GGR_SPIN_LIMIT) { atomicOr(&hist[GGR_HIST_FAULT], GGR_FAULT_SPIN); break; }  // never hang
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __hip_atomic_store(mine, (GGR_FLAG_INCL << 30) | ((excl + total) & GGR_COUNT_MASK), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            }
            // (only this thread touches digit d's column: no barrier between the reads above and these writes)
            dbase[d] += excl;
            dtot[q] = total;
            uint32_t pre = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { wcount[ww][d] = (uint16_t)pre; pre += cw[ww]; }
        }
    }
    PROBE(4);
    // exclusive scan of the tile's own digit counts → where each digit's run starts in the tile's sorted order
    {
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            if ((uint32_t)q * GGR_SORT_THREADS < bins) {  // (block-uniform)
                uint32_t incl = dtot[q];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
                    if (lane >= off) incl += t;
                }
                __syncthreads();  // (wsum: the previous round's readers are done)
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                uint32_t bef = run;
#pragma unroll
                for (int ww = 0; ww < NW; ww++) {
                    const uint32_t sw = wsum[ww];
                    if (ww < wave) bef += sw;
                    run += sw;
                }
                const uint32_t d = tid + q * GGR_SORT_THREADS;
                if (d < bins) texcl[d] = bef + incl - dtot[q];
            }
        }
    }
    __syncthreads();
    PROBE(5);
    // Local reorder: the tile's pairs go through LDS into (digit, wave, round, lane) order — the order of the output —
    // and thread i then stores elements i, i + 512, …: consecutive lanes write consecutive addresses inside a digit's
    // run instead of 64 unrelated 4-byte stores per instruction.
    uint32_t lpos[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        const uint32_t d = digit_r(r);
        lpos[r] = texcl[d] + wcount[wave][d] + rank[r];
        if (idx < n) sorted[lpos[r]] = make_uint2(key[r], val[r]);
    }
    __syncthreads();
    const uint32_t tile_n = (uint32_t)min((size_t)(GGR_SORT_THREADS * ITEMS), n - (size_t)tile * (GGR_SORT_THREADS * ITEMS));
    uint32_t gpos[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t j = tid + k * GGR_SORT_THREADS;
        gpos[k] = 0;
        if (j < tile_n) {
            const uint2 kv = sorted[j];
            const uint32_t d = digit_of(kv.x);
            gpos[k] = dbase[d] + (j - texcl[d]);
            if (MSD) {
                // (id, key): what the bucket sort reads.  The culled Gaussians (bucket 0: key 0, often a tenth of the frame and
                // far more than one workgroup should copy) are final as they stand — stable partition = id order — and leave here
                if (d != 0u) pair_out[gpos[k]] = make_uint2(kv.y, kv.x);
                else {
                    vals_out[gpos[k]] = kv.y;
                    if (gather_dst) gather_dst[gpos[k]] = gather_src[kv.y];
                }
            } else {
                keys_out[gpos[k]] = kv.x;
                vals_out[gpos[k]] = kv.y;
            }
        }
    }
    if (GATHER) {
        __syncthreads();  // every pair has been read: the same LDS now carries the payloads
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const size_t idx = base + r * 64 + lane;
            if (idx < n) sorted[lpos[r]] = pay[r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t j = tid + k * GGR_SORT_THREADS;
            if (j < tile_n) gather_dst[gpos[k]] = sorted[j];
        }
    }
    PROBE(6);
}

const uint32_t* radix_sort_fault_word(const uint32_t* hist) { return hist + GGR_HIST_FAULT; }

void radix_sort_pairs(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                      uint32_t* hist, size_t n, uint32_t segments, uint32_t** keys_out, uint32_t** vals_out,
                      hipStream_t s, bool hist_zeroed, uint32_t block_max_ready, bool identity_vals, const uint2* gather_src, uint2* gather_dst,
                      uint32_t* zero_area, uint32_t zero_words, bool buckets) {
    uint32_t *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
    if (n > 0 && buckets && identity_vals && block_max_ready && gather_src && gather_dst && radix_sort_buckets_ok(n / (segments ? segments : 1))) {
        // ---- the bucket form: fine histogram → ONE partition pass → every bucket sorted in LDS (+ payload gather) -------------
        const uint32_t S = segments ? segments : 1;
        const size_t nseg = n / S;
        int items = 8;
        for (int it = 8; it <= 16 && S * ggr_sort_blocks(nseg) > 256; it += 2)
            if (S * ((nseg + (size_t)it * GGR_SORT_THREADS - 1) / ((size_t)it * GGR_SORT_THREADS)) <= 256) { items = it; break; }
        const uint32_t ntiles = (uint32_t)((nseg + (size_t)items * GGR_SORT_THREADS - 1) / ((size_t)items * GGR_SORT_THREADS));
        const int tree = ggr_sort_blocks(nseg) * S <= GGR_SORT_TREE_MAX_TILES ? 1 : 0;
        uint32_t* block_max = hist + ggr_sort_block_max_at(n, S);
        uint32_t* block_min = hist + ggr_sort_block_min_at(n, S);
        uint32_t* fine = hist + ggr_sort_lsd_zero_words(n, S);
        uint2* ranges = reinterpret_cast<uint2*>(hist + ggr_sort_bucket_ranges_at(n, S));
        // the (id, key) records of the partition pass: keys_b and the (unused: identity values) vals_a behind it are one region
        uint2* pairs = reinterpret_cast<uint2*>(keys_b);
        if (!hist_zeroed) (void)hipMemsetAsync(hist, 0, ggr_sort_zero_words(n, S) * sizeof(uint32_t), s);
        const unsigned bps = (unsigned)((nseg + GGR_HIST_THREADS * GGR_HIST_ITEMS - 1) / (GGR_HIST_THREADS * GGR_HIST_ITEMS));
        hipLaunchKernelGGL(msd_hist_kernel, dim3(bps * S), dim3(GGR_HIST_THREADS), 0, s, keys_a, nseg, bps, hist, block_max,
                           block_min, block_max_ready, fine);
        // buckets of `target` keys (+ what the last fine bin brings): at most 1022 of them per segment beside the culled one
        const uint32_t target = std::max<uint32_t>(1024u, (uint32_t)(nseg / 1000) + 1u);
        const uint32_t target_inv = (uint32_t)((1ull << 32) / target);   // bucket of the e-th visible key = 1 + (e · inv) >> 32
#define GGR_PART(ITEMS_)                                                                                                   \
    hipLaunchKernelGGL((radix_onesweep_kernel<false, ITEMS_, true>), dim3(ntiles * S), dim3(GGR_SORT_THREADS), 0, s, keys_a, \
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, vals_b, nseg, 0, ntiles, S, tree, hist,                 \
                       gather_src, gather_dst, (uint32_t*)nullptr, 0u, fine, ranges, pairs, target_inv)
        if (items == 8) GGR_PART(8);
        else if (items == 10) GGR_PART(10);
        else if (items == 12) GGR_PART(12);
        else if (items == 14) GGR_PART(14);
        else GGR_PART(16);
#undef GGR_PART
        // the buckets, each by one workgroup: first the class that holds a regular bucket, then (mostly an empty launch) what
        // an overfull fine bin made longer; beyond GGR_TSORT_CAP_LARGE: copied if its keys are all equal, else the fault bit
        const uint32_t cls1 = target <= 1280u ? 2048u : target <= 2200u ? 3072u : 4096u;
        TileSortExtras ex{gather_src, gather_dst, zero_area, zero_words, hist + GGR_HIST_FAULT};
        launch_tile_depth_sort((size_t)S * GGR_SORT_MAX_BINS, ranges, vals_b, pairs, 0u, cls1, s, 0, nullptr, &ex);
        ex.zero_words = 0u;
        launch_bucket_sort_big(S, ranges, vals_b, pairs, cls1, s, ex);
        *keys_out = nullptr;
        *vals_out = vals_b;
        return;
    }
    if (n > 0) {
        const uint32_t S = segments ? segments : 1;
        const size_t nseg = n / S;  // (n is a multiple of S)
        // tile size: 4096 keys, or 5120 … 8192 if that brings the sort down to one tile per CU (≤ 2.1 M keys)
        int items = 8;
        for (int it = 8; it <= 16 && S * ggr_sort_blocks(nseg) > 256; it += 2)
            if (S * ((nseg + (size_t)it * GGR_SORT_THREADS - 1) / ((size_t)it * GGR_SORT_THREADS)) <= 256) { items = it; break; }
        const uint32_t ntiles = (uint32_t)((nseg + (size_t)items * GGR_SORT_THREADS - 1) / ((size_t)items * GGR_SORT_THREADS));
        // (the status area was sized — and cleared — for tiles of 4096 keys: ggr_sort_zero_words; same rule here)
        const int tree = ggr_sort_blocks(nseg) * S <= GGR_SORT_TREE_MAX_TILES ? 1 : 0;
        const uint32_t nmax = block_max_ready ? block_max_ready : (uint32_t)((n + GGR_PRE_THREADS - 1) / GGR_PRE_THREADS);
        uint32_t* block_max = hist + ggr_sort_block_max_at(n, S);
        if (!hist_zeroed)  // (ggr_forward: preprocess_fwd clears the area — one launch less)
            (void)hipMemsetAsync(hist, 0, ggr_sort_zero_words(n, S) * sizeof(uint32_t), s);
        if (!block_max_ready)  // (ggr_forward: preprocess_fwd leaves them)
            hipLaunchKernelGGL(radix_block_max_kernel, dim3(nmax), dim3(GGR_PRE_THREADS), 0, s, kin, n, block_max);
        const unsigned bps =
            (unsigned)((nseg + GGR_HIST_THREADS * GGR_HIST_ITEMS - 1) / (GGR_HIST_THREADS * GGR_HIST_ITEMS));
        hipLaunchKernelGGL(radix_global_hist_kernel, dim3(bps * S), dim3(GGR_HIST_THREADS), 0, s, kin, nseg, bps, hist,
                           block_max, nmax);
#define GGR_PASS(GATHER_, ITEMS_, SRC_, DST_, ZA_, ZW_)                                                                    \
    hipLaunchKernelGGL((radix_onesweep_kernel<GATHER_, ITEMS_>), dim3(ntiles * S), dim3(GGR_SORT_THREADS), 0, s, kin,       \
                       (p == 0 && identity_vals) ? (const uint32_t*)nullptr : (const uint32_t*)vin,                               \
                       kout, vout, nseg, p, ntiles, S, tree, hist, SRC_, DST_, ZA_, ZW_)
        for (int p = 0; p < GGR_SORT_PASSES; p++) {
            if (p == GGR_SORT_PASSES - 1 && gather_src) {
                if (items == 8) GGR_PASS(true, 8, gather_src, gather_dst, zero_area, zero_words);
                else if (items == 10) GGR_PASS(true, 10, gather_src, gather_dst, zero_area, zero_words);
                else if (items == 12) GGR_PASS(true, 12, gather_src, gather_dst, zero_area, zero_words);
                else if (items == 14) GGR_PASS(true, 14, gather_src, gather_dst, zero_area, zero_words);
                else GGR_PASS(true, 16, gather_src, gather_dst, zero_area, zero_words);
            } else {
                if (items == 8) GGR_PASS(false, 8, nullptr, nullptr, nullptr, 0u);
                else if (items == 10) GGR_PASS(false, 10, nullptr, nullptr, nullptr, 0u);
                else if (items == 12) GGR_PASS(false, 12, nullptr, nullptr, nullptr, 0u);
                else if (items == 14) GGR_PASS(false, 14, nullptr, nullptr, nullptr, 0u);
                else GGR_PASS(false, 16, nullptr, nullptr, nullptr, 0u);
            }
            uint32_t* t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
#undef GGR_PASS
    }
    *keys_out = kin;
    *vals_out = vin;
}

}  // namespace ggr
