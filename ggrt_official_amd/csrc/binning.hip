// binning.hip — the depth pre-sort of the tile binning for gfx950 (stable u32-key radix sort).
//
// Part of what replaces the scan / duplicateWithKeys / 64-bit radix sort / identifyTileRanges stages of
// the rasterizer behind reference cuda_splatting.py:114-125.
//
// MI355X-first reformulation (identical resulting lists, far less traffic):
//   the reference sorts N = Σ tiles_touched pairs by the 64-bit key (tile << 32 | depth_bits) — ≥6 radix
//   passes over N·12 B.  Here only the P Gaussians (P ≪ N) are sorted, by depth bits, with the stable
//   sort below (ties keep ascending id, 4 passes over P·8 B); tile_lists.hip then builds the per-tile
//   lists from that order with a stable counting sort by tile that never materialises the N pairs.
//
// The radix sort is hand-written for wave64: one upfront histogram of all digits, then per 8-bit digit
// ONE kernel ("onesweep") that ranks stably with ballot-based digit matching (8 ballots per 64 keys)
// + per-wave digit counters in LDS and obtains its tile's global offsets by decoupled look-back.
#include "ggr_common.h"

namespace ggr {

// ---------------------------------------------------------------------------------------------
// radix sort ("onesweep": one kernel per 8-bit digit, decoupled look-back)
// ---------------------------------------------------------------------------------------------
// tile t owns keys [t*4096, (t+1)*4096); wave w of the workgroup owns a contiguous 1024-key slice,
// round r of the wave covers 64 consecutive keys → order inside the tile is (wave, round, lane).
//
// Work area `hist` (u32 words):
//   [0, 1024)                       digit totals of every pass            (global_hist kernel)
//   [1024, 2048)                    exclusive digit bases of every pass   (global_scan kernel)
//   [2048, 2048+64)                 tile tickets, one per pass; [2048+8] = spin-timeout flag
//   [2112 + p*ntiles*256 ...)       look-back status words of pass p: status[tile][digit]
// Every word that is polled or atomically incremented is zeroed by ONE hipMemsetAsync per sort
// (MI355X guide §6 G16: re-initialise every call).
//
// Look-back protocol (guide §6 G16, recipe R2 — "the data IS the flag"): a status word is
// (flag << 30) | count with flag 1 = tile aggregate, 2 = inclusive prefix; it is written with ONE
// relaxed agent-scope atomic store (write-through, leaves the XCD's L2) and polled with relaxed
// agent-scope atomic loads (bypass the reader's L1), so no fence is needed and no ordering between
// different words is assumed.  Tiles take their index from an atomic ticket, so a tile only ever
// waits for tiles whose workgroups are already running: no dispatch-order assumption.

#define GGR_HIST_TOTALS 0
#define GGR_HIST_BASES 1024
#define GGR_HIST_TICKETS 2048
#define GGR_HIST_STATUS 2112
#define GGR_FLAG_AGG 1u
#define GGR_FLAG_INCL 2u
#define GGR_COUNT_MASK 0x3FFFFFFFu
#define GGR_SPIN_LIMIT (1u << 24)

// digit totals of all passes in one read of the keys
__global__ void __launch_bounds__(256)
radix_global_hist_kernel(const uint32_t* __restrict__ keys, size_t n, int npasses, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[4][GGR_RADIX];
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; p++) h[p][tid] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // four keys per thread per trip, all four loads issued before the first use (the kernel is latency-bound)
    constexpr int U = 4;
    for (size_t idx0 = (size_t)blockIdx.x * blockDim.x + tid; idx0 < ((n + 255) & ~(size_t)255); idx0 += stride * U) {
        uint32_t ks[U];
        bool vs[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t idx = idx0 + (size_t)u * stride;
            vs[u] = idx < n;
            ks[u] = vs[u] ? keys[idx] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool valid = vs[u];
            const uint32_t k = ks[u];
            for (int p = 0; p < npasses; p++) {
                const uint32_t d = (k >> (8 * p)) & 255u;
                // wave-aggregate the (very common) case of a digit shared by the whole wave
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                const uint64_t act = __ballot(valid);
                if (__ballot(valid && d == d0) == act) {
                    if (valid && (uint32_t)__builtin_ctzll(act) == (uint32_t)(tid & 63)) atomicAdd(&h[p][d0], (uint32_t)__popcll(act));
                } else if (valid) {
                    atomicAdd(&h[p][d], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < npasses; p++) {
        const uint32_t c = h[p][tid];
        if (c) atomicAdd(&hist[GGR_HIST_TOTALS + p * GGR_RADIX + tid], c);
    }
}

// exclusive scan of each pass's 256 totals → digit bases (one block, thread d = digit d)
__global__ void __launch_bounds__(256)
radix_global_scan_kernel(int npasses, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[256];
    const int tid = threadIdx.x;
    for (int p = 0; p < npasses; p++) {
        const uint32_t v = hist[GGR_HIST_TOTALS + p * GGR_RADIX + tid];
        sh[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t t = tid >= off ? sh[tid - off] : 0u;
            __syncthreads();
            sh[tid] += t;
            __syncthreads();
        }
        hist[GGR_HIST_BASES + p * GGR_RADIX + tid] = sh[tid] - v;
        __syncthreads();
    }
}

// GATHER (last pass of the depth sort only): every pair also carries an 8-byte payload looked up by its value,
// gather_dst[final position] = gather_src[val] — the tile rect of the Gaussian, so that the tile-list kernels can
// stream the rects in depth order without a separate gather launch; the pass also clears `zero_area`.
template <bool GATHER>
__global__ void __launch_bounds__(GGR_SORT_THREADS)
radix_onesweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int pass,
                      uint32_t ntiles, uint32_t* __restrict__ hist, const uint2* __restrict__ gather_src,
                      uint2* __restrict__ gather_dst, uint32_t* __restrict__ zero_area, uint32_t zero_words) {
    constexpr int NW = GGR_SORT_THREADS / 64;
    __shared__ uint32_t wcount[NW][GGR_RADIX];  // per-wave digit counters, later per-wave output bases
    __shared__ uint32_t tile_sh;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int shift = pass * GGR_RADIX_BITS;
    if (tid == 0) tile_sh = atomicAdd(&hist[GGR_HIST_TICKETS + pass], 1u);
    for (int x = tid; x < NW * GGR_RADIX; x += GGR_SORT_THREADS) (&wcount[0][0])[x] = 0;
    __syncthreads();
    const uint32_t tile = tile_sh;
    uint32_t* status = hist + GGR_HIST_STATUS + (size_t)pass * ntiles * GGR_RADIX;

    const size_t base = (size_t)tile * GGR_SORT_TILE + (size_t)wave * (64 * GGR_SORT_ITEMS);
    uint32_t key[GGR_SORT_ITEMS], val[GGR_SORT_ITEMS], rank[GGR_SORT_ITEMS];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < GGR_SORT_ITEMS; r++) {
        const size_t idx = base + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[r] = valid ? vals_in[idx] : 0u;
    }
    uint2 pay[GATHER ? GGR_SORT_ITEMS : 1];
    if (GATHER) {
        for (uint32_t w = blockIdx.x * GGR_SORT_THREADS + tid; w < zero_words; w += gridDim.x * GGR_SORT_THREADS)
            zero_area[w] = 0u;
        // issued now, consumed after the ranking and the look-back: the random 8-B reads hide behind them
#pragma unroll
        for (int r = 0; r < GGR_SORT_ITEMS; r++) {
            const size_t idx = base + r * 64 + lane;
            pay[r] = idx < n ? gather_src[val[r]] : make_uint2(0u, 0u);
        }
    }
    // wave-private counters: plain LDS accesses, ordered by wavefront-scope fences (LDS executes a wave's
    // operations in order; a `volatile` pointer here compiles to flat_load/flat_store + s_waitcnt vmcnt(0))
    uint32_t* wc = wcount[wave];
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
        // m = valid lanes of this round holding the same digit
        const uint32_t before = (uint32_t)__popcll(m & lt_mask);
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t prev = 0;
        if (valid) prev = wc[d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) wc[d] = prev + cnt;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rank[r] = prev + before;
    }
    __syncthreads();
    // thread d < 256 owns digit d: publish the tile aggregate, look back, publish the inclusive prefix
    uint32_t cw[NW], g = 0;
    if (tid < GGR_RADIX) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) { cw[w] = wcount[w][tid]; total += cw[w]; }
        uint32_t* mine = status + (size_t)tile * GGR_RADIX + tid;
        __hip_atomic_store(mine, ((tile == 0 ? GGR_FLAG_INCL : GGR_FLAG_AGG) << 30) | total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile > 0) {
            uint32_t t = tile - 1, spins = 0;
            for (;;) {
                const uint32_t v = __hip_atomic_load(status + (size_t)t * GGR_RADIX + tid, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = v >> 30;
                if (flag == 0) {
                    if (++spins > GGR_SPIN_LIMIT) { hist[GGR_HIST_TICKETS + 8] = 1u; break; }  // never hang
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                excl += v & GGR_COUNT_MASK;
                if (flag == GGR_FLAG_INCL || t == 0) break;
                t--;
            }
            __hip_atomic_store(mine, (GGR_FLAG_INCL << 30) | ((excl + total) & GGR_COUNT_MASK), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        g = hist[GGR_HIST_BASES + pass * GGR_RADIX + tid] + excl;
    }
    __syncthreads();
    if (tid < GGR_RADIX) {
        uint32_t run = g;
#pragma unroll
        for (int w = 0; w < NW; w++) { wcount[w][tid] = run; run += cw[w]; }
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
            if (GATHER) gather_dst[pos] = pay[r];
        }
    }
}

const uint32_t* radix_sort_fault_word(const uint32_t* hist) { return hist + GGR_HIST_TICKETS + 8; }

size_t radix_hist_words(size_t n) { return GGR_HIST_STATUS + 4 * ggr_sort_blocks(n ? n : 1) * GGR_RADIX; }

void radix_sort_pairs(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                      uint32_t* hist, size_t n, int nbits, uint32_t** keys_out, uint32_t** vals_out,
                      hipStream_t s, bool hist_zeroed, const uint2* gather_src, uint2* gather_dst,
                      uint32_t* zero_area, uint32_t zero_words) {
    uint32_t *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
    const int npasses = (nbits + GGR_RADIX_BITS - 1) / GGR_RADIX_BITS;
    if (n > 0 && npasses > 0) {
        const uint32_t ntiles = (uint32_t)ggr_sort_blocks(n);
        if (!hist_zeroed)  // (ggr_forward: preprocess_fwd clears the area — one launch less)
            (void)hipMemsetAsync(hist, 0, (GGR_HIST_STATUS + (size_t)npasses * ntiles * GGR_RADIX) * sizeof(uint32_t), s);
        // one block per CU: the kernel ends with 256·npasses global atomics per block, and at 2048
        // blocks those ≈2 M contended atomics cost more (≈45 µs) than reading the keys
        // (64 blocks was tried for n ≈ 1 M: slower, 0.121 → 0.151 ms — each thread then walks 61 keys serially)
        const unsigned hist_blocks = (unsigned)min((size_t)256, (n + 255) / 256);
        hipLaunchKernelGGL(radix_global_hist_kernel, dim3(hist_blocks), dim3(256), 0, s, kin, n, npasses, hist);
        hipLaunchKernelGGL(radix_global_scan_kernel, dim3(1), dim3(256), 0, s, npasses, hist);
        for (int p = 0; p < npasses; p++) {
            if (p == npasses - 1 && gather_src)
                hipLaunchKernelGGL(radix_onesweep_kernel<true>, dim3(ntiles), dim3(GGR_SORT_THREADS), 0, s, kin, vin,
                                   kout, vout, n, p, ntiles, hist, gather_src, gather_dst, zero_area, zero_words);
            else
                hipLaunchKernelGGL(radix_onesweep_kernel<false>, dim3(ntiles), dim3(GGR_SORT_THREADS), 0, s, kin, vin,
                                   kout, vout, n, p, ntiles, hist, nullptr, nullptr, nullptr, 0u);
            uint32_t* t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
    }
    *keys_out = kin;
    *vals_out = vin;
}

}  // namespace ggr
