// tile_sort.hip — stable per-tile sort of the tile lists by depth key (gfx950).
//
// Second form of the binning (the first: a global depth sort of the P Gaussians in front of the tile-list build,
// binning.hip).  Replaces duplicateWithKeys + the 64-bit radix sort + identifyTileRanges of the rasterizer behind
// reference cuda_splatting.py:114-125 with the same lists, bit for bit:
//
//   * the tile-list builder (tile_lists.hip) walks the Gaussians in ID order instead of depth order — its chunks are runs of
//     ids, the count needs no sorted input, and the in-order scatter leaves every tile's list in ascending id;
//   * this kernel then sorts each tile's list by the Gaussians' depth keys with a STABLE LSD radix sort in LDS — equal depths
//     keep ascending id, which is the order of the reference's stable sort on (tile << 32 | depth bits) with values emitted
//     in id order.
//
// What it buys: the global sort is four launches whose duration is a chain of cross-XCD look-backs (78-103 µs for 1 M keys
// that are 52 MB, NOTES r5 floor table), all of it on the forward's critical path; the per-tile sorts are independent
// workgroups with no global dependency.  What it costs: N list entries are sorted instead of P ≪ N Gaussians — in LDS.
//
// One 256-thread workgroup per tile; the entries live in REGISTERS (Q rounds of 64 positions per wave: position =
// (wave·q + round)·64 + lane), LDS is the exchange between passes.  Only the key bits that differ inside the tile count
// (OR ^ AND over its keys: a frame-wide log-uniform depth spread leaves 25-26).  The kernel is bound by LDS operations —
// their number, their bank conflicts (random digits: half the LDS cycles of the first versions) and their latency — so:
//
//   * ONE counting pass on the TOP digit (≤ 9 bits), stable, puts every entry into its bucket; inside a bucket — two to
//     four entries at these sizes — every entry counts the members that precede it in (key, position) order: a handful of
//     broadcast reads instead of two more passes.  A tile whose largest bucket exceeds GGR_TSORT_RANK_MAX (depths clustered
//     in few buckets) takes the LSD route instead: passes of evenly split digits from the lowest bit up.
//   * a pass is written as few long in-order instruction streams instead of one dependent round trip per step:
//       counts   per-wave digit counters, one order-free ds_add per entry, all rounds back to back;
//       scan     over (digit, wave): base[wave][digit] = entries with a smaller digit + the digit's entries in earlier waves;
//       match    lanes of a round that hold the same digit find each other through a per-wave lane-mask word per
//                (digit mod 64) — atomic OR, read, store 0, round after round WITHOUT waiting: one wave's LDS operations
//                execute in order (the idiom of bin_scatter; no reliance on how the LDS orders conflicting atomics) — and
//                three ballots separate the digits that share a word;
//       rank     the lowest lane of each digit group takes the group's positions with ONE ds_add_rtn on the base (distinct
//                addresses within an instruction; rounds in program order ⇒ stable), the others get it by ds_bpermute;
//       move     entries to their positions in the exchange buffer, barrier, every thread reads its positions back.
//     The last step writes the ids straight into the tile's list.
#include "ggr_common.h"
#include <algorithm>

namespace ggr {

#ifndef GGR_TSORT_BITS
#define GGR_TSORT_BITS 9   // widest digit
#endif
#define GGR_TSORT_BINS (1 << GGR_TSORT_BITS)
#define GGR_TSORT_DPT ((GGR_TSORT_BINS + 255) / 256)   // digits per thread in the scan
#ifndef GGR_TSORT_RANK_MAX
#define GGR_TSORT_RANK_MAX 24   // largest bucket the one-pass route ranks by counting
#endif

// registers: the small class is held to 80 per lane (6 workgroups per CU: the kernel waits on LDS and on two global round
// trips per tile, and what hides them is resident workgroups); the large class keeps its 32 rounds in registers as it can
#ifndef GGR_TSORT_WAVES_PER_EU
#define GGR_TSORT_WAVES_PER_EU 8
#endif
#define GGR_TSORT_WAVES(Q_) __attribute__((amdgpu_waves_per_eu((Q_) <= 8 ? GGR_TSORT_WAVES_PER_EU : (Q_) <= 12 ? 4 : (Q_) <= 16 ? 3 : 1, 8)))

// LDS words of a workgroup whose exchange buffer holds `cap` entries (see the kernel's layout)
__host__ __device__ static inline uint32_t tsort_lds_words(uint32_t cap) {
    const uint32_t r1 = 2u * cap + 2u * GGR_TSORT_BINS, r2 = cap + 4u * GGR_TSORT_BINS + 512u;
    return ((r1 > r2 ? r1 : r2) + 16u + 3u) & ~3u;
}

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t ts_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t ts_wave_scan_add(uint32_t v) {   // inclusive (tile_lists.hip wave_scan_add)
    v += ts_dpp<0x111, 0xf>(v);
    v += ts_dpp<0x112, 0xf>(v);
    v += ts_dpp<0x114, 0xf>(v);
    v += ts_dpp<0x118, 0xf>(v);
    v += ts_dpp<0x142, 0xa>(v);
    v += ts_dpp<0x143, 0xc>(v);
    return v;
}
// compiler-level ordering of one wave's LDS operations (the hardware executes them in order; no wait is generated)
__device__ __forceinline__ void ts_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace

// Q: rounds per wave the registers hold — lists of up to 256·Q entries; cap ≤ 256·Q sizes the exchange buffer
template <int Q>
__global__ void __launch_bounds__(256) GGR_TSORT_WAVES(Q)
tile_depth_sort_kernel(uint32_t T, const uint2* __restrict__ ranges, uint32_t* __restrict__ point_list,
                       const uint2* __restrict__ pair_list, uint32_t min_len, uint32_t cap, int copy_longer) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // LDS (words), the two routes below laid over each other:
    //   route 1  [ex: cap × (id, key) = 2·cap] [fill: BINS] [starts: BINS]
    //   route 2  [exw: cap]                    [cnt: 4 × BINS] [same: 4 × 64 × u64 = 512]
    //   [red: 16] behind the longer of the two
    uint2* ex = reinterpret_cast<uint2*>(lds);
    uint32_t* fill = lds + 2 * cap;                  // [BINS]: counts, then the buckets' fill pointers (= their ends at last)
    uint32_t* starts = fill + GGR_TSORT_BINS;        // [BINS]: the buckets' first positions
    uint32_t* exw = lds;                             // (route 2 moves ids and keys one after the other through its words)
    uint32_t* cnt = lds + cap;
    unsigned long long* same = reinterpret_cast<unsigned long long*>(cnt + 4 * GGR_TSORT_BINS);
    uint32_t* red = lds + tsort_lds_words(cap) - 16;

    const uint32_t tile = blockIdx.x;
    if (tile >= T) return;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    if (n <= min_len) return;   // (uniform: before any barrier; another launch's class)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* list = point_list + range.x;
    const uint2* pairs = pair_list + range.x;
    if (n > cap) {
        // longer than anything this launch sorts.  The last launch of a forward leaves such a list UNSORTED BUT VALID (ids in
        // id order): the host learns the frame's longest list behind the blend and then sorts it and blends again — until
        // then the blend must not find uninitialised ids
        if (copy_longer)
            for (uint32_t i = tid; i < n; i += 256) list[i] = pairs[i].x;
        return;
    }
    if (n < 2u) {   // (nothing to sort, but the list must hold the id)
        if (n == 1u && tid == 0) list[0] = pairs[0].x;
        return;
    }
    const uint32_t q = (n + 255u) >> 8;            // rounds per wave in use (≤ Q)
    const uint32_t p_lo = wave * q * 64u + lane;   // this thread's position in round r: p_lo + 64·r

    // Rounds are walked in groups of four: a group is skipped as a whole when the list does not reach it (uniform branch);
    // inside a group nothing is conditional but the lanes' validity, so that the group's LDS / global operations are issued
    // back to back and waited for once.
#define TS_GROUPS(...)                                                    \
    _Pragma("unroll") for (int g_ = 0; g_ < Q / 4; g_++)                  \
        if ((uint32_t)(4 * g_) < q) {                                     \
            _Pragma("unroll") for (int u_ = 0; u_ < 4; u_++) {            \
                const int r = 4 * g_ + u_;                                \
                __VA_ARGS__                                               \
            }                                                             \
        }

    // ---- load: the (id, key) entries as the id-order scatter wrote them (one coalesced sweep), the bits that differ ----
    uint32_t id[Q], ky[Q];
    bool valid[Q];
#pragma unroll
    for (int r = 0; r < Q; r++) { id[r] = 0u; ky[r] = 0u; valid[r] = (uint32_t)r < q && p_lo + 64u * r < n; }
    TS_GROUPS({ const uint2 e = pairs[min(p_lo + 64u * r, n - 1u)]; id[r] = e.x; ky[r] = e.y; })
    uint32_t k_or = 0u, k_and = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < Q; r++)
        if (valid[r]) { k_or |= ky[r]; k_and &= ky[r]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k_or |= (uint32_t)__shfl_xor((int)k_or, off);
        k_and &= (uint32_t)__shfl_xor((int)k_and, off);
    }
    for (uint32_t d = tid; d < GGR_TSORT_BINS; d += 256) fill[d] = 0u;
    if (lane == 0) { red[wave] = k_or; red[4 + wave] = k_and; }
    __syncthreads();
    const uint32_t diff = (red[0] | red[1] | red[2] | red[3]) ^ (red[4] & red[5] & red[6] & red[7]);
    if (diff == 0u) {   // every key equal: the entries are in id order already
        TS_GROUPS({ if (valid[r]) list[p_lo + 64u * r] = id[r]; })
        return;
    }
    const uint32_t nbits = 32u - (uint32_t)__builtin_clz(diff);


    // ---- route 1: the entries into the buckets of their top digit in ANY order, then ranked by (key, id) inside the bucket ----
    // The (key, id) order is total — ids are unique — so nothing here depends on the order in which the entries arrive or in
    // which the LDS serves conflicting atomics: one shared histogram (an order-free ds_add per entry), a scan over the digits,
    // one ds_add_rtn per entry on its bucket's fill pointer (whatever it returns is a free slot of the bucket), and every entry
    // counts the members of its bucket that precede it: one 64-bit compare per member, (key << 32 | id) as the exchange
    // buffer holds it.  No per-wave counters, no lane matching, no ballots.
    {
        const uint32_t top_bits = min(nbits, (uint32_t)GGR_TSORT_BITS), top_shift = nbits - top_bits;
        const uint32_t bins = 1u << top_bits, dmask = bins - 1u;
        TS_GROUPS({ if (valid[r]) atomicAdd(&fill[(ky[r] >> top_shift) & dmask], 1u); })
        __syncthreads();
        uint32_t c[GGR_TSORT_DPT], tot = 0u, big = 0u;
#pragma unroll
        for (int j = 0; j < GGR_TSORT_DPT; j++) {
            const uint32_t d = tid * GGR_TSORT_DPT + j;
            c[j] = d < bins ? fill[d] : 0u;
            tot += c[j];
            big = max(big, c[j]);
        }
        const uint32_t incl = ts_wave_scan_add(tot);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, off));
        if (lane == 63u) { red[8 + wave] = incl; red[12 + wave] = big; }
        __syncthreads();
        uint32_t run = incl - tot;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) run += w < wave ? red[8 + w] : 0u;
        const uint32_t longest = max(max(red[12], red[13]), max(red[14], red[15]));
        if (longest <= GGR_TSORT_RANK_MAX) {   // (uniform)
#pragma unroll
            for (int j = 0; j < GGR_TSORT_DPT; j++) {
                const uint32_t d = tid * GGR_TSORT_DPT + j;
                if (d < bins) { fill[d] = run; starts[d] = run; }
                run += c[j];
            }
            __syncthreads();
            uint32_t pos[Q];
#pragma unroll
            for (int r = 0; r < Q; r++) pos[r] = 0u;
            TS_GROUPS({ if (valid[r]) pos[r] = atomicAdd(&fill[(ky[r] >> top_shift) & dmask], 1u); })
            TS_GROUPS({ if (valid[r]) ex[pos[r]] = make_uint2(id[r], ky[r]); })
            __syncthreads();   // (every entry is in its bucket; fill[d] = end of bucket d)
            const unsigned long long* ex64 = reinterpret_cast<const unsigned long long*>(ex);
#pragma unroll
            for (int g_ = 0; g_ < Q / 4; g_++)
                if ((uint32_t)(4 * g_) < q) {
                    uint32_t bs[4], be[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t d = (ky[4 * g_ + u] >> top_shift) & dmask;
                        bs[u] = starts[d];
                        be[u] = fill[d];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int r = 4 * g_ + u;
                        const unsigned long long mine = ((unsigned long long)ky[r] << 32) | id[r];
                        uint32_t rank = 0u;
                        if (valid[r])
                            for (uint32_t j = bs[u]; j < be[u]; j++) rank += ex64[j] < mine ? 1u : 0u;
                        if (valid[r]) list[bs[u] + rank] = id[r];
                    }
                }
            return;
        }
    }
    // ---- route 2 (a bucket too large to rank by counting — depths clustered in few buckets): stable LSD passes over evenly
    // split digits.  It relies on the entries arriving in id order (the id-order scatter is stable): equal keys keep it.
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t p0[Q];
    uint32_t* my_cnt = cnt + wave * GGR_TSORT_BINS;
    unsigned long long* my_same = same + wave * 64;
    __syncthreads();       // (route 1's counters and this route's lie over each other: every thread is done with the former)
    my_same[lane] = 0ull;
    // One stable counting pass on the digit (key >> shift) & (2^bits − 1): leaves every entry's position in p0[]
    // (valid lanes) and `longest` = the largest bucket.  `cnt` must be zero on entry; on exit cnt[3·BINS + d] = end of bucket d
    // once every wave has ranked (barrier).
    uint32_t longest = 0u;
    (void)longest;
    auto counting_pass = [&](uint32_t shift, uint32_t bits) {
        const uint32_t bins = 1u << bits, dmask = bins - 1u;
#define TS_DIGIT(r_) ((ky[r_] >> shift) & dmask)   // (recomputed where it is needed: a register per round less)
        TS_GROUPS({ if (valid[r]) atomicAdd(&my_cnt[TS_DIGIT(r)], 1u); })
        __syncthreads();
        // scan over (digit, wave): thread t owns digits t·DPT …
        uint32_t c[GGR_TSORT_DPT][4], tot = 0u, big = 0u;
#pragma unroll
        for (int j = 0; j < GGR_TSORT_DPT; j++) {
            const uint32_t d = tid * GGR_TSORT_DPT + j;
            uint32_t td = 0u;
#pragma unroll
            for (int w = 0; w < 4; w++) { c[j][w] = d < bins ? cnt[w * GGR_TSORT_BINS + d] : 0u; td += c[j][w]; }
            tot += td;
            big = max(big, td);
        }
        const uint32_t incl = ts_wave_scan_add(tot);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, off));
        if (lane == 63u) { red[8 + wave] = incl; red[12 + wave] = big; }
        __syncthreads();
        uint32_t run = incl - tot;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) run += w < wave ? red[8 + w] : 0u;
        longest = max(max(red[12], red[13]), max(red[14], red[15]));
#pragma unroll
        for (int j = 0; j < GGR_TSORT_DPT; j++) {
            const uint32_t d = tid * GGR_TSORT_DPT + j;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if (d < bins) cnt[w * GGR_TSORT_BINS + d] = run;
                run += c[j][w];
            }
        }
        __syncthreads();
        // match: one in-order stream of LDS operations, nothing waits for a result
        // … four rounds at a time: a group's lane masks (two registers per round) are reduced to what the ranking needs — lanes
        // of the digit below this one, the digit's lowest lane and its count, packed into one register — before the next
        // group's are read
        uint32_t info[Q];   // before | leader << 8 | count << 16
#pragma unroll
        for (int g_ = 0; g_ < Q / 4; g_++)
            if ((uint32_t)(4 * g_) < q) {
                unsigned long long m[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = 4 * g_ + u;
                    const uint32_t h = TS_DIGIT(r) & 63u;
                    m[u] = 0ull;
                    if (valid[r]) atomicOr(&my_same[h], 1ull << lane);
                    ts_order();
                    if (valid[r]) m[u] = my_same[h];
                    ts_order();
                    if (valid[r]) my_same[h] = 0ull;   // (every lane of the word stores the same 0: no leader needed yet)
                    ts_order();
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = 4 * g_ + u;
                    // (digits that share a mask word differ in bits 6 …: a ballot per bit separates them.  Written on the
                    //  mask's halves with lane-wise 0 / ~0 words: five vector instructions per bit)
                    uint32_t mlo = (uint32_t)m[u], mhi = (uint32_t)(m[u] >> 32);
                    if (bits > 6) {
                        const uint32_t d = TS_DIGIT(r);
#pragma unroll
                        for (uint32_t k = 6; k < GGR_TSORT_BITS; k++) {
                            const uint32_t bit = (d >> k) & 1u;
                            const unsigned long long bal = __ballot(bit != 0u && valid[r]);
                            const uint32_t nb = bit - 1u;   // bit set: keep the ballot's lanes; clear: the others
                            mlo &= (uint32_t)bal ^ nb;
                            mhi &= (uint32_t)(bal >> 32) ^ nb;
                        }
                    }
                    // lanes of the digit below this one (v_mbcnt), the digit's lowest lane, its lane count
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                    const uint32_t lead = mlo ? (uint32_t)__builtin_ctz(mlo) : mhi ? 32u + (uint32_t)__builtin_ctz(mhi) : lane;
                    info[r] = below | (lead << 8) | (((uint32_t)__popc(mlo) + (uint32_t)__popc(mhi)) << 16);
                }
            }
        // rank: the group's lowest lane reserves its positions; rounds in program order
#pragma unroll
        for (int r = 0; r < Q; r++) p0[r] = 0u;
        TS_GROUPS({
            if (valid[r] && (info[r] & 0xFFu) == 0u) p0[r] = atomicAdd(&my_cnt[TS_DIGIT(r)], info[r] >> 16);
            ts_order();
        })
        TS_GROUPS({ p0[r] = (uint32_t)__shfl((int)p0[r], (int)((info[r] >> 8) & 0xFFu)) + (info[r] & 0xFFu); })
#undef TS_DIGIT
    };
    const uint32_t npass = (nbits + GGR_TSORT_BITS - 1) / GGR_TSORT_BITS;
    const uint32_t bits = (nbits + npass - 1) / npass;
    for (uint32_t pass = 0; pass < npass; pass++) {
        // the wave's counters back to zero (its own ranking is behind it: in-order LDS; the scan of the pass that wrote them
        // lies behind a barrier)
        for (uint32_t d = lane; d < GGR_TSORT_BINS; d += 64) my_cnt[d] = 0u;
        counting_pass(pass * bits, bits);
        if (pass + 1 == npass) {
            TS_GROUPS({ if (valid[r]) list[p0[r]] = id[r]; })
            break;
        }
        // the exchange buffer holds one word per entry: the ids first, then the keys
        TS_GROUPS({ if (valid[r]) exw[p0[r]] = id[r]; })
        __syncthreads();
        TS_GROUPS({ id[r] = exw[min(p_lo + 64u * r, n - 1u)]; })
        __syncthreads();
        TS_GROUPS({ if (valid[r]) exw[p0[r]] = ky[r]; })
        __syncthreads();
        TS_GROUPS({ ky[r] = exw[min(p_lo + 64u * r, n - 1u)]; })
        // (the next pass's exchange writes come behind two more barriers: every thread has read its entries by then)
    }
#undef TS_GROUPS
}

template <int Q>
static void launch_tsort_class(size_t T, const uint2* ranges, uint32_t* point_list, const uint2* keys, uint32_t min_len,
                               uint32_t cap, int copy_longer, hipStream_t s) {
    const size_t lds = (size_t)tsort_lds_words(cap) * 4;
    if (lds > 64 * 1024)   // (a workgroup may take the CU's whole 160 KB, but beyond 64 KB it has to be asked for)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_depth_sort_kernel<Q>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tile_depth_sort_kernel<Q>, dim3((unsigned)T), dim3(256), lds, s, (uint32_t)T, ranges, point_list, keys,
                       min_len, cap, copy_longer);
}

void launch_tile_depth_sort(size_t T, const uint2* ranges, uint32_t* point_list, const uint2* keys, uint32_t min_len,
                            uint32_t max_len, hipStream_t s, int copy_longer) {
    if (T == 0 || max_len == 0u || max_len <= min_len) return;   // (a list of ONE entry is still copied out of the pairs)
    const uint32_t cap = std::min<uint32_t>((max_len + 255u) & ~255u, GGR_TSORT_CAP_LARGE);   // the exchange buffer of the launch
    // the class = rounds per wave the registers hold: 8 (lists up to 2048: 6 workgroups per CU), 12 (3072), 16 (4096), 32 (8192)
    if (cap <= 2048) launch_tsort_class<8>(T, ranges, point_list, keys, min_len, cap, copy_longer, s);
    else if (cap <= 3072) launch_tsort_class<12>(T, ranges, point_list, keys, min_len, cap, copy_longer, s);
    else if (cap <= 4096) launch_tsort_class<16>(T, ranges, point_list, keys, min_len, cap, copy_longer, s);
    else launch_tsort_class<32>(T, ranges, point_list, keys, min_len, cap, copy_longer, s);
}

}  // namespace ggr
