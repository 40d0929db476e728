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
//     The last step writes the ids straight into the list (in place: the list was read completely before).
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
#define GGR_TSORT_WAVES_PER_EU 6
#endif
#define GGR_TSORT_WAVES(Q_) __attribute__((amdgpu_waves_per_eu((Q_) <= 8 ? GGR_TSORT_WAVES_PER_EU : 1, 8)))

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
                       const uint32_t* __restrict__ keys, uint32_t min_len, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // [exchange: cap words] [cnt: 4 × BINS] [same: 4 × 64 × u64] [red: 16]
    uint32_t* ex = lds;
    uint32_t* cnt = ex + cap;
    unsigned long long* same = reinterpret_cast<unsigned long long*>(cnt + 4 * GGR_TSORT_BINS);
    uint32_t* red = reinterpret_cast<uint32_t*>(same + 4 * 64);

    const uint32_t tile = blockIdx.x;
    if (tile >= T) return;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    if (n < 2u || n <= min_len || n > cap) return;   // (uniform: before any barrier)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* list = point_list + range.x;
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

    // ---- load: ids (coalesced), keys (gathered: 4 MB per million Gaussians, L2-resident), the bits that differ ----
    uint32_t id[Q], ky[Q];
    bool valid[Q];
#pragma unroll
    for (int r = 0; r < Q; r++) { id[r] = 0u; ky[r] = 0u; valid[r] = (uint32_t)r < q && p_lo + 64u * r < n; }
    TS_GROUPS({ id[r] = list[min(p_lo + 64u * r, n - 1u)]; })
    TS_GROUPS({ ky[r] = keys[id[r]]; })
    uint32_t k_or = 0u, k_and = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < Q; r++)
        if (valid[r]) { k_or |= ky[r]; k_and &= ky[r]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k_or |= (uint32_t)__shfl_xor((int)k_or, off);
        k_and &= (uint32_t)__shfl_xor((int)k_and, off);
    }
    uint32_t* my_cnt = cnt + wave * GGR_TSORT_BINS;
    unsigned long long* my_same = same + wave * 64;
    for (uint32_t d = lane; d < GGR_TSORT_BINS; d += 64) my_cnt[d] = 0u;
    my_same[lane] = 0ull;
    if (lane == 0) { red[wave] = k_or; red[4 + wave] = k_and; }
    __syncthreads();
    const uint32_t diff = (red[0] | red[1] | red[2] | red[3]) ^ (red[4] & red[5] & red[6] & red[7]);
    if (diff == 0u) return;   // every key equal: the list is in id order already
    const uint32_t nbits = 32u - (uint32_t)__builtin_clz(diff);
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    // One stable counting pass on the digit (key >> shift) & (2^bits − 1): leaves every entry's position in p0[]
    // (valid lanes) and `longest` = the largest bucket.  `cnt` must be zero on entry; on exit cnt[3·BINS + d] = end of bucket d
    // once every wave has ranked (barrier).
    uint32_t p0[Q];
    uint32_t longest = 0u;
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

    // ---- route 1: one pass on the top digit, then rank inside the buckets --------------------------------------------
    const uint32_t top_bits = min(nbits, (uint32_t)GGR_TSORT_BITS), top_shift = nbits - top_bits;
    counting_pass(top_shift, top_bits);
    if (top_shift == 0u) {   // the digit was the whole key: done
        TS_GROUPS({ if (valid[r]) list[p0[r]] = id[r]; })
        return;
    }
    if (longest <= GGR_TSORT_RANK_MAX) {
        // (only the KEYS travel: an entry stays in its thread's registers and is ranked where it is)
        TS_GROUPS({ if (valid[r]) ex[p0[r]] = ky[r]; })
        __syncthreads();   // (every wave has ranked: cnt[3][d] = end of bucket d)
        const uint32_t* ends = cnt + 3 * GGR_TSORT_BINS;
        const uint32_t top_mask = (1u << top_bits) - 1u;
#pragma unroll
        for (int g_ = 0; g_ < Q / 4; g_++)
            if ((uint32_t)(4 * g_) < q) {
                uint32_t bs[4], be[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t d = (ky[4 * g_ + u] >> top_shift) & top_mask;
                    be[u] = ends[d];
                    bs[u] = d ? ends[d - 1u] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = 4 * g_ + u;
                    // members in front of this one in (key, position) order: positions inside a bucket are in id order
                    // (the pass was stable)
                    // — one 64-bit compare of (key, position) per member
                    uint32_t rank = 0u;
                    const unsigned long long mine = ((unsigned long long)ky[r] << 32) | p0[r];
                    if (valid[r])
                        for (uint32_t j = bs[u]; j < be[u]; j++)
                            rank += ((((unsigned long long)ex[j]) << 32 | j) < mine) ? 1u : 0u;
                    if (valid[r]) list[bs[u] + rank] = id[r];
                }
            }
        return;
    }
    // ---- route 2 (clustered depths): LSD passes over evenly split digits ----------------------------------------------
    const uint32_t npass = (nbits + GGR_TSORT_BITS - 1) / GGR_TSORT_BITS;
    const uint32_t bits = (nbits + npass - 1) / npass;
    for (uint32_t pass = 0; pass < npass; pass++) {
        // the wave's counters back to zero (its own ranking is behind it: in-order LDS; the scan of the pass that wrote them
        // lies behind a barrier)
        for (uint32_t d = lane; d < GGR_TSORT_BINS; d += 64) my_cnt[d] = 0u;
        if (pass == 0) __syncthreads();   // (route 1's scan wrote EVERY wave's counters)
        counting_pass(pass * bits, bits);
        if (pass + 1 == npass) {
            TS_GROUPS({ if (valid[r]) list[p0[r]] = id[r]; })
            break;
        }
        // the exchange buffer holds one word per entry: the ids first, then the keys
        TS_GROUPS({ if (valid[r]) ex[p0[r]] = id[r]; })
        __syncthreads();
        TS_GROUPS({ id[r] = ex[min(p_lo + 64u * r, n - 1u)]; })
        __syncthreads();
        TS_GROUPS({ if (valid[r]) ex[p0[r]] = ky[r]; })
        __syncthreads();
        TS_GROUPS({ ky[r] = ex[min(p_lo + 64u * r, n - 1u)]; })
        // (the next pass's exchange writes come behind two more barriers: every thread has read its entries by then)
    }
#undef TS_GROUPS
}

template <int Q>
static void launch_tsort_class(size_t T, const uint2* ranges, uint32_t* point_list, const uint32_t* keys, uint32_t min_len,
                               uint32_t cap, hipStream_t s) {
    const size_t lds = (size_t)cap * 4 + 4 * GGR_TSORT_BINS * 4 + 4 * 64 * 8 + 64;
    if (lds > 64 * 1024)   // (a workgroup may take the CU's whole 160 KB, but beyond 64 KB it has to be asked for)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_depth_sort_kernel<Q>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tile_depth_sort_kernel<Q>, dim3((unsigned)T), dim3(256), lds, s, (uint32_t)T, ranges, point_list, keys,
                       min_len, cap);
}

void launch_tile_depth_sort(size_t T, const uint2* ranges, uint32_t* point_list, const uint32_t* keys, uint32_t min_len,
                            uint32_t max_len, hipStream_t s) {
    if (T == 0 || max_len < 2u || max_len <= min_len) return;
    const uint32_t cap = (max_len + 255u) & ~255u;   // (the exchange buffer: what the longest list of the launch needs)
    if (max_len <= GGR_TSORT_CAP_SMALL) launch_tsort_class<GGR_TSORT_CAP_SMALL / 256>(T, ranges, point_list, keys, min_len, cap, s);
    else launch_tsort_class<GGR_TSORT_CAP_LARGE / 256>(T, ranges, point_list, keys, min_len, std::min<uint32_t>(cap, GGR_TSORT_CAP_LARGE), s);
}

}  // namespace ggr
