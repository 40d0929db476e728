// tile_sort.h — the per-tile depth sort as a workgroup-level device function (gfx950): used by the stand-alone kernel
// (tile_sort.hip) and by the forward blend's prologue (blend_fwd.hip).  See tile_sort.hip for what it does and why.
#pragma once
#include "ggr_common.h"

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


// One 256-thread workgroup sorts the n <= cap <= 256·Q entries `pairs[0..n)` = (id, key) by (key, id) and writes the ids to
// `list[0..n)`.  `lds`: tsort_lds_words(cap) words (16-B aligned) that the caller may reuse behind a barrier.  Every thread of
// the workgroup must call it (barriers inside); n >= 2.
// REL (the depth sort's bucket form, binning.hip: the "list" is a depth bucket of the P Gaussians): the digits are cut out of
// the keys' distance from the list's smallest key instead of the bits that differ (see below), and every id is also followed by
// its 8-byte payload, gdst[position] = gsrc[id] (gdst: the list's own slice).  The per-tile sort (REL = false) compiles to what
// it was: its small class lives at 64 registers, and the range form cost it nine more spilled ones (43 -> 52 µs at C3).
template <int Q, bool REL = false>
__device__ __forceinline__ void tile_sort_body(uint32_t* __restrict__ lds, uint32_t cap, uint32_t n, uint32_t* __restrict__ list,
                                               const uint2* __restrict__ pairs, uint32_t* __restrict__ lsd_entries,
                                               const uint2* __restrict__ gsrc = nullptr, uint2* __restrict__ gdst = nullptr) {
    uint32_t fpos[REL ? Q : 1];   // REL: every entry's final position (ids and payloads are stored together at the end)
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
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
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
    // REL: the payloads are requested NOW — random 8-byte reads, 20 of the kernel's 39 µs when they were issued behind the
    // sort — and wait in registers until their entry knows its position (an unused slot repeats the list's last entry)
    uint2 payq[REL ? Q : 1];
    if (REL) { TS_GROUPS({ payq[REL ? r : 0] = gsrc[id[r]]; }) }
    auto finish_rel = [&]() {
        if (!REL) return;
        TS_GROUPS({ if (valid[r]) { list[fpos[REL ? r : 0]] = id[r]; gdst[fpos[REL ? r : 0]] = payq[REL ? r : 0]; } })
    };
    // Which bits count.  Per tile: the bits that DIFFER inside the list (OR ^ AND over its keys — a tile's keys spread over
    // octaves).  REL: the keys' RANGE — from here on a key is its distance from the list's smallest (same order): a depth
    // bucket is a narrow range, and a narrow range that straddles a high bit's boundary differs in that bit and in nothing
    // between it and its own width: its top nine "differing" bits take two values and every bucket went down the slow route.
    uint32_t diff;
    if (REL) {
        uint32_t k_max = 0u, k_min = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < Q; r++)
            if (valid[r]) { k_max = max(k_max, ky[r]); k_min = min(k_min, ky[r]); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            k_max = max(k_max, (uint32_t)__shfl_xor((int)k_max, off));
            k_min = min(k_min, (uint32_t)__shfl_xor((int)k_min, off));
        }
        for (uint32_t d = tid; d < GGR_TSORT_BINS; d += 256) fill[d] = 0u;
        if (lane == 0) { red[wave] = k_max; red[4 + wave] = k_min; }
        __syncthreads();
        const uint32_t lo_key = min(min(red[4], red[5]), min(red[6], red[7]));
        diff = max(max(red[0], red[1]), max(red[2], red[3])) - lo_key;
#pragma unroll
        for (int r = 0; r < Q; r++) ky[r] = valid[r] ? ky[r] - lo_key : 0u;
    } else {
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
        diff = (red[0] | red[1] | red[2] | red[3]) ^ (red[4] & red[5] & red[6] & red[7]);
    }
    if (diff == 0u) {   // every key equal: the entries are in id order already
        if (REL) {
#pragma unroll
            for (int r = 0; r < Q; r++) fpos[REL ? r : 0] = p_lo + 64u * r;
            finish_rel();
        } else {
            TS_GROUPS({ if (valid[r]) list[p_lo + 64u * r] = id[r]; })
        }
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
                        if (REL) fpos[REL ? r : 0] = bs[u] + rank;
                        else if (valid[r]) list[bs[u] + rank] = id[r];
                    }
                }
            finish_rel();
            return;
        }
    }
    // ---- route 2 (a bucket too large to rank by counting — depths clustered in few buckets): stable LSD passes over evenly
    // split digits.  It relies on the entries arriving in id order (the id-order scatter is stable): equal keys keep it.
    // (what the frame's tiles of this kind hold is counted: a host that finds most of a frame here does better with the
    //  global depth sort — ggr_sort_stats_async)
    if (tid == 0 && lsd_entries) atomicAdd(lsd_entries, n);
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
            if (REL) {
                // (this route's passes hand the ids on from thread to thread: the payloads requested at the start belong to
                //  other entries by now — fetched again, for the ids held at last)
                TS_GROUPS({ payq[REL ? r : 0] = gsrc[valid[r] ? id[r] : 0u]; })
#pragma unroll
                for (int r = 0; r < Q; r++) fpos[REL ? r : 0] = p0[r];
                finish_rel();
            } else {
                TS_GROUPS({ if (valid[r]) list[p0[r]] = id[r]; })
            }
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

}  // namespace ggr
