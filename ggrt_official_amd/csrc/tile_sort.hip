// tile_sort.hip — per-tile sort of the tile lists by (depth key, Gaussian id) (gfx950).
//
// Second form of the binning (the first: a global depth sort of the P Gaussians in front of the tile-list build,
// binning.hip).  Replaces duplicateWithKeys + the 64-bit radix sort + identifyTileRanges of the rasterizer behind
// reference cuda_splatting.py:114-125 with the same lists, bit for bit:
//
//   * the tile-list builder (tile_lists.hip) walks the Gaussians in ID order instead of depth order — its chunks are runs of
//     ids, the count needs no sorted input — and the scatter writes every list entry as (id, depth key);
//   * one workgroup per tile then sorts its entries by (key, id) — the order of the reference's stable sort on
//     (tile << 32 | depth bits) with values emitted in id order — and writes the ids into the list.
//
// What it buys: the global sort is four launches whose duration is a chain of cross-XCD look-backs (78-105 µs for 1 M keys
// that are 52 MB, NOTES r5 floor table), all of it on the forward's critical path; the per-tile sorts are independent
// workgroups with no global dependency.  What it costs: N list entries are sorted instead of P ≪ N Gaussians — in LDS.
//
// The workgroup-level algorithm (tile_sort.h `tile_sort_body`): the entries live in REGISTERS (Q rounds of 64 positions per
// wave), LDS is the exchange.  Only the key bits that differ inside the tile count (OR ^ AND over its keys: a frame-wide
// log-uniform depth spread leaves 25-26).
//   route 1  the (key, id) order is TOTAL (ids are unique), so nothing needs to be stable: one shared histogram of the top
//            <= 9 differing bits (an order-free ds_add per entry), a scan, one ds_add_rtn per entry on its bucket's fill pointer
//            (whatever it returns is a free slot of the bucket), and every entry counts the members of its bucket that precede
//            it — one 64-bit compare per member, two to four members at these sizes.  No lane matching, no ballots.
//   route 2  a tile whose largest bucket exceeds GGR_TSORT_RANK_MAX (depths clustered in few buckets): stable LSD passes over
//            evenly split digits; per-wave counters, lanes of a round that hold the same digit matched through per-wave
//            lane-mask words (atomic OR, read, store 0, round after round without waiting: one wave's LDS operations execute in
//            order — the idiom of bin_scatter), the lowest lane of each digit group reserves with one ds_add_rtn, the others
//            get it by ds_bpermute.  Relies on the scatter's id order for equal keys.
// Five versions and what each measured: NOTES r6 (124.7 -> 42 µs at C3).
#include "tile_sort.h"
#include <algorithm>

namespace ggr {

// Q: rounds per wave the registers hold — lists of up to 256·Q entries; cap ≤ 256·Q sizes the exchange buffer
template <int Q>
__global__ void __launch_bounds__(256) GGR_TSORT_WAVES(Q)
tile_depth_sort_kernel(uint32_t T, const uint2* __restrict__ ranges, uint32_t* __restrict__ point_list,
                       const uint2* __restrict__ pair_list, uint32_t min_len, uint32_t cap, int copy_longer,
                       uint32_t* __restrict__ lsd_entries) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tile = blockIdx.x;
    if (tile >= T) return;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    if (n <= min_len) return;   // (uniform: before any barrier; another launch's class)
    const uint32_t tid = threadIdx.x;
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
    tile_sort_body<Q>(lds, cap, n, list, pairs, lsd_entries);
}

// The same for the depth sort's bucket form (binning.hip): a "tile" is a depth bucket of the P Gaussians — a thousand
// workgroups, not eight thousand, so the small class need not be held to 64 registers (no spills; four workgroups per CU).
template <int Q>
__device__ __forceinline__ void bucket_sort_one(uint32_t* lds, uint2 range, uint32_t* __restrict__ point_list,
                                                const uint2* __restrict__ pair_list, uint32_t cap, int copy_longer,
                                                const TileSortExtras& ex) {
    const uint32_t tid = threadIdx.x;
    const uint32_t n = range.y - range.x;
    uint32_t* list = point_list + range.x;
    const uint2* pairs = pair_list + range.x;
    uint2* gdst = ex.gather_dst + range.x;
    if (n > cap) {
        // a bucket of more keys than a workgroup sorts (an overfull fine bin): left in id order — the stable partition's —
        // it is sorted iff all its keys are equal (a plane of constant depth); otherwise the fault bit, and the caller
        // sorts the frame again in three passes
        if (copy_longer) {
            uint32_t k_or = 0u, k_and = 0xFFFFFFFFu;
            // (one workgroup, n of any size: eight entries per thread and trip, every load of a trip in flight together)
            for (uint32_t i0 = 0; i0 < n; i0 += 8u * 256u) {
                uint2 e[8], pay[8];
#pragma unroll
                for (int u = 0; u < 8; u++) e[u] = pairs[min(i0 + (uint32_t)u * 256u + tid, n - 1u)];
#pragma unroll
                for (int u = 0; u < 8; u++) pay[u] = ex.gather_src[e[u].x];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t i = i0 + (uint32_t)u * 256u + tid;
                    if (i < n) {
                        list[i] = e[u].x;
                        k_or |= e[u].y; k_and &= e[u].y;
                        gdst[i] = pay[u];
                    }
                }
            }
            if (ex.fault_word && k_or != k_and) atomicOr(ex.fault_word, GGR_FAULT_BUCKET);
        }
        return;
    }
    if (n < 2u) {
        if (n == 1u && tid == 0) {
            const uint32_t id = pairs[0].x;
            list[0] = id;
            gdst[0] = ex.gather_src[id];
        }
        return;
    }
    tile_sort_body<Q, true>(lds, cap, n, list, pairs, nullptr, ex.gather_src, gdst);
}

template <int Q>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(Q <= 8 ? 4 : Q <= 16 ? 2 : 1, 8)))
bucket_sort_kernel(uint32_t T, const uint2* __restrict__ ranges, uint32_t* __restrict__ point_list,
                   const uint2* __restrict__ pair_list, uint32_t min_len, uint32_t cap, int copy_longer, TileSortExtras ex) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tile = blockIdx.x;
    if (ex.zero_words)   // (the words the three-pass form's last pass clears)
        for (uint32_t wz = blockIdx.x * 256u + threadIdx.x; wz < ex.zero_words; wz += gridDim.x * 256u) ex.zero_area[wz] = 0u;
    if (tile >= T) return;
    const uint2 range = ranges[tile];
    if (range.y - range.x <= min_len) return;
    bucket_sort_one<Q>(lds, range, point_list, pair_list, cap, copy_longer, ex);
}

// the buckets beyond the regular class: workgroup (x, segment) looks at ranges [16·x, 16·x + 16) of its segment
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 8)))
bucket_sort_big_kernel(const uint2* __restrict__ ranges, uint32_t* __restrict__ point_list, const uint2* __restrict__ pair_list,
                       uint32_t min_len, TileSortExtras ex) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t first = blockIdx.y * GGR_SORT_MAX_BINS + blockIdx.x * 16u;
    const uint2 mine = ranges[first + (threadIdx.x & 15u)];   // (lanes 0-15 of every wave hold the sixteen ranges)
    uint32_t nbig = 0u;
    for (int j = 0; j < 16; j++) {
        const uint2 range = make_uint2((uint32_t)__shfl((int)mine.x, j), (uint32_t)__shfl((int)mine.y, j));
        if (range.y - range.x <= min_len) continue;   // (uniform)
        __syncthreads();   // (the previous bucket's LDS)
        bucket_sort_one<32>(lds, range, point_list, pair_list, GGR_TSORT_CAP_LARGE, 1, ex);
        nbig++;
    }
    // how much of this was serial — buckets a workgroup sorted behind its first: much = depths concentrated in a small part of
    // the frame's range (a few far outliers and the rest inside an octave: every bucket is a single overfull fine bin), and then
    // this launch, 64 workgroups per view, is the slow way to sort them: the host is told (GGR_DEPTH_SORT_GLOBAL_SLOW) and keeps
    // such a shape on the three passes
    if (nbig > 1u && threadIdx.x == 0 && ex.fault_word) atomicAdd(ex.fault_word + (GGR_HIST_MSD_BIG - GGR_HIST_FAULT), nbig - 1u);
}

void launch_bucket_sort_big(uint32_t segments, const uint2* ranges, uint32_t* point_list, const uint2* pair_list, uint32_t min_len,
                            hipStream_t s, TileSortExtras ex) {
    const size_t lds = (size_t)tsort_lds_words(GGR_TSORT_CAP_LARGE) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sort_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bucket_sort_big_kernel, dim3(GGR_SORT_MAX_BINS / 16, segments ? segments : 1), dim3(256), lds, s, ranges, point_list,
                       pair_list, min_len, ex);
}

template <int Q>
static void launch_tsort_class(size_t T, const uint2* ranges, uint32_t* point_list, const uint2* keys, uint32_t min_len,
                               uint32_t cap, int copy_longer, uint32_t* lsd_entries, hipStream_t s, const TileSortExtras* ex) {
    const size_t lds = (size_t)tsort_lds_words(cap) * 4;
    if (ex) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sort_kernel<Q>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(bucket_sort_kernel<Q>, dim3((unsigned)T), dim3(256), lds, s, (uint32_t)T, ranges, point_list, keys,
                           min_len, cap, copy_longer, *ex);
        return;
    }
    if (lds > 64 * 1024)   // (a workgroup may take the CU's whole 160 KB, but beyond 64 KB it has to be asked for)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_depth_sort_kernel<Q>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tile_depth_sort_kernel<Q>, dim3((unsigned)T), dim3(256), lds, s, (uint32_t)T, ranges, point_list, keys,
                       min_len, cap, copy_longer, lsd_entries);
}

void launch_tile_depth_sort(size_t T, const uint2* ranges, uint32_t* point_list, const uint2* keys, uint32_t min_len,
                            uint32_t max_len, hipStream_t s, int copy_longer, uint32_t* lsd_entries, const TileSortExtras* extras) {
    if (T == 0 || max_len == 0u || max_len <= min_len) return;   // (a list of ONE entry is still copied out of the pairs)
    const TileSortExtras* ex = extras;   // (the depth sort's bucket form: gather_src / gather_dst must be given)
    const uint32_t cap = std::min<uint32_t>((max_len + 255u) & ~255u, GGR_TSORT_CAP_LARGE);   // the exchange buffer of the launch
    // the class = rounds per wave the registers hold: 8 (lists up to 2048: 6 workgroups per CU), 12 (3072), 16 (4096), 32 (8192)
    if (cap <= 2048) launch_tsort_class<8>(T, ranges, point_list, keys, min_len, cap, copy_longer, lsd_entries, s, ex);
    else if (cap <= 3072) launch_tsort_class<12>(T, ranges, point_list, keys, min_len, cap, copy_longer, lsd_entries, s, ex);
    else if (cap <= 4096) launch_tsort_class<16>(T, ranges, point_list, keys, min_len, cap, copy_longer, lsd_entries, s, ex);
    else launch_tsort_class<32>(T, ranges, point_list, keys, min_len, cap, copy_longer, lsd_entries, s, ex);
}

}  // namespace ggr
