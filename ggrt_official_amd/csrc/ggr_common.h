// ggr_common.h — shared constants, HBM layouts and launch helpers for the gfx950 kernels.
//
// Data layout in HBM (all caller-owned, carved out of opaque buffers; 256-B aligned sections)
//
//   geom buffer   (per Gaussian, P entries; kept for backward)
//     splat[P]         2 × float4 = 32 B  {x, y, conic.xx, conic.xy | conic.yy, opacity, f, qmax}
//     colour[P]        1 × float4 = 16 B  {r, g, b, 0}
//                      together everything the blend kernels need for a list entry (f = view-space z or the
//                      caller's aux feature; qmax = 2·ln(255·opacity): the largest dᵀ·conic·d at which α still
//                      reaches 1/255): a tile-list fetch is a 32-B and a 16-B gather, both sector-aligned.  Two
//                      arrays because two kernels write them: the geometry half of preprocess_fwd sits on the
//                      forward's critical path in front of the depth sort, the colour half (the SH rows: 4/5 of
//                      the stage's bytes) runs on a side stream BESIDE the latency-bound sort and tile-list
//                      kernels and is only needed by the blend (preprocess.hip, api.hip forward_impl)
//     sh_jac[3][P]     float4 per channel c = r, g, b (three planes): {∂colour_c/∂dir.x, ∂colour_c/∂dir.y, ∂colour_c/∂dir.z, 0}, the Jacobian of the
//                      (unclamped) SH colour w.r.t. the unit view direction, left by the forward's colour evaluation while
//                      the SH row is in LDS anyway — the backward's view-direction term is then Jᵀ·dL/dcolour and it does
//                      not read the SH rows again (192 B per Gaussian at 16 coefficients, 300 B at GGRt's 25)
//     rect[P]          u32×2 packed tile rect (minx | miny<<16, maxx | maxy<<16)
//     clamped[P]       u32   bit c set ⇔ SH colour channel c was clamped at 0
//     cov3D[P]         6 × f32 (scale/rot path only; otherwise the caller's cov3D_precomp is used)
//     depth-sort double buffers (keys = depth bits − bits of 0.2f, 0 if culled; vals = ids), counters, radix work area
//   work buffer   (forward only, obtained from the allocator, may be released after ggr_forward)
//     table[chunks][tiles] u32, gsum[groups][tiles] u32, tile_start[tiles] u32   (tile_lists.hip)
//   binning buffer (kept for backward)
//     point_list[N] u32: Gaussian ids, tile by tile, each tile's run in (depth, id) order
//   image buffer  (kept for backward)
//     ranges[tiles] (uint2), final_T[H·W] f32, n_contrib[H·W] u32, tile_top[tiles] u32,
//     ckpt[16][5][H·W] f32 (images below 4096 tiles only: per-pixel checkpoints for the segmented backward)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define GGR_TILE 16
#define GGR_TILE_PIX 256
#define GGR_NEAR_CULL 0.2f
#define GGR_DILATION 0.3f
#define GGR_FRUSTUM_CLAMP 1.3f
#define GGR_ALPHA_MIN (1.0f / 255.0f)
#define GGR_ALPHA_MAX 0.99f
#define GGR_T_MIN 0.0001f

// depth sort geometry (binning.hip): 512 threads × 8 items = 4096 keys per sort tile (8 ranking rounds per wave: the
// rounds are a chain of dependent LDS round trips, and a tile per CU leaves room for 8 waves)
#define GGR_SORT_THREADS 512
#define GGR_SORT_ITEMS 8
#define GGR_SORT_TILE (GGR_SORT_THREADS * GGR_SORT_ITEMS)
// The sort key of a Gaussian is (float bits of its view depth) − GGR_KEY_BASE: every visible Gaussian has depth > 0.2
// (the near cull), so its bits exceed those of 0.2f; culled Gaussians carry key 0 (they touch no tile — where they end
// up in the order is irrelevant).  Subtracting the base removes the constant high part of the float bits: a scene
// with depths in [0.2, 13 000) has 27-bit keys, and the sort takes THREE passes of ⌈bits/3⌉-bit digits (≤ 10 bits, i.e.
// every depth below 6.8e37) instead of four 8-bit ones.
#define GGR_KEY_BASE 0x3E4CCCCDu  // __float_as_uint(0.2f)
#define GGR_KEY_MAX 0x3FFFFFFFu   // preprocess clamps here (3 × 10 bits)
#define GGR_SORT_PASSES 3
#define GGR_SORT_MAX_BITS 10      // per digit
#define GGR_SORT_MAX_BINS (1 << GGR_SORT_MAX_BITS)
#define GGR_PRE_THREADS 256       // preprocess_fwd block size: one key maximum per block is left for the sort

// tile-list builder: Gaussians (in depth order) per chunk
#ifndef GGR_BIN_CHUNK
#define GGR_BIN_CHUNK 1024
#endif
#ifndef GGR_COUNT_CPG
#define GGR_COUNT_CPG 4    // chunks per workgroup of the count kernel (their running per-tile counts stay in registers)
#endif
#define GGR_COUNT_SLOTS 12   // (row, 64-tile piece) pairs per count wave: bounds a count band; a tile row of more than
                             // 64·12 = 768 tiles (12 288 px) is counted in column windows (tile_lists.hip)
// per-tile depth sort (tile_sort.hip): list lengths of the two launch classes.  16 B of LDS per entry + 12 KB: a workgroup of
// the small class leaves room for four per CU, one of the large class has the CU to itself
#define GGR_TSORT_CAP_SMALL 2048
#define GGR_TSORT_CAP_LARGE 8192
#define GGR_COUNT_GROUPS 8 // groups of count workgroups: the prefix over workgroups runs per group (T·groups-way parallel)

// The kernels of the forward's critical path that run beside the colour kernel of the side stream (api.hip forward_impl:
// depth sort, tile counts) raise their waves' issue priority: a colour wave on the same SIMD then only issues in the slots
// they leave (s_setprio; -DGGR_PRIO=0: off)
#ifndef GGR_PRIO
#define GGR_PRIO 1
#endif
#if GGR_PRIO
#define GGR_CRITICAL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define GGR_CRITICAL_PRIO() ((void)0)
#endif

// ---- streaming accesses of the two preprocess kernels (round 4) ------------------------------------------------------------
// What this launch set never reads again — the gradient tensors, radii, clamp bits — is stored NON-TEMPORALLY, and what
// preprocess_bwd reads exactly once in whole lines — SH rows, gradient records, means, covariances — is loaded so: the lines
// neither stay in L2 / the Infinity Cache nor linger there dirty.  Measured at C3 (same box, `scripts/run_variants`-style A/B,
// HIP-event medians of 200 steps): step 0.885 → 0.861 ms; preprocess_bwd 0.129 → 0.118 ms, and preprocess_fwd — whose only
// change is 8 B per Gaussian of radii / clamp-bit stores — 0.075 → 0.063 ms: 245 MB of plainly stored gradients of the
// PREVIOUS step were still being written back while it streamed.  What did not help or hurt (NOTES r4): non-temporal loads of
// the forward's inputs (=), of SH rows read a third at a time (the thirds share lines: +0.022 ms), non-temporal stores of the
// splat records (re-read by the blend: =), of the image (=), of the scratch clearing (=).  (-DGGR_NT=0: plain accesses.)
#ifndef GGR_NT
#define GGR_NT 1
#endif
#if defined(__HIPCC__)
typedef float ggr_v4f __attribute__((ext_vector_type(4)));
#if GGR_NT
__device__ __forceinline__ float4 ggr_ld_f4(const float* p) {
    const ggr_v4f v = __builtin_nontemporal_load(reinterpret_cast<const ggr_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ggr_ld(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void ggr_st_f4(float* p, float4 v) {
    ggr_v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<ggr_v4f*>(p));
}
__device__ __forceinline__ void ggr_st(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void ggr_st(int32_t* p, int32_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void ggr_st(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ float4 ggr_ld_f4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float ggr_ld(const float* p) { return *p; }
__device__ __forceinline__ void ggr_st_f4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void ggr_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void ggr_st(int32_t* p, int32_t v) { *p = v; }
__device__ __forceinline__ void ggr_st(uint32_t* p, uint32_t v) { *p = v; }
#endif
#endif

static inline size_t ggr_align(size_t x) { return (x + 255) & ~(size_t)255; }

static inline size_t ggr_sort_blocks(size_t n) { return (n + GGR_SORT_TILE - 1) / GGR_SORT_TILE; }
// The sort is SEGMENTED: the n keys are S segments of n/S keys, each sorted on its own into its own range (S = 1: a
// plain sort).  A launch set of V views sorts V segments — the per-view orders are all the tile lists need — with V
// independent ticket sequences and look-back chains in the same three launches: a chain over the V·P/4096 tiles of all
// views is what made one 4 M-key sort cost as much as four 1 M-key sorts one after the other.
// Sort work area (u32 words): header [0, 256): tickets [pass·64 + segment], fault word, digit parameters | digit totals,
// (segment·3 + pass)·1024 + digit | look-back status words, ((pass·S + segment)·tiles_per_segment + tile)·bins + digit |
// one key maximum per preprocess block.  Everything before the block maxima is zeroed by preprocess_fwd
// (ggr_sort_zero_words); the maxima are plain stores.
#define GGR_SORT_MAX_SEGMENTS 64
#define GGR_HIST_TICKETS 0
#define GGR_HIST_FAULT 192
// values of the exact mode's pinned num_rendered word that are not counts (counts stop at 0x7FFFFFFF)
#define GGR_HOST_FAULT_SPIN 0xFFFFFFFFu    // a look-back spin of the depth sort hit its bound
#define GGR_HOST_ARMED 0xFFFFFFFEu         // not written yet
#define GGR_HOST_FAULT_RANGE 0xFFFFFFFDu   // a sort key beyond 30 bits (cannot happen behind preprocess_fwd, which clamps)
#define GGR_HIST_PARAMS 200   // [0] = bits per digit
#define GGR_HIST_TOTALS 256
static inline size_t ggr_sort_segments(size_t views) { return views >= 1 && views <= GGR_SORT_MAX_SEGMENTS ? views : 1; }
__host__ __device__ static inline size_t ggr_sort_status_base(size_t S) { return GGR_HIST_TOTALS + S * GGR_SORT_PASSES * GGR_SORT_MAX_BINS; }
// look-back status words per (pass, segment): GGR_SORT_LEVELS arrays of [tiles][MAX_BINS] — level 0 the tiles' own digit
// counts (and, in the walking look-back of big sorts, their inclusive prefixes), levels 1 / 2 the span-8 / span-64
// aggregates of the tree look-back (binning.hip)
#define GGR_SORT_LEVELS 3
#define GGR_SORT_TREE_MAX_TILES 512   // (tiles of 4096 keys) up to here all tiles of a sort are resident at once — as one
                                      // tile of up to 8192 keys per CU, or two per CU — and look back through the tree
                                      // (the tree's three levels span 8·8·8 tiles: do not raise without a fourth)
// The BUCKET form of the depth sort (binning.hip, round 6): ONE stable partition pass into <= 1024 depth buckets per segment whose
// boundaries are equal-frequency splitters of a 4096-bin histogram of the frame's own key range, then every bucket sorted in
// LDS by the per-tile sort's workgroup routine.  Its words in the work area: the fine histogram (per segment 4096 bins of
// visible keys + [4096] = the culled Gaussians, key 0), inside the zeroed part; one key MINIMUM (over the visible keys) per
// preprocess block behind the maxima; the buckets' ranges for the bucket-sort launch.
#define GGR_MSD_FINE 4096
#define GGR_MSD_FINE_WORDS 4352          // per segment (4097 used)
#define GGR_MSD_MAX_POINTS (2u << 20)    // per segment: beyond, buckets would exceed what a workgroup sorts in LDS
#define GGR_HIST_MSD_KMIN (GGR_HIST_PARAMS + 1)   // smallest visible key of the frame
#define GGR_HIST_MSD_SHIFT (GGR_HIST_PARAMS + 2)  // fine bin of a visible key = (key - kmin) >> shift
#define GGR_HIST_MSD_BIG (GGR_HIST_PARAMS + 3)    // buckets beyond the regular class that a workgroup of the small launch sorted BEHIND its first (zeroed)
#define GGR_MSD_BIG_MANY 16u                      // from here on the frame's depths are too concentrated for the bucket form
#define GGR_FAULT_SPIN 1u    // a look-back spin hit its bound
#define GGR_FAULT_RANGE 2u   // a key needs more than 3 x 10 bits (depth >= 6.8e37)
#define GGR_FAULT_BUCKET 4u  // bucket form: a bucket of more than GGR_TSORT_CAP_LARGE keys that are not all equal was left unsorted
static inline size_t ggr_sort_lsd_zero_words(size_t n, size_t S = 1) {
    const size_t tps = ggr_sort_blocks((n ? n : 1) / S ? (n ? n : 1) / S : 1);
    const size_t levels = tps * S <= GGR_SORT_TREE_MAX_TILES ? GGR_SORT_LEVELS : 1;
    return ggr_sort_status_base(S) + (size_t)GGR_SORT_PASSES * S * tps * GGR_SORT_MAX_BINS * levels;
}
// words of the work area that start from zero (cleared by preprocess_fwd): header, digit totals, status words, fine histogram
static inline size_t ggr_sort_zero_words(size_t n, size_t S = 1) { return ggr_sort_lsd_zero_words(n, S) + S * GGR_MSD_FINE_WORDS; }
__host__ __device__ static inline size_t ggr_sort_block_words(size_t n) {
    // (+ MAX_SEGMENTS: a launch set of several Gaussian sets rounds its preprocess blocks up per set — at most one block per view more)
    return ((n ? n : 1) + GGR_PRE_THREADS - 1) / GGR_PRE_THREADS + GGR_SORT_MAX_SEGMENTS;
}
// offsets (words) of the sections behind the zeroed part: block maxima | block minima | bucket ranges (uint2 per bucket)
static inline size_t ggr_sort_block_max_at(size_t n, size_t S = 1) { return ggr_sort_zero_words(n, S); }
static inline size_t ggr_sort_block_min_at(size_t n, size_t S = 1) { return ggr_sort_block_max_at(n, S) + ggr_sort_block_words(n); }
static inline size_t ggr_sort_bucket_ranges_at(size_t n, size_t S = 1) { return (ggr_sort_block_min_at(n, S) + ggr_sort_block_words(n) + 1) & ~(size_t)1; }
static inline size_t ggr_sort_hist_words(size_t n, size_t S = 1) { return ggr_sort_bucket_ranges_at(n, S) + S * 2 * GGR_SORT_MAX_BINS; }

struct GeomLayout {
    float4* splat;    // [P][2]
    float4* colour;   // [P]
    float4* sh_jac;   // [3][P]
    uint2* rect;
    uint32_t* clamped;
    float* cov3D;
    uint32_t* keys_a;
    uint32_t* keys_b;
    uint32_t* vals_a;
    uint32_t* vals_b;
    uint32_t* hist;       // sort work area (ggr_sort_hist_words)
    uint32_t* counters;   // [64]
    size_t bytes;
};

// P = (view, Gaussian) pairs of the launch set; `segments` = ggr_sort_segments(views).  `with_jac` = false: the geometry
// buffer of a forward that no backward will follow (GgrForwardOut.no_backward, ggr_geom_bytes_inference) — the Jacobian planes
// (48 B per pair, written by a training forward for its backward) come LAST and are left out; every other section sits where it
// sits in the full layout
static inline GeomLayout ggr_carve_geom(void* base, size_t P, size_t segments = 1, bool with_jac = true) {
    GeomLayout L;
    char* p = (char*)base;
    size_t o = 0;
    size_t Pp = P ? P : 1;
    auto take = [&](size_t bytes) { char* r = p ? p + o : nullptr; o += ggr_align(bytes); return r; };
    L.splat = (float4*)take(Pp * 32);
    L.colour = (float4*)take(Pp * 16);
    L.rect = (uint2*)take(Pp * 8);
    L.clamped = (uint32_t*)take(Pp * 4);
    L.cov3D = (float*)take(Pp * 24);
    L.keys_a = (uint32_t*)take(Pp * 4);
    L.keys_b = (uint32_t*)take(Pp * 4);
    L.vals_a = (uint32_t*)take(Pp * 4);
    L.vals_b = (uint32_t*)take(Pp * 4);
    L.counters = (uint32_t*)take(64 * 4);  // (before the sort area: its offset must not depend on `segments`)
    L.hist = (uint32_t*)take(ggr_sort_hist_words(Pp, segments) * 4);
    L.sh_jac = with_jac ? (float4*)take(Pp * 48) : nullptr;
    L.bytes = o;
    return L;
}

static inline size_t ggr_point_list_bytes(size_t N) { return ggr_align((N ? N : 1) * 4); }
// bytes of the (id, key) entries the id-order scatter writes for the per-tile depth sort (tile_sort.hip): scratch of one forward
static inline size_t ggr_pair_list_bytes(size_t N) { return ggr_align((N ? N : 1) * 8); }

// Depth segments of the blend backward (blend_bwd.hip).  An image with few tiles cannot fill 256 CUs with one
// workgroup per tile (480×352 = 660 tiles = 2.6 waves per SIMD: the backward ran at half the per-entry rate of
// a 1080p frame), so there the forward leaves per-pixel CHECKPOINTS (T, Σw·c, Σw·z) at up to `slots − 1` list
// positions per tile (every `ckpt_stride` entries, blend_common.h) and the backward replays each interval in
// its own workgroup.  The slot count is a function of the image size only (the image-state buffer is sized
// before anything runs): 16 below 4096 tiles (320 B per pixel, ≤ 0.34 GB), none from there on — a 1080p frame
// fills the chip as it is (measured: 16 slots at 8160 tiles, 0.44 → 0.48 ms).
// Blend backward, no checkpoints → 16 slots:  480×352, 1.01 M pixel-aligned Gaussians 0.434 → 0.271 ms;
// 960×640, 4.9 M: 1.03 → 0.86 ms; 256×256, 10 k: 0.050 → 0.038 ms.  Forward cost of writing them: < 1 %.
// (4 views of 660 tiles = 2640 tiles in one launch set: 16 / 8 / 4 slots measured the same within noise)
static inline int ggr_ckpt_slots(size_t tiles) {  // slot 0 holds the final sums; 0 = no checkpoints
    return (tiles == 0 || tiles >= 4096) ? 0 : 16;
}
static inline int ggr_bwd_segments(size_t tiles) { const int k = ggr_ckpt_slots(tiles); return k ? k : 1; }
#define GGR_CKPT_FLOATS 5  // T, Σw·r, Σw·g, Σw·b, Σw·z — each a [H·W] plane: ckpt[(slot·5 + v)·H·W + pixel]

struct ImageLayout {
    uint2* ranges;
    float* final_T;
    uint32_t* n_contrib;
    uint32_t* tile_top;  // [tiles]: max n_contrib of the tile = list entries the backward replays
    float* ckpt;         // [slots][5][H·W] or null
    int ckpt_slots, bwd_segments;
    size_t bytes, bytes_no_ckpt;
};

// (V views of one launch set: V frames stacked — tiles and pixels of view v follow those of view v-1 in every array)
static inline ImageLayout ggr_carve_image(void* base, int W, int H, int V = 1) {
    ImageLayout L;
    char* p = (char*)base;
    size_t o = 0;
    size_t tiles = (size_t)((W + GGR_TILE - 1) / GGR_TILE) * ((H + GGR_TILE - 1) / GGR_TILE) * (size_t)V;
    size_t pix = (size_t)W * H * (size_t)V;
    auto take = [&](size_t bytes) { char* r = p ? p + o : nullptr; o += ggr_align(bytes ? bytes : 4); return r; };
    L.ranges = (uint2*)take(tiles * 8);
    L.final_T = (float*)take(pix * 4);
    L.n_contrib = (uint32_t*)take(pix * 4);
    L.tile_top = (uint32_t*)take(tiles * 4);
    L.ckpt_slots = ggr_ckpt_slots(tiles);
    L.bwd_segments = ggr_bwd_segments(tiles);
    L.bytes_no_ckpt = o;  // (the checkpoint area comes last: an inference-only forward does without it)
    L.ckpt = L.ckpt_slots ? (float*)take((size_t)L.ckpt_slots * GGR_CKPT_FLOATS * pix * 4) : nullptr;
    L.bytes = o;
    return L;
}

// backward scratch: per-Gaussian accumulators filled by the blend backward
// One 64-byte record per Gaussian, so that the ≤ 10 atomic adds a wave commits for a list entry fall into
// ONE cache line: the blend backward was bound by atomic line transactions, not by arithmetic (spread over
// four arrays: 0.89 ms at C3, in one line: 0.70 ms).  Float index inside the record:
#define GGR_G2D_RGB 0      // 0..2  dL/dcolour
#define GGR_G2D_MEAN 3     // 3..4  dL/dmean2D (NDC-scaled x, y)
#define GGR_G2D_CONIC 5    // 5..7  dL/dconic (xx, xy[half convention], yy)
#define GGR_G2D_OPACITY 8  // 8     dL/dopacity
#define GGR_G2D_Z 9        // 9     dL/d(depth or aux feature); 10..15 unused
#define GGR_G2D_STRIDE 16
struct BwdScratch {
    float* grad2d;    // [P][16]
    float* pose_acc;  // rows of 64: dL/dviewmatrix [0:16], dL/dprojmatrix [16:32], dL/dcampos [32:35] partials
    size_t bytes;
};

static inline BwdScratch ggr_carve_bwd(void* base, size_t P, size_t V = 1) {
    BwdScratch L;
    char* p = (char*)base;
    size_t o = 0;
    size_t Pp = P ? P : 1;
    auto take = [&](size_t bytes) { char* r = p ? p + o : nullptr; o += ggr_align(bytes); return r; };
    L.grad2d = (float*)take(Pp * V * GGR_G2D_STRIDE * 4);                 // one record per (view, Gaussian)
    L.pose_acc = (float*)take((64 + V * ((Pp + 255) / 256) * 64) * 4);    // one row of partials per (view, block)
    L.bytes = o;
    return L;
}

// Input forms (call-site fusion): what reference render_cuda does with torch ops on the P-sized
// tensors before every rasterizer call is applied on load instead (and chained through in backward).
struct InputForm {
    int cov_stride;            // 6: [P,6] upper triangle.  9: [P,3,3] row-major, entries (0,1,2,4,5,8) used —
                               // the triu gather of :116,124
    int sh_channel_major;      // 0: [P,M,3] (upstream).  1: [P,3,M] = GGRt's harmonics layout — saves the
                               // transpose copy of :77 and its backward
    int aux_affine;            // 1: blended feature = max(aux_a + aux_b·z/s, 0) instead of z — GGRt's depth pass
    float aux_a, aux_b;        //    (:240-269: depth as a degree-0 SH coefficient) without a per-Gaussian tensor
    int sh_cap;                // highest SH band evaluated: 3 (graphdeco / w-depth family, default) or 4
    int sc_x0, sc_y0, sc_x1, sc_y1;  // GgrSettings.scissor in TILES, half-open, inside the tile grid (whole grid = none)
    int sh_aligned;            // every set's SH rows (and gradient rows) start 16-B aligned: flat float4 staging allowed
    int tight_rects;           // 1 (default): tile rects clipped to the α ≥ 1/255 ellipse's bounding box (ggr_tighten_rect)
};

// ---- TIGHT tile rects (round 3; restated in oracle/ggr_oracle.c `tighten_rect`, same operations in the same order) ----
// The reference lists a Gaussian in every tile of the square of radius ceil(3·sqrt(λmax)) around its mean.  A pixel can
// only take it if α = opacity·exp(−q/2) ≥ 1/255, i.e. inside the ellipse q ≤ qmax = 2·ln(255·opacity), whose bounding box
// has the half widths sqrt(qmax·cov_xx), sqrt(qmax·cov_yy): tiles outside that box (half a pixel and 1 % to spare) hold
// only pixels that `continue` past the Gaussian.  Dropping them changes no output — images, final_T, radii, every gradient
// are bit-identical (oracle-checked) — only the lists get shorter (C3: 10.76 M → 8.0 M entries).  The bound on ln comes from
// the float's exponent and a cubic in its mantissa (≥ ln on [1, 2)): exact-order fp32 operations, so host oracle and
// kernel agree on every rect to the bit, which libm's logf and the device's would not.
__host__ __device__ static inline float ggr_qmax_upper(float opacity) {   // ≥ 2·ln(255·opacity); negative: α < 1/255 everywhere
#pragma clang fp contract(off)   // (no FMA contraction: the oracle's plain fp32 operations, bit for bit)
    const float u = 255.0f * opacity;
    if (!(u >= 1.0f)) return -1.0f;
    union { float f; uint32_t b; } cv;
    cv.f = u;
    const int e = (int)(cv.b >> 23) - 127;
    cv.b = (cv.b & 0x007FFFFFu) | 0x3F800000u;
    const float x = cv.f - 1.0f, t = x * x;
    const float lnm = (x - 0.5f * t) + 0.33333334f * (t * x);
    return 2.0f * ((float)e * 0.69314718f + lnm) + 0.02f;
}

// The cameras of one launch set: V views of the SAME P Gaussians (V = 1: the reference's call).  Per-Gaussian state of
// view v lives at index v·P + g; its tiles are tiles [v·T, (v+1)·T) of a virtual image of V frames stacked vertically
// (tile row = v·grid_y + row), so the depth sort, the tile-list builder and the blend kernels run ONCE over all views
// (replaces the per-view loop of reference cuda_splatting.py:93-127).
struct ViewSet {
    int V;
    int sets, vps;             // the V views are `sets` groups of `vps` views; group b renders Gaussian set b of the
                               // caller's [sets, P, …] inputs (sets = 1: every view renders the same P Gaussians)
    const float* view;         // device [V,16]
    const float* proj;         // device [V,16]
    const float* campos;       // device [V,3]
    const float* bg;           // device [V,3]
    const float* tanfov;       // device [V,2] or NULL → tanfovx / tanfovy below for every view
    const float* input_scale;  // device [V] or NULL (= 1): means·s, cov·s², scales·s — the 1/near renormalisation of
                               // cuda_splatting.py:66-73
    float tanfovx, tanfovy;
};

// SH bands actually evaluated: min(D, cap), and never more than a row of M coefficients holds
__host__ __device__ static inline int ggr_sh_degree(int D, int M, int cap) {
    int deg = D < cap ? D : cap;
    if (deg < 0) deg = 0;
    while (deg > 0 && (deg + 1) * (deg + 1) > M) deg--;
    return deg;
}

// ---- kernel launchers (defined in the .hip translation units) -------------------------------
namespace ggr {

// geom `g` is carved for V·P Gaussians; radii [V,P]; aux_precomp [V,P] or NULL
void launch_preprocess_fwd(int P, int D, int M, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, float scale_modifier, const float* cov3D_precomp,
                           const float* aux_precomp, ViewSet vs, int W, int H, int32_t* radii,
                           GeomLayout g, InputForm inf, hipStream_t s,
                           int part = 0 /*GGR_PRE_ALL, GGR_PRE_GEOMETRY, GGR_PRE_COLOUR (preprocess.hip)*/,
                           int colour_grid = 0 /*COLOUR: persistent blocks walking the chunks; 0 = one block per chunk*/,
                           int keep_jacobian = 1 /*0: no backward will follow (inference) — sh_jac is not written*/,
                           uint32_t* zero_area2 = nullptr /*also cleared (the tile-list builder's per-tile totals when no
                           depth sort runs in front of it)*/, uint32_t zero_words2 = 0,
                           int sort_area_untouched = 0 /*1: the depth sort's work area is NOT cleared (no sort will run)*/);
#define GGR_PRE_ALL 0
#define GGR_PRE_GEOMETRY 1
#define GGR_PRE_COLOUR 2

// stable LSD radix sort of (u32 key, u32 val) pairs, keys in the depth-sort form above (< 2^30, else the fault word
// is raised); three passes; returns the buffers that hold the result (b after three passes).  The key maxima of the
// n/256 producer blocks must be in the work area (preprocess_fwd; `block_max_ready` = false makes the sort compute
// them itself: tools/sort_bench.hip)
void radix_sort_pairs(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                      uint32_t* hist, size_t n, uint32_t segments /*n must be a multiple*/, uint32_t** keys_out,
                      uint32_t** vals_out, hipStream_t s,
                      bool hist_zeroed = false /*the caller already cleared ggr_sort_zero_words(n, segments)*/,
                      uint32_t block_max_ready = 0 /*> 0: that many key maxima are already in the work area*/,
                      bool identity_vals = false /*vals_a holds nothing: the first pass uses val = index in the whole array*/,
                      const uint2* gather_src = nullptr /*last pass also writes gather_dst[pos] = gather_src[val]*/,
                      uint2* gather_dst = nullptr, uint32_t* zero_area = nullptr /*and clears these words*/,
                      uint32_t zero_words = 0,
                      bool buckets = false /*the bucket form (binning.hip): needs identity_vals, block_max_ready (preprocess_fwd's
                      block maxima AND minima), the payload gather and n / segments <= GGR_MSD_MAX_POINTS — else three passes; keys_out is not filled (NULL); a bucket it
                      could not sort raises GGR_FAULT_BUCKET in the fault word — the order is then a permutation that is NOT
                      fully sorted and the caller must sort again without `buckets`*/);
// the bucket form is possible for this many keys per segment
static inline bool radix_sort_buckets_ok(size_t n_per_segment) { return n_per_segment > 0 && n_per_segment <= GGR_MSD_MAX_POINTS; }


// tile-list builder (tile_lists.hip)
struct TileListPlan {
    uint32_t nchunks, nsbands, sband_tiles;
    uint32_t nw, groups, wpg;   // count workgroups (GGR_COUNT_CPG chunks each), groups of them, workgroups per group
    size_t table_words, wsum_words, gsum_words, work_bytes;
};
TileListPlan plan_tile_lists(size_t P, size_t T);
// K1 + K2: fills the work area, ranges[T], total_out[0] (= N, device) and total_out[1] (= N > capacity).
// capacity = entries the list buffer can hold (sync-free mode: ranges are cut there); ~0u = sized exactly later
void launch_tile_list_count(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                            const uint2* rect, void* work, uint2* ranges, uint32_t* total_out, uint32_t capacity,
                            hipStream_t s, bool rects_gathered = false /*the depth sort already filled rect_sorted
                            and cleared the per-tile totals (tile_list_gather_targets)*/,
                            uint32_t* host_total = nullptr /*pinned, device-visible: also receives N …*/,
                            hipEvent_t after_scan = nullptr /*… and this event is recorded right behind the scan*/,
                            const uint32_t* sort_fault = nullptr /*device word the depth sort raises when a look-back
                            spin hit its bound: folded into total_out[1] (bit 1) and host_total (~0u)*/,
                            bool id_order = false /*no depth sort ran: chunks = runs of Gaussian ids, `rect` is read as it is
                            and the per-tile totals have been cleared by preprocess_fwd (tile_sort.hip)*/,
                            uint32_t list_limit = 0xFFFFFFFFu /*longest list the per-tile depth sort takes: a longer one
                            raises bit 3 of total_out[1]; total_out[2] / host_total[1] = the longest list*/);
// the depth sort's fault word inside its work area (binning.hip)
const uint32_t* radix_sort_fault_word(const uint32_t* hist);
// where the depth sort's last pass should put the rects in depth order / which words it should clear
void tile_list_gather_targets(const TileListPlan& pl, void* work, size_t T, uint2** rect_sorted,
                              uint32_t** zero_area, uint32_t* zero_words);
// K3: writes point_list[min(N, capacity)]; order == NULL: id order (rect = the Gaussians' rects in id order)
void launch_tile_list_ranges(const TileListPlan& pl, size_t T, void* work, uint2* ranges, hipStream_t s);
void launch_tile_list_scatter(const TileListPlan& pl, size_t P, size_t T, int grid_x, const uint32_t* order,
                              const uint2* rect, const void* work, uint32_t* point_list, uint32_t capacity,
                              hipStream_t s, const uint32_t* keys = nullptr /*id order: the Gaussians' depth keys …*/,
                              uint2* pair_list = nullptr /*… and where the (id, key) entries go instead of point_list*/);

// what the depth sort's bucket form (binning.hip) asks of the same kernels: its "tiles" are the depth buckets of the P keys
struct TileSortExtras {
    const uint2* gather_src;   // every sorted id also fetches its 8-byte payload: gather_dst[position] = gather_src[id]
    uint2* gather_dst;
    uint32_t* zero_area;       // … and the launch clears these words (the tile-list builder's totals)
    uint32_t zero_words;
    uint32_t* fault_word;      // a list longer than the launch sorts whose keys are NOT all equal ORs GGR_FAULT_BUCKET in here
};
// tile_sort.hip, for the depth sort's bucket form: the buckets of more than `min_len` keys (what an overfull fine bin made longer
// than the regular class sorts) of `segments` x GGR_SORT_MAX_BINS ranges — up to GGR_TSORT_CAP_LARGE sorted, beyond copied if all
// keys are equal, else the fault bit.  A SMALL grid (64 workgroups per segment, each looking at 16 ranges): the class that sorts
// 8192 keys takes a CU's registers, and one such workgroup per bucket — all but a few with nothing to do — cost 5 to 60 µs
// depending on what else was resident.
void launch_bucket_sort_big(uint32_t segments, const uint2* ranges, uint32_t* point_list, const uint2* pair_list, uint32_t min_len,
                            hipStream_t s, TileSortExtras ex);
// tile_sort.hip: stable sort of every tile's list by the Gaussians' depth keys — lists with min_len < length <= max_len
// (others are left alone; max_len <= GGR_TSORT_CAP_LARGE)
void launch_tile_depth_sort(size_t T, const uint2* ranges, uint32_t* point_list, const uint2* pair_list, uint32_t min_len,
                            uint32_t max_len, hipStream_t s, int copy_longer /*1: lists longer than max_len are copied out
                            unsorted (the last launch of a forward: the blend must find valid ids)*/,
                            uint32_t* lsd_entries = nullptr /*+= the entries of the lists that took the kernel's slow route
                            (depths clustered in few buckets; tile_sort.h)*/,
                            const TileSortExtras* extras = nullptr);


void launch_blend_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const float4* splat,
                      const float4* colour,
                      const float* bg, float* out_color, float* final_T, uint32_t* n_contrib,
                      float* out_depth, float* ckpt /*or null*/, int ckpt_slots, uint32_t* tile_top, int views,
                      int scissored /*the lists are confined to a window of the frame (GgrSettings.scissor)*/,
                      void* zero_area /*or null: also cleared, on the side*/, size_t zero_bytes /*multiple of 16*/,
                      hipStream_t s);

// dev counters of the two blend kernels (zero unless the library was built with -DGGR_DEV_COUNTERS)
void blend_fwd_counters(unsigned long long* out4, int reset);
void blend_bwd_counters(unsigned long long* out4, int reset);

void launch_blend_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const float4* splat,
                      const float4* colour,
                      const float* bg, const float* final_T, const uint32_t* n_contrib,
                      const float* dL_dpix, const float* dL_ddepth /*or null*/, float* grad2d /*[P][16], zeroed*/,
                      const uint32_t* tile_top, const float* ckpt /*or null*/, int ckpt_slots, int segments, int views,
                      hipStream_t s);

// radii / clamped / grad2d / dL_dmeans2D / dL_daux are per view ([V,P,…]); the other gradients are summed over the
// views; pose_acc holds V·blocks rows of 64 floats; dL_dview/dL_dproj [V,16], dL_dcampos [V,3]
void launch_preprocess_bwd(int P, int D, int M, const float* means3D, const float* shs /*only says: SH path*/,
                           const float4* sh_jac /*the forward's ∂colour/∂direction (GeomLayout.sh_jac)*/,
                           int has_colors_precomp, const float* scales, const float* rotations,
                           float scale_modifier, const float* cov3D, ViewSet vs, int W, int H, const int32_t* radii,
                           const uint32_t* clamped,
                           const float* grad2d, int has_dz, float* dL_dmeans3D, float* dL_dmeans2D,
                           float* dL_dopacity, float* dL_dsh,
                           float* dL_dcolors_precomp, float* dL_dcov3D, float* dL_dscales,
                           float* dL_drotations, float* dL_daux, float* pose_acc /*null: no camera gradient*/,
                           float* dL_dview, float* dL_dproj, float* dL_dcampos, InputForm inf,
                           int cov_is_input /*cov3D is the caller's tensor (in its form), not the stored one*/,
                           hipStream_t s);

void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                         hipStream_t s);

// camera.hip: per-view view / full projection matrices, camera position, tan(fov/2), 1/near — one launch
void launch_camera_setup(int n, const float* extrinsics, const float* intrinsics, const float* near, const float* far,
                         int scale_invariant, float* view, float* full, float* campos, float* tanfov, float* scale,
                         hipStream_t s);

// util.hip: float4 streaming copy of `bytes` (multiple of 16) — the measured HBM ceiling bench.py quotes
void launch_copy_f4(const void* src, void* dst, size_t bytes, int blocks /*0 = default*/, hipStream_t s);
// util.hip: a one-wave kernel that occupies its stream for `ticks` of the 100 MHz wall clock / an empty kernel (side-stream probe)
void launch_spin(unsigned long long ticks, uint32_t* started /*host-visible, or NULL*/, hipStream_t s);
void launch_noop(hipStream_t s);

void launch_unpack_geom(GeomLayout g, int P, float* depth, float* xy, float* conic_opacity, float* rgb,
                        int32_t* tiles_touched, uint8_t* clamped, hipStream_t s);

}  // namespace ggr
