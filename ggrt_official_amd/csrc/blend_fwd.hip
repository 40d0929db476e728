// blend_fwd.hip — per-tile front-to-back alpha compositing for gfx950.
//
// Replaces the forward `renderCUDA` stage of the rasterizer behind reference
// cuda_splatting.py:114-125 with identical skip/stop rules.
//
// Mapping: one 256-thread workgroup per 16×16 tile = 4 wave64; wave w owns the 8×8 quadrant
// (w&1, w>>1) of the tile so that a wave's pixels are spatially compact (tight early-out and
// tight per-wave culling).  The tile's list is staged through LDS in batches of 256 entries
// (one 48-B splat record gathered per thread); inside a batch every wave first culls the entries
// against its own 8×8 quadrant (lane-per-entry conservative test + ballot) and then walks only
// the surviving entries, reading the record as LDS broadcasts.  A wave that is done (all 64
// pixels saturated) stops walking; the workgroup stops fetching when all four waves are done.
//
// The quadrant cull is exact with respect to the reference semantics: an entry is dropped for a
// quadrant only if α < 1/255 (or power > 0 is impossible) for every pixel of the quadrant, i.e.
// only entries the per-pixel rule would `continue` past anyway; their list positions are still
// counted, so n_contrib is unchanged.
#include "blend_common.h"

namespace ggr {

// dev counters (-DGGR_DEV_COUNTERS builds only; ggr_debug_counters): [0] survivors listed by the quadrant cull, [1] survivors
// walked (a wave leaves the walk when its pixels are saturated), [2] (survivor, lane) pairs that composited, [3] batches culled
__device__ unsigned long long g_fwd_counters[4];
void blend_fwd_counters(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_counters), sizeof(g_fwd_counters));
    if (reset) { const unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_counters), z, sizeof z); }
}

#define BATCH GGR_BATCH
#ifndef SURV_GROUP
#define SURV_GROUP 4   // survivors per trip of the blend loop (one broadcast read of their offsets; measured, 2 / 4 / 8:
                       // C3 150 / 150 / 176 µs, C5′ 195 / 173 / 173 — at 8 the compiler keeps all eight records live)
#endif

// TRAIN = false (GgrForwardOut.no_backward: inference, torch.no_grad()): nothing is kept for a backward — no last contributor
// per pixel (one select per survivor in a loop bound by vector issue: 150.6 → 142.0 µs at C3), no final T; tile_top = 0.
// [budget: prologue]  (scripts/valu_budget.py)
template <bool TRAIN>
__global__ void __launch_bounds__(256)
blend_fwd_kernel(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ splat,
                 const float4* __restrict__ colour,
                 const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ final_T,
                 uint32_t* __restrict__ n_contrib, float* __restrict__ out_depth, float* __restrict__ ckpt,
                 int ckpt_slots, uint32_t* __restrict__ tile_top, int views, int interleaved, float4* __restrict__ zero4,
                 size_t zero4_n) {
    __shared__ StagedSplat stage[BATCH + 1];   // + the null record that pads a wave's survivor list
    typedef uint32_t surv_t;   // (u32, not u16: four offsets are one 16-B broadcast read and need no unpacking — one vector
                               //  instruction per survivor less in a loop bound by vector issue: 152.0 → 150.2 µs at C3)
    __shared__ __attribute__((aligned(16))) surv_t surv[4][BATCH + SURV_GROUP];  // per wave: LDS byte offsets of its survivors
    __shared__ int wave_done[4];
    __shared__ uint32_t wave_last[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // On the side: clear the backward's per-Gaussian gradient records (GgrForwardOut.backward_scratch).  This kernel is
    // VALU-bound with an idle memory pipe — two 16-B stores per thread at C3 cost nothing here and save the backward a
    // 65 MB memset (0.014 ms).
    for (size_t z = (size_t)blockIdx.x * 256 + tid; z < zero4_n; z += (size_t)gridDim.x * 256)
        zero4[z] = make_float4(0.f, 0.f, 0.f, 0.f);
    // `views` frames stacked vertically (ggr_common.h ViewSet): tile vt of the launch = tile (vt mod T) of view vt / T;
    // per-view outputs and per-pixel state follow each other in the caller's [V, …] arrays
    const int tiles1 = grid_x * ((H + GGR_TILE - 1) / GGR_TILE), ntiles = tiles1 * views;
    const int vtile = xcd_tile((int)blockIdx.x, ntiles, interleaved != 0);
    if (vtile < 0) return;  // padding workgroup (before any barrier)
    const int view = vtile / tiles1, tile = vtile - view * tiles1;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int qx0 = tile_x * GGR_TILE + (wave & 1) * 8, qy0 = tile_y * GGR_TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    // quadrant rect clipped to the image (pixels outside never contribute)
    const float rx0 = (float)qx0, ry0 = (float)qy0;
    const float rx1 = (float)min(qx0 + 7, W - 1), ry1 = (float)min(qy0 + 7, H - 1);
    const bool quad_live = qx0 < W && qy0 < H;

    const uint2 range = ranges[vtile];
    const int total = (int)(range.y - range.x);
    // images with few tiles: per-pixel checkpoints every `ck_every` list positions let the backward replay the
    // list in independent depth segments (ggr_common.h, ImageLayout).  A pixel that is done keeps its final
    // state, which is all the backward can ask of it (it never looks behind a pixel's last contributor).
    const size_t hw = (size_t)H * W;
    const size_t pid = inside ? (size_t)py * W + px : 0;
    const int ck_every = ckpt ? ckpt_stride(total, ckpt_slots, ntiles) : 0;
    {   // this view's slices
        const size_t vo = (size_t)view * hw;
        out_color += 3 * vo; n_contrib += vo; bg += 3 * view;
        if (TRAIN) final_T += vo;   // (null without TRAIN)
        if (out_depth) out_depth += vo;
        if (ckpt) ckpt += (size_t)ckpt_slots * GGR_CKPT_FLOATS * vo;
    }

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f;
    uint32_t last = 0;
#ifdef GGR_DEV_COUNTERS
    unsigned long long dc_listed = 0, dc_walked = 0, dc_taken = 0, dc_batches = 0;
#endif
    // false: the pixel is saturated (or outside the image) and takes no further entry.  A lane MASK in scalar
    // registers, like every per-pixel condition below: a vector compare costs 4.6 cycles, a select 2.4, a literal operand
    // 2 more (tools/valu_peak_bench.hip, round 3) — the conditions are combined with scalar ANDs and applied by ONE select
    // on the weight and one on the position (as float factors and per-condition selects: 57 instead of 42 issue cycles)
    bool live = inside;
    float amax = GGR_ALPHA_MAX;
    __asm__ volatile("" : "+s"(amax));   // (a scalar register operand instead of a 32-bit literal in every v_min)
    if (tid == 0) {                      // the null record: opacity 0 → α = 0 → never contributes
        stage[BATCH].a = make_float4(0.f, 0.f, 0.f, 0.f); stage[BATCH].b = make_float4(0.f, 0.f, 0.f, 0.f);
        stage[BATCH].c = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (lane == 0) wave_done[wave] = quad_live ? 0 : 1;
    bool wdone = !quad_live;

    // [budget: stage]
    // the list ids of a batch are requested one batch ahead: id → record is a chain of two global round trips,
    // only the second is left on the batch's critical path
    uint32_t g_next = tid < total ? point_list[range.x + tid] : 0u;
    for (int b0 = 0; b0 < total; b0 += BATCH) {
        __syncthreads();  // previous batch fully consumed; wave_done visible
        if (wave_done[0] & wave_done[1] & wave_done[2] & wave_done[3]) break;
        const int nb = min(BATCH, total - b0);
        const uint32_t g = g_next;
        if (b0 + BATCH + tid < total) g_next = point_list[range.x + b0 + BATCH + tid];
        if (tid < nb) {
            // the entry's 32-B geometry record and 16-B colour record (ggr_common.h), into the staged layout {x, y, cxx, cxy |
            // cyy, opacity, r, g | b, z, qmax, –}
            float4 a = splat[2 * (size_t)g];
            const float4 ge = splat[2 * (size_t)g + 1], co = colour[g];
            float4 b = make_float4(ge.x, ge.y, co.x, co.y), c = make_float4(co.z, ge.z, ge.w, 0.f);
            stage_scale_conic(a, b, c);  // (blend_common.h: the pixel loop works on k·q, k = log2(e)/2)
            stage[tid].a = a;
            stage[tid].b = b;
            stage[tid].c = c;
        }
        __syncthreads();
        if (ckpt && b0 > 0 && b0 % ck_every == 0 && !wdone && inside) {
            float* ck = ckpt + (size_t)(b0 / ck_every) * GGR_CKPT_FLOATS * hw + pid;
            ck[0] = T; ck[hw] = C0; ck[2 * hw] = C1; ck[3 * hw] = C2; ck[4 * hw] = Dz;
        }
        if (!wdone) {
            // [budget: cull]
            // ---- cull: this wave's survivors of the whole batch, compacted into a wave-private list of LDS byte
            // offsets.  Round 2 walked the ballot mask with s_ff1 / s_and per survivor and
            // combined its per-pixel conditions in scalar mask registers: ≈ 20 SALU per ≈ 22 VALU instructions, and a
            // CU has ONE scalar unit for its four SIMDs — the kernel ran at the scalar unit's pace (85 M SALU against
            // 105 M VALU per launch at C3).  Now the WALK costs no scalar instruction: a survivor's offset arrives in a
            // VGPR (a broadcast LDS read).  The per-pixel conditions are lane masks in scalar registers again (three
            // scalar ANDs per survivor, 38 M SALU per launch): as float factors and one select per condition they cost
            // more vector issue cycles than they saved scalar ones (see `live` above).
            int ns = 0;
            surv_t* my_surv = surv[wave];
            float bx0 = rx0, by0 = ry0, bx1 = rx1, by1 = ry1;   // the pixels that are not saturated yet
            {
                const uint64_t act = __ballot(live);
                if (act) active_box(act, rx0, ry0, bx0, by0, bx1, by1);
            }
            for (int s0 = 0; s0 < nb; s0 += 64) {
                const int e = s0 + lane;
                bool keep = false;
                if (e < nb) {
                    const float4 a = stage[e].a;
                    const float4 b = stage[e].b;
                    keep = staged_box_may_contribute(a, b, stage[e].c.z, bx0, by0, bx1, by1);
                }
                const uint64_t mk = __ballot(keep);
                if (keep) my_surv[ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = (surv_t)(e * 48);
                ns += __popcll(mk);
            }
#ifdef GGR_DEV_COUNTERS
            dc_listed += (unsigned long long)ns; dc_batches++;
#endif
            if (lane < SURV_GROUP) my_surv[ns + lane] = (surv_t)(BATCH * 48);  // pad with the null record (opacity 0)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // [budget: evaluate]
            const char* stage_bytes = reinterpret_cast<const char*>(stage);
            uint32_t hit_off = 0xFFFFFFFFu;   // LDS offset of the pixel's latest contributor in this batch (none yet)
            for (int k0 = 0; k0 < ns; k0 += SURV_GROUP) {
                uint32_t pkw[SURV_GROUP];       // the group's offsets, one broadcast read
                __builtin_memcpy(pkw, my_surv + k0, sizeof pkw);
#pragma unroll
                for (int u = 0; u < SURV_GROUP; u++) {
                    const uint32_t off = pkw[u];   // (VGPR, uniform)
                    const StagedSplat* rec = reinterpret_cast<const StagedSplat*>(stage_bytes + off);
                    const float4 a = rec->a, rb = rec->b;
                    const float2 rc = *reinterpret_cast<const float2*>(&rec->c);   // (blue, z)
                    const float q2 = staged_q2(a, rb, a.x - pixx, a.y - pixy);  // = −power·log2(e)
                    const float alpha = fminf(amax, rb.y * __builtin_amdgcn_exp2f(-q2));
                    // skip: power > 0, α < 1/255, or the pixel is saturated
                    const bool cand = live & (q2 >= 0.0f) & (alpha >= GGR_ALPHA_MIN);
                    const float wr = alpha * T;
                    const float test_T = T - wr;               // T·(1−α)
                    const bool stop = cand & (test_T < GGR_T_MIN);
                    const bool take = cand & !stop;
                    live = live & !stop;
                    const float w = take ? wr : 0.f;
                    C0 = fmaf(rb.z, w, C0); C1 = fmaf(rb.w, w, C1); C2 = fmaf(rc.x, w, C2); Dz = fmaf(rc.y, w, Dz);
                    T -= w;
                    // (which entry: its LDS offset, already in a register — the list position follows from it after the
                    //  batch; reading it from the record was a fourth LDS read per survivor, 2 of 12 LDS cycles)
                    if (TRAIN) hit_off = take ? off : hit_off;
#ifdef GGR_DEV_COUNTERS
                    if (k0 + u < ns) { dc_walked++; dc_taken += (unsigned long long)__popcll(__ballot(take)); }
#endif
                }
                if (!__any(live)) { wdone = true; break; }
            }
            // list position + 1 of entry e = offset / 48 of this batch: what n_contrib records
            if (TRAIN && hit_off != 0xFFFFFFFFu) last = (uint32_t)b0 + 1u + (((hit_off >> 4) * 0xAAABu) >> 17);
            if (wdone && lane == 0) wave_done[wave] = 1;
        }
    }
#ifdef GGR_DEV_COUNTERS
    if (lane == 0) {
        atomicAdd(&g_fwd_counters[0], dc_listed); atomicAdd(&g_fwd_counters[1], dc_walked);
        atomicAdd(&g_fwd_counters[2], dc_taken); atomicAdd(&g_fwd_counters[3], dc_batches);
    }
#endif
    // [budget: epilogue]
    // the tile's last contributor: the backward replays the list entries before it (and nothing else)
    if (TRAIN) {
        uint32_t wl = last;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, off));
        if (lane == 0) wave_last[wave] = wl;
        __syncthreads();
        if (tid == 0) tile_top[vtile] = max(max(wave_last[0], wave_last[1]), max(wave_last[2], wave_last[3]));
    } else if (tid == 0) {
        // nothing was kept: a ggr_backward handed this image_buffer by mistake replays NO list entry (all-zero blend
        // gradients) instead of walking lists with an uninitialised final T; one 4-B store per tile
        tile_top[vtile] = 0u;
    }
    if (inside) {
        if (TRAIN) {
            if (ckpt) {  // slot 0: the final sums (what lies behind a checkpoint = final − checkpoint)
                float* ck = ckpt + pid;
                ck[hw] = C0; ck[2 * hw] = C1; ck[3 * hw] = C2; ck[4 * hw] = Dz;
            }
            final_T[pid] = T;
            n_contrib[pid] = last;
        }
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        if (out_depth) out_depth[pid] = Dz;
    }
}

void launch_blend_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const float4* splat,
                      const float4* colour,
                      const float* bg, float* out_color, float* final_T, uint32_t* n_contrib,
                      float* out_depth, float* ckpt, int ckpt_slots, uint32_t* tile_top, int views, int scissored,
                      void* zero_area, size_t zero_bytes, hipStream_t s) {
    const int gx = (W + GGR_TILE - 1) / GGR_TILE, gy = (H + GGR_TILE - 1) / GGR_TILE;
    if (gx * gy * views == 0) return;
    // (final_T == nullptr: the caller keeps nothing for a backward)
#define GGR_LAUNCH_BFWD(TRAIN_)                                                                                              \
    hipLaunchKernelGGL(blend_fwd_kernel<TRAIN_>, dim3(xcd_grid(gx * gy * views)), dim3(256), 0, s, W, H, gx, ranges, point_list,  \
                       splat, colour, bg, out_color, final_T, n_contrib, out_depth, ckpt, ckpt_slots, tile_top, views,               \
                       xcd_forward_interleaved(gx * gy * views, scissored != 0) ? 1 : 0, (float4*)zero_area,                 \
                       zero_area ? zero_bytes / 16 : 0)
    if (final_T) GGR_LAUNCH_BFWD(true);
    else GGR_LAUNCH_BFWD(false);
#undef GGR_LAUNCH_BFWD
}

}  // namespace ggr
