// blend_bwd.hip — back-to-front replay of the per-tile compositing, gradients w.r.t. the 2D splat
// parameters, for gfx950.
//
// Replaces the backward `renderCUDA` stage of the rasterizer behind reference
// cuda_splatting.py:114-125 / train_ggrt_stable.py:143.
//
// The upstream kernel issues ~9 atomicAdds per (pixel, Gaussian) pair — 64 lanes hitting the same
// address.  Here each wave64 owns an 8×8 pixel quadrant of the tile and
//   * drops, wave-wide, every list entry that cannot reach α ≥ 1/255 anywhere in its quadrant (exact
//     ellipse-vs-box test, lane-per-entry + ballot) or that lies behind every pixel's last contributor;
//   * processes the survivors in batches of 8: per (entry, pixel) only Σw·dL/dc (3), Σm, Σm·y′, Σm·y′² are formed
//     (moments about the quadrant centre, SEPARABLE in the lane's pixel coordinates); these six arrays of 8
//     values are reduced over the 8 pixel rows with a TRANSPOSING tree — v_permlane32_swap, v_permlane16_swap,
//     DPP row_ror:8 — the x-moments are products of the column sums with the lane's x′, and three DPP
//     butterflies over the 8 columns finish the nine (ten) totals of a slot in all 8 lanes of its group;
//   * commits a batch with two vector atomic instructions into ONE 64-byte record per Gaussian
//     (ggr_common.h GGR_G2D_*), every slot's values in ONE of the two — one line transaction per slot: the
//     device sustains only ≈ 20 G atomic line transactions/s (tools/atomic_line_bench.hip), and with the values
//     spread over four arrays (0.89 ms at C3), or with a second / third instruction per slot, the kernel was
//     bound by that, not by arithmetic.
// Tried and rejected (round 1, measured): taking the nine sums on the idle MATRIX pipe instead — two
// v_mfma_f32_16x16x4_f32 stages with polynomial pixel weights, layout pinned by tools/mfma_reduce_test.hip —
// is exact but slower (0.70 ms vs 0.56 ms): 6 dependent MFMAs per entry serialise the wave.
#include "blend_common.h"

namespace ggr {

#define BATCH GGR_BATCH
#define RB 8  // entries per reduction batch

// dev counters (-DGGR_DEV_COUNTERS builds only; ggr_debug_counters): [0] (quadrant, entry) slots that survive the cull, [1] slots
// in which NO lane is valid, [2] valid (slot, lane) pairs, [3] batches culled
__device__ unsigned long long g_bwd_counters[4];
void blend_bwd_counters(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_counters), sizeof(g_bwd_counters));
    if (reset) { const unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_counters), z, sizeof z); }
}

// [budget: reduce-dpp-moves]  (scripts/valu_budget.py attributes the ISA below each marker to that phase)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    // every lane has a source under these controls (rotations / permutations inside a row), so `old` is never
    // used: passing the value itself with bound_ctrl lets the compiler fold the move into the consuming
    // v_add_f32_dpp instead of materialising a zero and a v_mov_b32_dpp (12 of 33 per batch did not fold)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// [budget: reduce-rows]
// Transposing reduction of v[0..7] over the 8 pixel ROWS of the quadrant (lane = 8·row + column).  Returns,
// in lane l, the sum over the 8 lanes {column (l & 7) of every row} of v[l >> 3].
__device__ __forceinline__ float transpose_rows8(float (&v)[RB], int lane) {
    // level 32: lanes 0-31 keep slots 0-3, lanes 32-63 keep slots 4-7.  The exchange goes through the LDS crossbar
    // (ds_bpermute: no vector-issue slot, two selects + one add = 6 issue cycles) instead of v_permlane32_swap + add
    // (8 + 2) — the kernel is bound by vector issue, its five waves per SIMD hide the round trip: 364 → 347 µs at C3.
    // Measured beside it (NOTES r4): level 16 the same way 375 µs (a second DEPENDENT round trip), two or three of the
    // four pairs 349, all exchanges of three / six arrays requested before the first use 352 / 366, the whole row
    // reduction through a transposing LDS tile (8 stores + two 16-B loads per array) 359-368.
    float w4[4];
    const bool h32 = (lane & 32) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float keep = h32 ? v[i + 4] : v[i], send = h32 ? v[i] : v[i + 4];
        w4[i] = keep + __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(send)));
    }
    // level 16: even 16-lane rows keep the lower half of their slots, odd rows the upper half
    float w2[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w4[i]), __float_as_uint(w4[i + 2]), false, false);
        w2[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    // level 8: lanes with bit 3 clear keep slot 0 of the pair, the others slot 1 (row_ror:8 = lane ^ 8)
    const bool hi = (lane & 8) != 0;
    const float keep = hi ? w2[1] : w2[0];
    const float send = hi ? w2[0] : w2[1];
    return keep + dpp_mov<0x128>(send);
}

// [budget: reduce-columns]
// Butterfly sum over the 8 lanes of a group (the 8 pixel columns): every lane of the group gets the total.
__device__ __forceinline__ float sum_cols8(float s) {
    s += dpp_mov<0xB1>(s);   // quad_perm [1,0,3,2]
    s += dpp_mov<0x4E>(s);   // quad_perm [2,3,0,1]
    s += dpp_mov<0x141>(s);  // row_half_mirror
    // keep this add where it is: sunk into the commit branch it can no longer fold with its DPP move (a DPP read
    // cannot move under a narrower exec mask), which left 9 v_mov_b32_dpp + 9 v_add per batch instead of 9 v_add_dpp
    asm volatile("" : "+v"(s));
    return s;
}

// Transposing reduction of o[0..7] over the 8 lanes of a group (the 8 pixel columns): lane c of the group gets the
// group's sum of ONE of the values, o[π(c)] with π(c) = c for c < 4 and 11 − c for c ≥ 4 (the third level pairs lane c
// with lane 7 − c: row_half_mirror is the only DPP move that crosses the two quads of a group) — 7 DPP adds and 14
// selects for 8 sums; as 8 butterflies (sum_cols8: every lane gets every total) they are 24 DPP adds at twice a select's
// issue cost (tools/valu_peak_bench.hip).  A lane keeps the values whose index bits match its own (mirrored in the
// upper quad), sends the others to its partner and adds what the partner sends.
__device__ __forceinline__ float transpose_cols8(const float (&o)[8], int lane) {
    const bool up = (lane & 4) != 0;
    const bool e0 = (((lane ^ (lane >> 2)) & 1) != 0), e1 = ((((lane >> 1) ^ (lane >> 2)) & 1) != 0);
    float k1[4], k2[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float keep = e0 ? o[2 * i + 1] : o[2 * i], send = e0 ? o[2 * i] : o[2 * i + 1];
        k1[i] = keep + dpp_mov<0xB1>(send);   // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float keep = e1 ? k1[2 * j + 1] : k1[2 * j], send = e1 ? k1[2 * j] : k1[2 * j + 1];
        k2[j] = keep + dpp_mov<0x4E>(send);   // quad_perm [2,3,0,1]
    }
    const float keep = up ? k2[1] : k2[0], send = up ? k2[0] : k2[1];
    return keep + dpp_mov<0x141>(send);       // row_half_mirror
}

// [budget: prologue]
template <bool HAS_DEPTH>
__global__ void __launch_bounds__(256)
blend_bwd_kernel(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ splat,
                 const float4* __restrict__ colour,
                 const float* __restrict__ bg, const float* __restrict__ final_T,
                 const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                 const float* __restrict__ dL_ddepth, float* __restrict__ grad2d,
                 const uint32_t* __restrict__ tile_top, const float* __restrict__ ckpt, int ckpt_slots,
                 int segments, int views) {
    __shared__ StagedSplat stage[BATCH];
    __shared__ __attribute__((aligned(16))) uint16_t surv[4][BATCH];  // per wave: stage indices of the entries that survive its quadrant cull
    __shared__ uint32_t wave_top[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup = (depth segment, tile), segment-major: all tiles' segment 0 first.  (Tile-major order interleaves
    // the loaded and the empty segments with a fixed period, and the dispatcher's round robin then puts all loaded
    // workgroups on the same fraction of the CUs: C3 with 2 segments per tile ran 0.75 ms instead of 0.43.)
    // (`views` frames stacked vertically, as in blend_fwd: tile vt of the launch = tile (vt mod T) of view vt / T)
    const int tiles1 = grid_x * ((H + GGR_TILE - 1) / GGR_TILE), ntiles = tiles1 * views;
    const int seg = segments > 1 ? (int)blockIdx.x / xcd_grid(ntiles) : 0;
    const int vtile = xcd_tile((int)blockIdx.x - seg * xcd_grid(ntiles), ntiles, true);
    if (vtile < 0) return;  // padding workgroup (before any barrier)
    const int view = vtile / tiles1, tile = vtile - view * tiles1;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int qx0 = tile_x * GGR_TILE + (wave & 1) * 8, qy0 = tile_y * GGR_TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const float rx0 = (float)qx0, ry0 = (float)qy0;
    // quadrant centre and the lane's pixel coordinates relative to it (the moment reduction is separable in them)
    const float qcx = rx0 + 3.5f, qcy = ry0 + 3.5f;
    const float pxc = (float)(lane & 7) - 3.5f, pyc = (float)(lane >> 3) - 3.5f;
    const float rx1 = (float)min(qx0 + 7, W - 1), ry1 = (float)min(qy0 + 7, H - 1);

    const uint2 range = ranges[vtile];
    // entries [0, top) are replayed: nothing behind the tile's last contributor (left by the forward) can matter
    const int top = (int)min(tile_top[vtile], range.y - range.x);
    // ---- depth segment [seg_lo, seg_hi) of the replayed entries ---------------------------------------------------
    // One segment = the whole list unless the forward left checkpoints (ggr_common.h, ImageLayout).  Then segment
    // s is checkpoint interval s of this tile's list; the workgroups of intervals behind `top` leave here, having
    // read two words.
    int seg_lo = 0, seg_hi = top, stride = 0;
    if (segments > 1) {
        stride = ckpt_stride((int)(range.y - range.x), ckpt_slots, ntiles);
        seg_lo = seg * stride;
        seg_hi = min(top, seg_lo + stride);
    }
    if (seg_lo >= seg_hi) return;  // (block-uniform, before any barrier)
    const size_t hw = (size_t)H * W;
    const size_t pid = inside ? (size_t)py * W + px : 0;
    {   // this view's slices
        const size_t vo = (size_t)view * hw;
        final_T += vo; n_contrib += vo; dL_dpix += 3 * vo; bg += 3 * view;
        if (HAS_DEPTH) dL_ddepth += vo;
        if (ckpt) ckpt += (size_t)ckpt_slots * GGR_CKPT_FLOATS * vo;
    }

    const float T_final = inside ? final_T[pid] : 0.f;
    uint32_t last = inside ? n_contrib[pid] : 0u;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dpz = 0.f;
    if (inside) {
        dp0 = dL_dpix[pid]; dp1 = dL_dpix[hw + pid]; dp2 = dL_dpix[2 * hw + pid];
        if (HAS_DEPTH) dpz = dL_ddepth[pid];
    }
    // ---- zero-gradient window skip ------------------------------------------------------------------------------
    // A pixel whose upstream gradient is exactly zero contributes exactly zero to every sum below (each term is a
    // product with dL/dpixel): it is treated as a pixel without contributors.  The reference's fine-tune loop
    // back-propagates, per crop cell, a gradient that is zero outside the cell through a full-frame render
    // (finetune_ggrt_stable.py:126-142): a wave whose quadrant carries no gradient then has no survivors, and a
    // workgroup whose 256 pixels carry none — or whose live pixels all end before this depth segment — leaves below.
    if (dp0 == 0.f && dp1 == 0.f && dp2 == 0.f && dpz == 0.f) last = 0u;
    const float bg_dot = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    // wave maximum of `last`: nothing behind it can matter to this quadrant
    uint32_t wl = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, off));
    // … and the workgroup's: nothing behind it is staged at all (the forward's tile_top is the same maximum over
    // ALL pixels of the tile; with a dense gradient the two agree)
    if (lane == 0) wave_top[wave] = wl;
    __syncthreads();
    seg_hi = min(seg_hi, (int)max(max(wave_top[0], wave_top[1]), max(wave_top[2], wave_top[3])));
    if (seg_lo >= seg_hi) return;  // (block-uniform; no barrier is pending)

    float T = T_final;
    float R = T_final * bg_dot;  // everything behind the current entry, dotted with dL/dpixel (see the slot body)
    // (scalar register operands instead of 32-bit literals in every v_min / v_cmp of the slot body: a literal costs 2 more
    //  issue cycles per instruction — tools/valu_peak_bench.hip; scripts/valu_budget.py: 12 literal instructions per trip)
    float amax = GGR_ALPHA_MAX, amin = GGR_ALPHA_MIN;
    __asm__ volatile("" : "+s"(amax), "+s"(amin));

    // A pixel whose list goes on behind this segment starts from the forward's checkpoint there: T as the forward
    // had it, and R = (final sums − checkpoint sums)·dL/dpixel + T_final·(bg·dL/dpixel) — the same "everything
    // behind" that the single-segment replay accumulates entry by entry.
    if (stride > 0 && seg_hi == seg_lo + stride && last > (uint32_t)seg_hi) {
        const float* ck = ckpt + (size_t)(seg_hi / stride) * GGR_CKPT_FLOATS * hw + pid;
        const float* fin = ckpt + pid;
        T = ck[0];
        R += (fin[hw] - ck[hw]) * dp0 + (fin[2 * hw] - ck[2 * hw]) * dp1 + (fin[3 * hw] - ck[3 * hw]) * dp2;
        if (HAS_DEPTH) R += (fin[4 * hw] - ck[4 * hw]) * dpz;
    }

    // which of the 9 (10) values this lane commits: value index vi = lane & 7 for the first atomic
    // instruction; lanes with (lane & 7) == 0 also commit value 8 (opacity) and 9 (depth) afterwards
    const int vi = lane & 7;
    const int vix = vi < 4 ? vi : 11 - vi;   // the value transpose_cols8 leaves in this lane
    const int my_slot = lane >> 3;

#ifdef GGR_DEV_COUNTERS
    unsigned long long dc_slots = 0, dc_dead = 0, dc_pairs = 0, dc_batches = 0;
#endif
    // [budget: stage]
    // the list ids of a batch are requested one batch ahead (id → record is a chain of two global round trips)
    uint32_t g_next = tid < seg_hi - seg_lo ? point_list[range.x + seg_hi - 1 - tid] : 0u;
    for (int hi_ = seg_hi; hi_ > seg_lo; hi_ -= BATCH) {
        const int nb = min(BATCH, hi_ - seg_lo);
        __syncthreads();
        const uint32_t g = g_next;
        if (hi_ - BATCH - 1 - tid >= seg_lo) g_next = point_list[range.x + hi_ - BATCH - 1 - tid];
        if (tid < nb) {
            float4 a = splat[2 * (size_t)g];
            const float4 ge = splat[2 * (size_t)g + 1], co = colour[g];   // (as blend_fwd.hip)
            float4 b = make_float4(ge.x, ge.y, co.x, co.y), c = make_float4(co.z, ge.z, ge.w, 0.f);
            stage_scale_conic(a, b, c);  // (blend_common.h: the pixel loop works on k·q, k = log2(e)/2)
            c.w = __uint_as_float(g);
            stage[tid].a = a;
            stage[tid].b = b;
            stage[tid].c = c;
        }
        __syncthreads();
        // [budget: cull]
        // ---- cull: compact this wave's survivors of the whole 256-entry batch into a wave-private list,
        //      so that every reduction batch below is full (a reduction costs >1000 cycles whether 1 or
        //      8 of its slots are used)
        // (plain LDS accesses ordered by wavefront fences — a `volatile` pointer would turn every access
        //  into flat_load/flat_store + s_waitcnt vmcnt(0))
        int ns = 0;
        uint16_t* my_surv = surv[wave];
        // the pixels that take ANY entry of this batch: those whose last contributor lies in it or in front of it
        float bx0 = rx0, by0 = ry0, bx1 = rx1, by1 = ry1;
        {
            const uint64_t act = __ballot(last > (uint32_t)(hi_ - nb));
            if (act) active_box(act, rx0, ry0, bx0, by0, bx1, by1);
        }
        for (int s0 = 0; s0 < nb; s0 += 64) {
            const int e = s0 + lane;
            bool keep = false;
            if (e < nb && (uint32_t)(hi_ - 1 - e) < wl) {
                const float4 a = stage[e].a;
                const float4 b = stage[e].b;
                keep = staged_box_may_contribute(a, b, stage[e].c.z, bx0, by0, bx1, by1);
            }
            const uint64_t mk = __ballot(keep);
            if (keep) my_surv[ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = (uint16_t)e;
            ns += __popcll(mk);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef GGR_DEV_COUNTERS
        dc_slots += (unsigned long long)ns; dc_batches++;
#endif
        {
            for (int k0 = 0; k0 < ns; k0 += RB) {
                // [budget: slot-setup]
                const uint4 pk = *reinterpret_cast<const uint4*>(my_surv + k0);  // 8 × u16 indices, one broadcast read
                const uint32_t pkw[4] = {pk.x, pk.y, pk.z, pk.w};
                // ---- one reduction batch: up to RB surviving entries ---------------------------------
                // Per pixel only RAW MOMENTS are formed: with m = G·dL/dα (zero on skipped lanes)
                //   S0 = Σm, Sx = Σm·dx, Sy = Σm·dy, Sxx = Σm·dx², Sxy = Σm·dx·dy, Syy = Σm·dy²
                // (taken about the quadrant centre and shifted to the mean at commit, see below)
                // and the per-Gaussian algebra (× opacity, × conic, × W/2 …) is done ONCE per entry after the
                // wave reduction.  A skipped lane is treated as α = 0: T·1/(1-0), w = 0, so T and the
                // "everything behind" sum R pass through unchanged — exactly the reference's `continue`,
                // without per-value selects.
                float g_r[RB], g_g[RB], g_b[RB], g_m[RB], g_my[RB], g_myy[RB], g_z[RB];
                int my_e = -1;  // stage index of slot `my_slot` (-1 = empty slot)
                // slot → stage index, resolved with scalar ops first so that the 8 slot bodies below are
                // straight-line code: the compiler can batch their LDS broadcast reads and overlap slot
                // k+1's geometry with slot k's dependent T / colour-behind chain
                int e_sl[RB];
                bool ok_sl[RB];
#pragma unroll
                for (int sl = 0; sl < RB; sl++) {
                    ok_sl[sl] = k0 + sl < ns;
                    // empty slot: any staged entry, masked by ok_sl (the list word behind `ns` is stale)
                    e_sl[sl] = ok_sl[sl] ? __builtin_amdgcn_readfirstlane((int)((pkw[sl >> 1] >> (16 * (sl & 1))) & 0xffffu)) : 0;
                    if (ok_sl[sl] && my_slot == sl) my_e = e_sl[sl];
                }
                // [budget: evaluate]
#pragma unroll
                for (int sl = 0; sl < RB; sl++) {
                    {
                        const int e2 = e_sl[sl];
                        const uint32_t idx = (uint32_t)(hi_ - 1 - e2);  // position in the tile list
                        const float4 a = stage[e2].a;
                        const float4 b = stage[e2].b;
                        const float4 c = stage[e2].c;
                        const float q2 = staged_q2(a, b, a.x - pixx, a.y - pixy);  // = −power·log2(e)
                        const float G = __builtin_amdgcn_exp2f(-q2);
                        const float alpha_raw = fminf(amax, b.y * G);
                        const bool valid = ok_sl[sl] && idx < last && q2 >= 0.0f && alpha_raw >= amin;
#ifdef GGR_DEV_COUNTERS
                        if (ok_sl[sl]) { const uint64_t vm = __ballot(valid); dc_dead += vm ? 0ull : 1ull; dc_pairs += (unsigned long long)__popcll(vm); }
#endif
                        const float alpha = valid ? alpha_raw : 0.f;
                        const float inv = __builtin_amdgcn_rcpf(1.f - alpha);  // v_rcp_f32 (1 ulp); __frcp_rn would expand to a 10-instruction IEEE division
                        T = T * inv;
                        const float w = alpha * T;
                        // dL/dα_s = T_s·(c_s·dp) − R_s/(1−α_s),  R_s = Σ_{s' behind s} w_s'·(c_s'·dp) + T_final·(bg·dp):
                        // the reference's per-channel "colour behind" recurrence collapsed to ONE scalar per
                        // pixel (T_s·acc_c = R-part/(1−α_s) summed over channels) — 8 VALU ops instead of 15
                        float cdp = b.z * dp0 + b.w * dp1 + c.x * dp2;
                        g_r[sl] = w * dp0; g_g[sl] = w * dp1; g_b[sl] = w * dp2;
                        if (HAS_DEPTH) {
                            cdp += c.y * dpz;
                            g_z[sl] = w * dpz;
                        }
                        const float dL_dalpha = T * cdp - R * inv;
                        R += w * cdp;
                        const float mm = valid ? G * dL_dalpha : 0.f;
                        // moments about the QUADRANT CENTRE in the lane's own (constant) pixel coordinates: only the
                        // y factors are formed per (entry, pixel); the x factors follow after the reduction over y
                        const float my_ = mm * pyc;
                        g_m[sl] = mm; g_my[sl] = my_; g_myy[sl] = my_ * pyc;
                    }
                }
                // [budget: reduce-calls]
                // ---- reductions.  lane = 8·y + x, so the three transposing levels (32, 16, 8) sum over the pixel
                // ROWS and leave, in lane (slot, x), column x's partial sums of that slot: six arrays go through
                // them (Σw·dp_{r,g,b}, Σm, Σm·y', Σm·y'²) instead of nine — the moments in x are products of the
                // column sums with the lane's x' (once per batch, not per entry) — then ONE transposing reduction over
                // the 8 columns finishes the eight committed values (see the commit below; two butterflies give the
                // opacity / depth pair to every lane)
                const float Cm = transpose_rows8(g_m, lane), Cy = transpose_rows8(g_my, lane);
                const float Cyy = transpose_rows8(g_myy, lane);
                float o[8];
                o[0] = transpose_rows8(g_r, lane); o[1] = transpose_rows8(g_g, lane); o[2] = transpose_rows8(g_b, lane);
                const float S0 = sum_cols8(Cm);   // (every lane: the pair value of the commit)
                float t_z = 0.f;
                if (HAS_DEPTH) t_z = sum_cols8(transpose_rows8(g_z, lane));
                // [budget: commit]
                // ---- commit: lane (slot, c) finishes value π(c) of its slot (transpose_cols8) -------------------
                // The five geometric values are LINEAR in the six moment sums, with coefficients of the slot's entry:
                // every lane forms them on its column's partial sums, and one transposing reduction over the columns
                // finishes all eight values (the three colour sums ride along) — instead of nine butterflies that leave
                // all nine totals in every lane and a select of the lane's own.
                // A slot has 9 (10) values but only 8 lanes.  Sending values 8, 9 in a second instruction of their
                // own made two 64-B line transactions per slot; instead the 8-lane groups of an even and an odd
                // slot (the two halves of a 16-lane row) help each other: instruction A carries the even slots'
                // values 0–7 from their own lanes and their values 8, 9 from lanes 0, 1 of the odd neighbour
                // group (fetched with one row_ror:8 move each), instruction B the odd slots' — every slot's values
                // leave in ONE instruction, one line transaction per slot, still two atomic instructions per batch.
                float val = 0.f;
                uint32_t gid = 0u;
                {
                    const int e = max(my_e, 0);
                    const float4 a = stage[e].a;
                    const float4 b = stage[e].b;
                    gid = __float_as_uint(stage[e].c.w);
                    const float op = b.y;
                    // shift the moments from the quadrant centre to the Gaussian's mean: d = mean − pixel = o − p',
                    // o = mean − quadrant centre, p' the centred pixel coordinates
                    const float ox = a.x - qcx, oy = a.y - qcy;
                    const float Mx = pxc * Cm, Mxx = pxc * Mx, Mxy = pxc * Cy;   // (this column's share of the moments in x)
                    const float Sx = ox * Cm - Mx, Sy = oy * Cm - Cy;
                    // (the staged conic is k·cxx, 2k·cxy, k·cyy: back to the plain one with 1/k)
                    o[3] = -op * GGR_INV_KQ * (a.z * Sx + 0.5f * a.w * Sy) * ddelx_dx;
                    o[4] = -op * GGR_INV_KQ * (b.x * Sy + 0.5f * a.w * Sx) * ddely_dy;
                    o[5] = -0.5f * op * (ox * (Sx - Mx) + Mxx);
                    o[6] = -0.5f * op * (ox * Sy - oy * Mx + Mxy);
                    o[7] = -0.5f * op * (oy * (Sy - Cy) + Cyy);
                    val = transpose_cols8(o, lane);
                }
                if (my_e < 0) val = 0.f;
                // (a third atomic instruction — a third line transaction per slot — once cost the depth variant 45 %:
                //  C3 0.44 → 0.645 ms; the atomics are cheap only while there are few line transactions per slot)
                static_assert(GGR_G2D_Z == GGR_G2D_OPACITY + 1, "opacity and depth are committed as one pair");
                constexpr int NPAIR = HAS_DEPTH ? 2 : 1;
                const float pair_own = my_e < 0 ? 0.f : ((HAS_DEPTH && vi == 1) ? t_z : S0);
                // the other half of the 16-lane row: its slot's pair value (same vi) and record
                const float pair_nb = dpp_mov<0x128>(pair_own);
                const uint32_t gid_nb = __float_as_uint(dpp_mov<0x128>(__uint_as_float(gid)));
                const bool odd = (my_slot & 1) != 0;
                float* const rec_own = grad2d + GGR_G2D_STRIDE * (size_t)gid;
                float* const rec_nb = grad2d + GGR_G2D_STRIDE * (size_t)gid_nb;
                const bool help = vi < NPAIR && pair_nb != 0.f;  // (an empty neighbour slot has pair value 0)
                {   // instruction A: the even slots' records
                    float* const p = odd ? rec_nb + GGR_G2D_OPACITY + vi : rec_own + vix;
                    const float v = odd ? pair_nb : val;
                    if (odd ? help : v != 0.f) atomicAdd(p, v);
                }
                {   // instruction B: the odd slots' records
                    float* const p = odd ? rec_own + vix : rec_nb + GGR_G2D_OPACITY + vi;
                    const float v = odd ? val : pair_nb;
                    if (odd ? v != 0.f : help) atomicAdd(p, v);
                }
            }
        }
    }
#ifdef GGR_DEV_COUNTERS
    if (lane == 0) {
        atomicAdd(&g_bwd_counters[0], dc_slots); atomicAdd(&g_bwd_counters[1], dc_dead);
        atomicAdd(&g_bwd_counters[2], dc_pairs); atomicAdd(&g_bwd_counters[3], dc_batches);
    }
#endif
}

void launch_blend_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const float4* splat,
                      const float4* colour,
                      const float* bg, const float* final_T, const uint32_t* n_contrib,
                      const float* dL_dpix, const float* dL_ddepth, float* grad2d, const uint32_t* tile_top,
                      const float* ckpt, int ckpt_slots, int segments, int views, hipStream_t s) {
    const int gx = (W + GGR_TILE - 1) / GGR_TILE, gy = (H + GGR_TILE - 1) / GGR_TILE;
    if (gx * gy * views == 0) return;
    const int nt = gx * gy * views;
    // one workgroup per (checkpoint interval, tile): `segments` == the forward's slot count, or 1 without checkpoints
    segments = (ckpt && ckpt_slots >= 2) ? ckpt_slots : 1;
    if (dL_ddepth)
        hipLaunchKernelGGL(blend_bwd_kernel<true>, dim3(xcd_grid(nt) * segments), dim3(256), 0, s, W, H, gx, ranges,
                           point_list, splat, colour, bg, final_T, n_contrib, dL_dpix, dL_ddepth, grad2d, tile_top, ckpt,
                           ckpt_slots, segments, views);
    else
        hipLaunchKernelGGL(blend_bwd_kernel<false>, dim3(xcd_grid(nt) * segments), dim3(256), 0, s, W, H, gx, ranges,
                           point_list, splat, colour, bg, final_T, n_contrib, dL_dpix, dL_ddepth, grad2d, tile_top, ckpt,
                           ckpt_slots, segments, views);
}

}  // namespace ggr
