// blend_bwd.hip — back-to-front replay of the per-tile compositing, gradients w.r.t. the 2D splat
// parameters, for gfx950.
//
// Replaces the backward `renderCUDA` stage of the rasterizer behind reference
// cuda_splatting.py:114-125 / train_ggrt_stable.py:143.
//
// The upstream kernel issues ~9 atomicAdds per (pixel, Gaussian) pair — 64 lanes hitting the
// same address.  Here each wave64 (an 8×8 pixel quadrant of the tile) reduces its 64 per-pixel
// partials with DPP row shifts / row broadcasts (no LDS traffic) and issues ONE set of atomics per
// (wave, Gaussian); entries that cannot reach α ≥ 1/255 anywhere in the quadrant, or that lie
// behind every pixel's last contributor, are skipped wave-wide before any per-pixel work.
#include "ggr_common.h"

namespace ggr {

#define BATCH 256

struct __attribute__((aligned(16))) StagedSplatB {
    float4 a;  // x, y, conic.xx, conic.xy
    float4 b;  // conic.yy, opacity, r, g
    float2 c;  // b, z
    uint32_t id;
    uint32_t pad;
};

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_get<0x111>(v);              // row_shr:1
    v += dpp_get<0x112>(v);              // row_shr:2
    v += dpp_get<0x114>(v);              // row_shr:4
    v += dpp_get<0x118>(v);              // row_shr:8   → lane 15 of every row holds the row sum
    v += dpp_get<0x142, 0xa>(v);         // row_bcast:15 into rows 1 and 3
    v += dpp_get<0x143, 0xc>(v);         // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ bool quad_may_contribute_b(float mx, float my, float cxx, float cxy, float cyy,
                                                      float opacity, float x0, float y0, float x1, float y1) {
    const float qx = fminf(fmaxf(mx, x0), x1), qy = fminf(fmaxf(my, y0), y1);
    const float dx = mx - qx, dy = my - qy;
    const float d2 = dx * dx + dy * dy;
    if (d2 == 0.f) return true;
    const float tr = cxx + cyy;
    const float det = cxx * cyy - cxy * cxy;
    const float lmin = det / tr;
    if (!(lmin > 0.f)) return true;
    const float bound = opacity * __expf(-0.5f * lmin * d2 * 0.999f);
    return bound >= GGR_ALPHA_MIN * 0.999f;
}

template <bool HAS_DEPTH>
__global__ void __launch_bounds__(256)
blend_bwd_kernel(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ splat,
                 const float* __restrict__ bg, const float* __restrict__ final_T,
                 const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                 const float* __restrict__ dL_ddepth, float* __restrict__ dL_dmean2D,
                 float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_drgb,
                 float* __restrict__ dL_dz) {
    __shared__ StagedSplatB stage[BATCH];
    __shared__ uint32_t wave_last_sh[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int qx0 = tile_x * GGR_TILE + (wave & 1) * 8, qy0 = tile_y * GGR_TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px, pixy = (float)py;
    const float rx0 = (float)qx0, ry0 = (float)qy0;
    const float rx1 = (float)min(qx0 + 7, W - 1), ry1 = (float)min(qy0 + 7, H - 1);

    const uint2 range = ranges[tile];
    const size_t hw = (size_t)H * W;
    const size_t pid = inside ? (size_t)py * W + px : 0;

    const float T_final = inside ? final_T[pid] : 0.f;
    const uint32_t last = inside ? n_contrib[pid] : 0u;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dpz = 0.f;
    if (inside) {
        dp0 = dL_dpix[pid]; dp1 = dL_dpix[hw + pid]; dp2 = dL_dpix[2 * hw + pid];
        if (HAS_DEPTH) dpz = dL_ddepth[pid];
    }
    const float bg_dot = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    // wave / block maxima of `last`: nothing behind them can matter
    uint32_t wl = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, off));
    if (lane == 0) wave_last_sh[wave] = wl;
    __syncthreads();
    const uint32_t block_last = max(max(wave_last_sh[0], wave_last_sh[1]), max(wave_last_sh[2], wave_last_sh[3]));
    const int top = (int)min(block_last, range.y - range.x);  // entries [0, top) are replayed

    float T = T_final;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accz = 0.f;
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, lcz = 0.f, last_alpha = 0.f;

    for (int hi = top; hi > 0; hi -= BATCH) {
        const int nb = min(BATCH, hi);
        __syncthreads();
        if (tid < nb) {
            const uint32_t g = point_list[range.x + hi - 1 - tid];
            const float4 a = splat[3 * (size_t)g], b = splat[3 * (size_t)g + 1], c = splat[3 * (size_t)g + 2];
            stage[tid].a = a;
            stage[tid].b = b;
            stage[tid].c = make_float2(c.x, c.y);
            stage[tid].id = g;
        }
        __syncthreads();
        for (int s0 = 0; s0 < nb; s0 += 64) {
            const int e = s0 + lane;
            bool keep = false;
            if (e < nb && (uint32_t)(hi - 1 - e) < wl) {
                const float4 a = stage[e].a;
                const float4 b = stage[e].b;
                keep = quad_may_contribute_b(a.x, a.y, a.z, a.w, b.x, b.y, rx0, ry0, rx1, ry1);
            }
            uint64_t m = __ballot(keep);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const int e2 = s0 + j;
                const uint32_t idx = (uint32_t)(hi - 1 - e2);  // position in the tile list
                const float4 a = stage[e2].a;
                const float4 b = stage[e2].b;
                const float2 c = stage[e2].c;
                const float dx = a.x - pixx, dy = a.y - pixy;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(GGR_ALPHA_MAX, b.y * G);
                const bool valid = idx < last && power <= 0.0f && alpha >= GGR_ALPHA_MIN;
                if (__ballot(valid) == 0ull) continue;

                float g_r = 0.f, g_g = 0.f, g_b = 0.f, g_z = 0.f, g_mx = 0.f, g_my = 0.f, g_cxx = 0.f, g_cxy = 0.f,
                      g_cyy = 0.f, g_op = 0.f;
                if (valid) {
                    const float inv = __frcp_rn(1.f - alpha);
                    T = T * inv;
                    const float w = alpha * T;
                    acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
                    acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
                    acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
                    lc0 = b.z; lc1 = b.w; lc2 = c.x;
                    float dL_dalpha = (b.z - acc0) * dp0 + (b.w - acc1) * dp1 + (c.x - acc2) * dp2;
                    g_r = w * dp0; g_g = w * dp1; g_b = w * dp2;
                    if (HAS_DEPTH) {
                        accz = last_alpha * lcz + (1.f - last_alpha) * accz;
                        lcz = c.y;
                        dL_dalpha += (c.y - accz) * dpz;
                        g_z = w * dpz;
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final * inv) * bg_dot;
                    const float dL_dG = b.y * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * a.z - gdy * a.w;
                    const float dG_ddely = -gdy * b.x - gdx * a.w;
                    g_mx = dL_dG * dG_ddelx * ddelx_dx;
                    g_my = dL_dG * dG_ddely * ddely_dy;
                    g_cxx = -0.5f * gdx * dx * dL_dG;
                    g_cxy = -0.5f * gdx * dy * dL_dG;
                    g_cyy = -0.5f * gdy * dy * dL_dG;
                    g_op = G * dL_dalpha;
                }
                g_r = wave_sum_to_lane63(g_r);
                g_g = wave_sum_to_lane63(g_g);
                g_b = wave_sum_to_lane63(g_b);
                g_mx = wave_sum_to_lane63(g_mx);
                g_my = wave_sum_to_lane63(g_my);
                g_cxx = wave_sum_to_lane63(g_cxx);
                g_cxy = wave_sum_to_lane63(g_cxy);
                g_cyy = wave_sum_to_lane63(g_cyy);
                g_op = wave_sum_to_lane63(g_op);
                if (HAS_DEPTH) g_z = wave_sum_to_lane63(g_z);
                if (lane == 63) {
                    const size_t g = stage[e2].id;
                    atomicAdd(&dL_drgb[3 * g], g_r);
                    atomicAdd(&dL_drgb[3 * g + 1], g_g);
                    atomicAdd(&dL_drgb[3 * g + 2], g_b);
                    atomicAdd(&dL_dmean2D[3 * g], g_mx);
                    atomicAdd(&dL_dmean2D[3 * g + 1], g_my);
                    atomicAdd(&dL_dconic[3 * g], g_cxx);
                    atomicAdd(&dL_dconic[3 * g + 1], g_cxy);
                    atomicAdd(&dL_dconic[3 * g + 2], g_cyy);
                    atomicAdd(&dL_dopacity[g], g_op);
                    if (HAS_DEPTH) atomicAdd(&dL_dz[g], g_z);
                }
            }
        }
    }
}

void launch_blend_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const float4* splat,
                      const float* bg, const float* final_T, const uint32_t* n_contrib,
                      const float* dL_dpix, const float* dL_ddepth, float* dL_dmean2D, float* dL_dconic,
                      float* dL_dopacity, float* dL_drgb, float* dL_dz, hipStream_t s) {
    const int gx = (W + GGR_TILE - 1) / GGR_TILE, gy = (H + GGR_TILE - 1) / GGR_TILE;
    if (gx * gy == 0) return;
    if (dL_ddepth)
        hipLaunchKernelGGL(blend_bwd_kernel<true>, dim3(gx * gy), dim3(256), 0, s, W, H, gx, ranges, point_list,
                           splat, bg, final_T, n_contrib, dL_dpix, dL_ddepth, dL_dmean2D, dL_dconic,
                           dL_dopacity, dL_drgb, dL_dz);
    else
        hipLaunchKernelGGL(blend_bwd_kernel<false>, dim3(gx * gy), dim3(256), 0, s, W, H, gx, ranges, point_list,
                           splat, bg, final_T, n_contrib, dL_dpix, dL_ddepth, dL_dmean2D, dL_dconic,
                           dL_dopacity, dL_drgb, dL_dz);
}

}  // namespace ggr
