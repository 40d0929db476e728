// preprocess.hip — per-Gaussian forward stage for gfx950.
//
// Replaces the `preprocessCUDA` stage of the third-party rasterizer that GGRt reaches through
// reference ggrt/model/pixelsplat/decoder/cuda_splatting.py:114-125.
// One lane per Gaussian; streaming, HBM-bound (algorithmic bytes: 12+24+4+12·K read, 48+20 written).
//
// Arithmetic follows the operation order of oracle/ggr_oracle.c with FMA contraction disabled, so
// the discrete outputs (cull decision, radius, tile rect → num_rendered) are bit-identical to the
// CPU restatement; gfx950 fp32 divide/sqrt are correctly rounded under hipcc's defaults.
#include "ggr_common.h"
#include "sh_stage.h"
#include "sh_terms.h"
#include <algorithm>

#pragma clang fp contract(off)

namespace ggr {

__device__ __constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};
__device__ __constant__ float kSH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                                           -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                                           0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f

__device__ __forceinline__ void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* cov6) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
    const float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float Mx[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Mx[3 * i + j] = R[3 * i + j] * sc[j];
    float S[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) a += Mx[3 * i + k] * Mx[3 * j + k];
            S[3 * i + j] = a;
        }
    cov6[0] = S[0]; cov6[1] = S[1]; cov6[2] = S[2]; cov6[3] = S[4]; cov6[4] = S[5]; cov6[5] = S[8];
}

// SH basis in the rasterizer's sign convention; B must hold 25 floats (band 4: see oracle/ggr_oracle.c header)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* B) {
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = kSH_C2[0] * xy; B[5] = kSH_C2[1] * yz; B[6] = kSH_C2[2] * (2.0f * zz - xx - yy);
            B[7] = kSH_C2[3] * xz; B[8] = kSH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = kSH_C3[0] * y * (3.0f * xx - yy);
                B[10] = kSH_C3[1] * xy * z;
                B[11] = kSH_C3[2] * y * (4.0f * zz - xx - yy);
                B[12] = kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = kSH_C3[4] * x * (4.0f * zz - xx - yy);
                B[14] = kSH_C3[5] * z * (xx - yy);
                B[15] = kSH_C3[6] * x * (xx - 3.0f * yy);
                if (deg > 3) {
                    B[16] = kSH_C4[0] * xy * (xx - yy);
                    B[17] = kSH_C4[1] * yz * (3.0f * xx - yy);
                    B[18] = kSH_C4[2] * xy * (7.0f * zz - 1.0f);
                    B[19] = kSH_C4[3] * yz * (7.0f * zz - 3.0f);
                    B[20] = kSH_C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
                    B[21] = kSH_C4[5] * xz * (7.0f * zz - 3.0f);
                    B[22] = kSH_C4[6] * (xx - yy) * (7.0f * zz - 1.0f);
                    B[23] = kSH_C4[7] * xz * (xx - 3.0f * yy);
                    B[24] = kSH_C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
                }
            }
        }
    }
}

// MULTI = false: one view, the loop over views folds away at compile time; MULTI = true: run-time loop over the views.
// KC = 0: the block's SH rows are staged whole (sh_stage.h).  KC = 16 / 25 (one view, degree 3 / 4): the rows go
// through LDS one THIRD at a time — KC floats per Gaussian: one colour channel of channel-major rows, floats
// [J·KC, (J+1)·KC) of k-major ones — with the next third's loads in flight while the current one is consumed.  A whole
// 75-float row per Gaussian is 76.8 KB of LDS per block = 2 blocks per CU, and the kernel then runs at 2.9 TB/s; a third
// is 25.6 KB (tools/sh_stage_bench.hip: the bare access pattern 0.105 → 0.082 ms at 1 M × 75 floats, 0.063 → 0.050 at
// 48; the rows' lines are fetched three times, L2 hits after the first).  Every channel still sums its coefficients in
// ascending k, so the colours stay bit-identical to the whole-row path.
//
// PART (GGR_PRE_*): ALL = the whole stage in one launch.  GEOMETRY = everything but the SH colour: projection, 2-D
// covariance, radius, tile rect, sort key, the 32-B geometry record — ≈ 90 B per Gaussian, the only part the depth
// sort and the tile lists wait for.  COLOUR = the SH evaluation alone (the rows: 192 / 300 B per Gaussian, 4/5 of the
// stage's bytes) for the Gaussians GEOMETRY found visible (radii > 0), into the 16-B colour record and the clamp bits —
// only the blend needs it, so the forward runs it on a side stream beside the latency-bound sort / tile-list kernels
// (api.hip forward_impl).  Same arithmetic in the same order in every PART: the outputs are bit-identical.
// JAC: also leave the Jacobian ∂colour/∂direction of every visible Gaussian (ggr_common.h sh_jac) for the backward — nine
// more sums over the coefficients the colour evaluation has in LDS anyway, and 48 B written per Gaussian, against the
// backward re-reading the whole SH row (192 / 300 B per Gaussian) for its view-direction term.
template <bool MULTI, int KC, int PART, bool JAC>
__global__ void __launch_bounds__(GGR_PRE_THREADS)
preprocess_fwd_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
                      const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      float scale_modifier, const float* __restrict__ cov3D_precomp,
                      const float* __restrict__ aux_precomp, ViewSet vs, int W, int H,
                      int32_t* __restrict__ radii, float4* __restrict__ splat, float4* __restrict__ colour,
                      float4* __restrict__ sh_jac, size_t jac_plane,
                      uint32_t* __restrict__ depth_key,
                      uint2* __restrict__ rect, uint32_t* __restrict__ clamped_out,
                      float* __restrict__ cov3D_out, uint32_t* __restrict__ zero_area, uint32_t zero_words,
                      uint32_t* __restrict__ zero_area2, uint32_t zero_words2,
                      uint32_t* __restrict__ block_max, uint32_t* __restrict__ block_min, InputForm inf) {
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];  // [256][sh_row_stride]
    // zero the depth sort's histogram / ticket / look-back words here instead of with a separate fill launch
    // (zero_area2: the tile-list builder's per-tile totals when NO depth sort runs in front of it — the sort's last pass
    //  clears them otherwise; tile_sort.hip)
    if (PART != GGR_PRE_COLOUR) {
        for (uint32_t wz = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; wz < zero_words;
             wz += gridDim.x * gridDim.y * blockDim.x)
            zero_area[wz] = 0u;
        for (uint32_t wz = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; wz < zero_words2;
             wz += gridDim.x * gridDim.y * blockDim.x)
            zero_area2[wz] = 0u;
    }
    // ---- Gaussian set blockIdx.y of the launch set (ViewSet.sets; one set: nothing moves) ----------------------------
    // The set's inputs are rows [set·P, (set+1)·P) of the caller's arrays, its views are views [v0, v0 + vps): every
    // pointer is rebased once, here, so that the rest of the kernel indexes (view, Gaussian) relative to the set.
    const int set = (int)blockIdx.y, v0 = set * vs.vps;
    const size_t in_off = (size_t)set * (size_t)P, st_off = (size_t)v0 * (size_t)P;  // input rows / per-view state rows
    means3D += 3 * in_off; opacities += in_off;
    if (shs) shs += in_off * (size_t)M * 3;
    if (colors_precomp) colors_precomp += 3 * in_off;
    if (cov3D_precomp) cov3D_precomp += (size_t)inf.cov_stride * in_off;
    if (scales) { scales += 3 * in_off; rotations += 4 * in_off; }
    if (aux_precomp) aux_precomp += st_off;
    radii += st_off; splat += 2 * st_off; colour += st_off; depth_key += st_off;
    if (JAC) sh_jac += st_off;   // (three planes of V·P float4s: channel c of pair o at c·jac_plane + o)
    rect += st_off; clamped_out += st_off;
    if (cov3D_out) cov3D_out += 6 * st_off;
    vs.view += 16 * v0; vs.proj += 16 * v0; vs.campos += 3 * v0;
    if (vs.tanfov) vs.tanfov += 2 * v0;
    if (vs.input_scale) vs.input_scale += v0;
    // Cooperative, coalesced staging of the block's SH rows (the per-Gaussian row is 12·K bytes: read
    // lane-per-Gaussian it would touch 64 different cache lines per load instruction).
    const int sh_deg = ggr_sh_degree(D, shs ? M : 25, inf.sh_cap);
    const int sh_rowf = 3 * (sh_deg + 1) * (sh_deg + 1);
    // odd row length (GGRt: 3·M = 75 floats): the block's rows are ONE contiguous, 16-B aligned region (g0 is
    // a multiple of 256) → copy it flat with float4 loads; an odd LDS stride is already conflict-free for
    // the per-lane row reads, so nothing needs repacking.  Otherwise repack to the odd stride 3K | 1.
    const bool sh_flat = ((M * 3) & 1) != 0 && inf.sh_aligned != 0;
    // channel-major rows ([3][M], GGRt's harmonics layout): coefficient k of channel c sits at c·M + k, so the
    // whole row is staged; k-major rows ([M][3], upstream) only need their first 3K floats
    const int copy_row = inf.sh_channel_major ? M * 3 : sh_rowf;
    const bool sh_compact = M * 3 > sh_rowf && M * 3 <= 128 && (sh_flat || inf.sh_channel_major);
    const int sh_stride = sh_compact ? (sh_rowf | 1) : sh_flat ? M * 3 : (copy_row | 1);
    const int sh_ks = inf.sh_channel_major ? 1 : 3, sh_cs = inf.sh_channel_major ? (sh_compact ? sh_rowf / 3 : M) : 1;
    // A block works on the 256-Gaussian chunks bx = blockIdx.x, blockIdx.x + gridDim.x, …: one chunk per block (gridDim.x =
    // the number of chunks) everywhere except for a THROTTLED colour launch — a few persistent blocks per CU, so that the
    // kernel leaves the CUs' wave slots, LDS and most of the HBM queue to the depth sort it runs beside (api.hip)
    const int nchunks = (P + (int)blockDim.x - 1) / (int)blockDim.x;
#pragma clang loop unroll(disable)
    for (int bx = blockIdx.x; bx < nchunks; bx += gridDim.x) {
    if (bx != (int)blockIdx.x) __syncthreads();   // the previous chunk's rows in LDS have been consumed
    const int i = bx * blockDim.x + threadIdx.x;
    // this thread's own inputs are requested BEFORE the SH staging, so that their round trip overlaps it
    // (clamped index: threads past P load Gaussian P-1 and drop it)
    const size_t il = (size_t)min(i, P - 1);
    const float m0 = means3D[3 * il], m1 = means3D[3 * il + 1], m2 = means3D[3 * il + 2];   // (non-temporal: no difference)
    const float opac = PART != GGR_PRE_COLOUR ? opacities[il] : 0.f;
    float cin[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rin[4] = {0.f, 0.f, 0.f, 0.f};
    if (PART == GGR_PRE_COLOUR) {
    } else if (cov3D_precomp) {
        if (inf.cov_stride == 9) {
            const float* c9 = cov3D_precomp + 9 * il;
            cin[0] = c9[0]; cin[1] = c9[1]; cin[2] = c9[2]; cin[3] = c9[4]; cin[4] = c9[5]; cin[5] = c9[8];
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) cin[k] = cov3D_precomp[6 * il + k];
        }
    } else {
        cin[0] = scales[3 * il]; cin[1] = scales[3 * il + 1]; cin[2] = scales[3 * il + 2];
        cin[3] = cin[4] = cin[5] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) rin[k] = rotations[4 * il + k];
    }
    float cp_in[3] = {0.f, 0.f, 0.f};
    if (colors_precomp) { cp_in[0] = colors_precomp[3 * il]; cp_in[1] = colors_precomp[3 * il + 1]; cp_in[2] = colors_precomp[3 * il + 2]; }
    // chunked staging (KC > 0): sh_stage.h ShThirds
    constexpr int KCN = KC > 0 ? KC : 1, KC_STRIDE = KC | 1;
    ShThirds<KCN> thirds;
    float ch_v[ShThirds<KCN>::ITS];
    if (shs && PART != GGR_PRE_GEOMETRY) {
        const size_t g0 = (size_t)bx * blockDim.x;
        const int nG = (int)min((size_t)blockDim.x, (size_t)P - g0);
        const size_t row = (size_t)M * 3;
        if (KC == 0) {
            if (sh_compact) stage_sh_rows_compact(sh_lds, shs, g0, nG, M, sh_rowf / 3, sh_stride, inf.sh_channel_major != 0);
            else stage_sh_rows(sh_lds, shs, g0, nG, row, copy_row, sh_stride, sh_flat);
            __syncthreads();
        } else {
            thirds.init(shs + g0 * row, nG, (int)row, inf.sh_channel_major ? M : KC);
        }
    }
    const bool in_range = i < P;  // (threads past P run on Gaussian P-1's inputs and store nothing)
    const int gy = (H + GGR_TILE - 1) / GGR_TILE;   // (tile rows per view: the views' tile rows are stacked)
    uint32_t km = 0u;  // largest sort key of this thread over all views
    uint32_t kmn = 0xFFFFFFFFu;   // … and (smallest visible key) − 1

    // ---- one pass per view: the Gaussian's inputs (and its SH row in LDS) are read ONCE for all of them ----------
    const int NV = MULTI ? vs.vps : 1;
#pragma clang loop unroll(disable)
    for (int v = 0; v < NV; v++) {   // (v: view inside the set; its global index is v0 + v)
        float V[16], PM[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { V[k] = vs.view[16 * v + k]; PM[k] = vs.proj[16 * v + k]; }
        const float* campos = vs.campos + 3 * v;
        const float tanfovx = vs.tanfov ? vs.tanfov[2 * v] : vs.tanfovx;      // device-resident tan(fov/2)
        const float tanfovy = vs.tanfov ? vs.tanfov[2 * v + 1] : vs.tanfovy;
        const float in_s = vs.input_scale ? vs.input_scale[v] : 1.0f;
        const size_t o = (size_t)v * P + (size_t)il;  // per-view index of this Gaussian's state
        const float aux_in = (aux_precomp && PART != GGR_PRE_COLOUR) ? aux_precomp[o] : 0.f;
        // COLOUR: what GEOMETRY decided (a visible Gaussian has radius >= 1)
        const int rad_seen = PART == GGR_PRE_COLOUR ? radii[o] : 0;

        // defaults for a culled Gaussian
        int rad_out = 0;
        uint32_t key_out = 0u, clamp_bits = 0;  // sort key 0: culled (ggr_common.h GGR_KEY_BASE)
        uint2 rect_out = make_uint2(0, 0);
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0;

        // call-site fusion: the reference's 1/near renormalisation (means·s, cov·s², scales·s — cuda_splatting.py:
        // 66-73) and its upper-triangle gather out of [P,3,3] covariances (:116,124) happen on load.  One fp32
        // multiply per value, exactly what the torch ops of the unfused call site do.
        const float p0 = in_s * m0, p1 = in_s * m1, p2 = in_s * m2;
        float cov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (PART == GGR_PRE_COLOUR) {
        } else if (cov3D_precomp) {
            const float s2 = in_s * in_s;
#pragma unroll
            for (int k = 0; k < 6; k++) cov6[k] = cin[k] * s2;
        } else {
            float sc[3] = {in_s * cin[0], in_s * cin[1], in_s * cin[2]};
            cov3d_from_scale_rot(sc, scale_modifier, rin, cov6);
            if (in_range) {
#pragma unroll
                for (int k = 0; k < 6; k++) cov3D_out[6 * o + k] = cov6[k];
            }
        }

        // ---- geometry: visible (in front of the near plane, invertible 2D covariance, touches a tile) or culled ----
        bool vis = PART == GGR_PRE_COLOUR && in_range && rad_seen > 0;
        float px = 0.f, py = 0.f, con0 = 0.f, con1 = 0.f, con2 = 0.f;
        float t0 = V[0] * p0 + V[4] * p1 + V[8] * p2 + V[12];
        float t1 = V[1] * p0 + V[5] * p1 + V[9] * p2 + V[13];
        const float t2 = V[2] * p0 + V[6] * p1 + V[10] * p2 + V[14];
        if (PART != GGR_PRE_COLOUR && t2 > GGR_NEAR_CULL && in_range) {
            const float ph0 = PM[0] * p0 + PM[4] * p1 + PM[8] * p2 + PM[12];
            const float ph1 = PM[1] * p0 + PM[5] * p1 + PM[9] * p2 + PM[13];
            const float ph3 = PM[3] * p0 + PM[7] * p1 + PM[11] * p2 + PM[15];
            const float pw = 1.0f / (ph3 + 0.0000001f);
            const float ppx = ph0 * pw, ppy = ph1 * pw;

            const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
            const float limx = GGR_FRUSTUM_CLAMP * tanfovx, limy = GGR_FRUSTUM_CLAMP * tanfovy;
            const float txtz = t0 / t2, tytz = t1 / t2;
            t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
            t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
            const float J00 = fx / t2, J02 = -(fx * t0) / (t2 * t2);
            const float J11 = fy / t2, J12 = -(fy * t1) / (t2 * t2);
            // A = J·R with R[i][j] = V[4*j+i]
            float A0[3], A1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                A0[j] = J00 * V[4 * j + 0] + J02 * V[4 * j + 2];
                A1[j] = J11 * V[4 * j + 1] + J12 * V[4 * j + 2];
            }
            const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
            float AS0[3], AS1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) { a0 += A0[k] * S[3 * k + j]; a1 += A1[k] * S[3 * k + j]; }
                AS0[j] = a0; AS1[j] = a1;
            }
            float c00 = 0.f, c01 = 0.f, c11 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) { c00 += AS0[k] * A0[k]; c01 += AS0[k] * A1[k]; c11 += AS1[k] * A1[k]; }
            const float a = c00 + GGR_DILATION, b = c01, c = c11 + GGR_DILATION;
            const float det = a * c - b * b;
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                con0 = c * det_inv; con1 = -b * det_inv; con2 = a * det_inv;
                const float mid = 0.5f * (a + c);
                const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float l1 = mid + sq, l2 = mid - sq;
                const float radf = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
                px = ((ppx + 1.0f) * (float)W - 1.0f) * 0.5f;
                py = ((ppy + 1.0f) * (float)H - 1.0f) * 0.5f;
                // NON-FINITE INPUTS (include/ggr_raster.h "non-finite inputs"; same test at the same place in
                // oracle/ggr_oracle.c): a Gaussian whose projected geometry or opacity is not finite — or whose radius would
                // overflow the int — takes no part in the frame.  (area stays 0: culled like an off-screen one.)
                const bool finite = isfinite(px) && isfinite(py) && isfinite(con0) && isfinite(con1) && isfinite(con2) &&
                                    isfinite(opac) && isfinite(a) && isfinite(c) && isfinite(t2) && radf < 1073741824.f;
                const int rad = finite ? (int)radf : 0;
                // (the reference clips the rect to the tile grid [0, gx] × [0, gy]; the scissor extension to its window's
                //  tiles — the whole grid unless GgrSettings.scissor is set)
                const int rminx = min(inf.sc_x1, max(inf.sc_x0, (int)((px - (float)rad) / (float)GGR_TILE)));
                const int rminy = min(inf.sc_y1, max(inf.sc_y0, (int)((py - (float)rad) / (float)GGR_TILE)));
                const int rmaxx = min(inf.sc_x1, max(inf.sc_x0, (int)((px + (float)rad + (float)(GGR_TILE - 1)) / (float)GGR_TILE)));
                const int rmaxy = min(inf.sc_y1, max(inf.sc_y0, (int)((py + (float)rad + (float)(GGR_TILE - 1)) / (float)GGR_TILE)));
                const int area = finite ? (rmaxx - rminx) * (rmaxy - rminy) : 0;
                if (area != 0) {   // (visibility — radii, colours — follows the REFERENCE's rect whatever the lists hold)
                    vis = true;
                    rad_out = rad;
                    int tx0 = rminx, ty0 = rminy, tx1 = rmaxx, ty1 = rmaxy;
                    if (inf.tight_rects) {   // ggr_common.h ggr_qmax_upper: tiles the α ≥ 1/255 ellipse cannot reach are dropped
                        const float qmax = ggr_qmax_upper(opac);
                        if (qmax < 0.f) { tx1 = tx0; ty1 = ty0; }
                        else {
                            const float hx = sqrtf(qmax * a) * 1.01f + 0.5f, hy = sqrtf(qmax * c) * 1.01f + 0.5f;
                            tx0 = max(tx0, (int)floorf((px - hx) / (float)GGR_TILE));
                            ty0 = max(ty0, (int)floorf((py - hy) / (float)GGR_TILE));
                            tx1 = min(tx1, (int)floorf((px + hx) / (float)GGR_TILE) + 1);
                            ty1 = min(ty1, (int)floorf((py + hy) / (float)GGR_TILE) + 1);
                            if (tx1 < tx0) tx1 = tx0;
                            if (ty1 < ty0) ty1 = ty0;
                        }
                    }
                    const int area_t = (tx1 - tx0) * (ty1 - ty0);
                    // > 0: t2 > 0.2f.  The three-pass sort takes 30-bit keys: depths ≥ 6.8e37 (incl. +inf) share the last key
                    // and keep ascending id among themselves instead of voiding the frame (ggr_raster.h "depth order")
                    key_out = min(__float_as_uint(t2) - GGR_KEY_BASE, GGR_KEY_MAX);
                    // tile rows of view v sit below those of views 0 … v-1 in the virtual stacked image
                    const uint32_t yo = (uint32_t)((v0 + v) * gy);
                    if (area_t) rect_out = make_uint2((uint32_t)tx0 | (((uint32_t)ty0 + yo) << 16),
                                                      (uint32_t)tx1 | (((uint32_t)ty1 + yo) << 16));
                }
            }
        }

        // ---- colour ---------------------------------------------------------------------------------------------------
        float rgb[3] = {0.f, 0.f, 0.f};
        if (colors_precomp) {
            rgb[0] = cp_in[0]; rgb[1] = cp_in[1]; rgb[2] = cp_in[2];
        } else if (PART != GGR_PRE_GEOMETRY && (KC > 0 || vis)) {
            // (chunked staging: every thread walks the thirds — the barriers between them are block-wide; a culled
            //  Gaussian's sums are dropped below)
            const int deg = KC == 16 ? 3 : KC == 25 ? 4 : sh_deg;
            float d0 = p0 - campos[0], d1 = p1 - campos[1], d2 = p2 - campos[2];
            const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            d0 /= len; d1 /= len; d2 /= len;
            float B[25];
            sh_basis(deg, d0, d1, d2, B);
            float r[3] = {0.f, 0.f, 0.f};
            float jac[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // [channel][x, y, z]
            if (KC == 0) {
                const int K = (deg + 1) * (deg + 1);
                const float* sh = sh_lds + threadIdx.x * sh_stride;
                if (!JAC) {
                    for (int k = 0; k < K; k++) {
                        r[0] += B[k] * sh[k * sh_ks]; r[1] += B[k] * sh[k * sh_ks + sh_cs]; r[2] += B[k] * sh[k * sh_ks + 2 * sh_cs];
                    }
                } else {
                    // colour and Jacobian in ONE walk over the row (the term generator's basis values are sh_basis's,
                    // expression for expression, and the colour sums keep their order and their unfused multiply-add: the
                    // colours stay bit-identical; the Jacobian sums are fused multiply-adds)
                    float x = d0, y = d1, z = d2;
#define SH_JAC(k, Bk, bx, by, bz)                                                                                          \
    {                                                                                                                     \
        const float b_ = (Bk), s0 = sh[(k) * sh_ks], s1 = sh[(k) * sh_ks + sh_cs], s2 = sh[(k) * sh_ks + 2 * sh_cs];         \
        r[0] += b_ * s0; r[1] += b_ * s1; r[2] += b_ * s2;                                                                \
        jac[0] = fmaf((bx), s0, jac[0]); jac[1] = fmaf((by), s0, jac[1]); jac[2] = fmaf((bz), s0, jac[2]);                \
        jac[3] = fmaf((bx), s1, jac[3]); jac[4] = fmaf((by), s1, jac[4]); jac[5] = fmaf((bz), s1, jac[5]);                \
        jac[6] = fmaf((bx), s2, jac[6]); jac[7] = fmaf((by), s2, jac[7]); jac[8] = fmaf((bz), s2, jac[8]);                \
    }
#define SH_FENCE __asm__ volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(xx), "+v"(yy), "+v"(zz), "+v"(xy), "+v"(yz), "+v"(xz) :: "memory");
                    GGR_SH_TERMS(SH_JAC, deg, SH_FENCE)
#undef SH_FENCE
#undef SH_JAC
                }
            } else {
                const float* seg = sh_lds + threadIdx.x * KC_STRIDE;
                const bool cm = inf.sh_channel_major != 0;
                // (the first third is requested here, not before the geometry: its 26 values carried through that code
                //  cost a wave per SIMD — 170 instead of 148 VGPRs at KC = 25 — and the block's neighbours on the CU
                //  cover the round trip)
                thirds.load(ch_v, 0);
#pragma unroll
                for (int J = 0; J < 3; J++) {
                    if (J) __syncthreads();  // the previous third has been consumed
                    thirds.store(sh_lds, ch_v);
                    if (J < 2) thirds.load(ch_v, J + 1);
                    __syncthreads();
                    if (!JAC) {
                        if (cm) {   // third J = channel J, coefficients 0 … KC-1
#pragma unroll
                            for (int k = 0; k < KCN; k++) r[J] += B[k] * seg[k];
                        } else {    // third J = floats [J·KC, (J+1)·KC) of the k-major row: float f = coefficient f / 3, channel f mod 3
#pragma unroll
                            for (int e = 0; e < KCN; e++) {
                                const int f = J * KCN + e;
                                r[f % 3] += B[f / 3] * seg[e];
                            }
                        }
                    } else {
                        // colour and Jacobian in one walk over the third (see the whole-row form above); the basis and its
                        // gradient are formed anew in every third, band by band behind compiler fences — shared across the
                        // three unrolled copies they are ≈ 100 live values
                        float x = d0, y = d1, z = d2;
                        __asm__ volatile("" : "+v"(x), "+v"(y), "+v"(z));
#define SH_JAC_CM(k, Bk, bx, by, bz)                                                                   \
    {                                                                                                  \
        const float s_ = seg[k];                                                                       \
        r[J] += (Bk) * s_;                                                                             \
        jac[3 * J] = fmaf((bx), s_, jac[3 * J]); jac[3 * J + 1] = fmaf((by), s_, jac[3 * J + 1]);      \
        jac[3 * J + 2] = fmaf((bz), s_, jac[3 * J + 2]);                                               \
    }
#define SH_JAC_KM(k, Bk, bx, by, bz)                                                                   \
    _Pragma("unroll") for (int c = 0; c < 3; c++)                                                      \
        if ((3 * (k) + c) / KCN == J) {                                                                \
            const float s_ = seg[3 * (k) + c - J * KCN];                                               \
            r[c] += (Bk) * s_;                                                                         \
            jac[3 * c] = fmaf((bx), s_, jac[3 * c]); jac[3 * c + 1] = fmaf((by), s_, jac[3 * c + 1]);  \
            jac[3 * c + 2] = fmaf((bz), s_, jac[3 * c + 2]);                                           \
        }
#define SH_FENCE __asm__ volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(xx), "+v"(yy), "+v"(zz), "+v"(xy), "+v"(yz), "+v"(xz) :: "memory");
                        if (cm) GGR_SH_TERMS(SH_JAC_CM, (KC == 16 ? 3 : 4), SH_FENCE)
                        if (!cm) GGR_SH_TERMS(SH_JAC_KM, (KC == 16 ? 3 : 4), SH_FENCE)
#undef SH_FENCE
#undef SH_JAC_CM
#undef SH_JAC_KM
                    }
                }
            }
            if (JAC && vis && in_range) {
                // (one plane per channel: a wave's store covers 1 KB of whole lines — as one 48-B record per Gaussian the
                //  three partial-line stores cost the kernel 16 µs at C3 on top of the bytes)
                ggr_st_f4(reinterpret_cast<float*>(sh_jac + o), make_float4(jac[0], jac[1], jac[2], 0.f));
                ggr_st_f4(reinterpret_cast<float*>(sh_jac + jac_plane + o), make_float4(jac[3], jac[4], jac[5], 0.f));
                ggr_st_f4(reinterpret_cast<float*>(sh_jac + 2 * jac_plane + o), make_float4(jac[6], jac[7], jac[8], 0.f));
            }
            r[0] += 0.5f; r[1] += 0.5f; r[2] += 0.5f;
            if (vis) clamp_bits = (r[0] < 0.f ? 1u : 0u) | (r[1] < 0.f ? 2u : 0u) | (r[2] < 0.f ? 4u : 0u);
            rgb[0] = fmaxf(r[0], 0.f); rgb[1] = fmaxf(r[1], 0.f); rgb[2] = fmaxf(r[2], 0.f);
            if (!(isfinite(r[0]) && isfinite(r[1]) && isfinite(r[2]))) rgb[0] = r[0] + r[1] + r[2];   // (NaN / Inf: caught below)
        }
        // non-finite colour (the contract above, colour part): the Gaussian leaves the frame
        if (vis && !(isfinite(rgb[0]) && isfinite(rgb[1]) && isfinite(rgb[2]))) {
            vis = false;
            rgb[0] = rgb[1] = rgb[2] = 0.f;
            clamp_bits = 0u;
            rad_out = 0; key_out = 0u; rect_out = make_uint2(0, 0);
            if (PART == GGR_PRE_COLOUR && in_range) {
                // the geometry half has listed it already: its record gets opacity 0 and qmax < 0 — no quadrant cull keeps
                // it, no pixel takes it — and its radius becomes 0 (no gradient)
                float4 g1 = splat[2 * o + 1];
                g1.y = 0.f; g1.w = -1.f;
                splat[2 * o + 1] = g1;
                ggr_st(radii + o, 0);
            }
        }
        if (vis) {
            s0 = make_float4(px, py, con0, con1);
            // 4th blended feature: the caller's aux value, or view z, or (fused GGRt depth pass, :240-269)
            // max(a + b·z_unscaled, 0) with z_unscaled = z / s
            float feat = t2;
            if (aux_precomp) feat = aux_in;
            else if (inf.aux_affine) feat = fmaxf(inf.aux_a + inf.aux_b * (t2 / in_s), 0.f);
            s1 = make_float4(con2, opac, feat, 2.f * logf(255.f * opac));  // .w = qmax for the box cull
            s2 = make_float4(rgb[0], rgb[1], rgb[2], 0.f);
        }
        if (in_range) {
            if (PART != GGR_PRE_COLOUR) {
                ggr_st(radii + o, rad_out);   // (an output tensor: not read again before the backward — or, split, by COLOUR)
                depth_key[o] = key_out;       // written straight into the depth sort's key / value input buffers
                // (no sort VALUES are written: the depth sort's first pass forms the identity — the global (view, Gaussian)
                //  index — itself; and no tiles_touched: it is the area of the packed rect.  Two output streams and 8 B per
                //  Gaussian less)
                rect[o] = rect_out;
                splat[2 * o] = s0;          // (plain stores: the records, keys and rects are read again within the forward —
                splat[2 * o + 1] = s1;      //  non-temporal they measured the same or worse)
            }
            // the colour record and the clamp bits: by whoever evaluates the colour (precomputed colours: GEOMETRY)
            if (PART != GGR_PRE_GEOMETRY || colors_precomp) {
                colour[o] = s2;
                ggr_st(clamped_out + o, clamp_bits);   // (not read again before the backward)
            }
        }
        km = max(km, key_out);
        kmn = min(kmn, key_out - 1u);   // (a culled Gaussian's key 0 wraps to ~0: the minimum is over the VISIBLE keys, less one)
    }
    // the largest sort key of this block: the depth sort derives its digit width from these (binning.hip)
    if (PART == GGR_PRE_COLOUR) continue;
    // … and the smallest VISIBLE one (less one; ~0: none): the bucket form of the sort takes the frame's key range from both
    __shared__ uint32_t kmax[GGR_PRE_THREADS / 64], kmin[GGR_PRE_THREADS / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        km = max(km, (uint32_t)__shfl_xor((int)km, off));
        kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, off));
    }
    if ((threadIdx.x & 63) == 0) { kmax[threadIdx.x >> 6] = km; kmin[threadIdx.x >> 6] = kmn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_max[blockIdx.y * nchunks + bx] = max(max(kmax[0], kmax[1]), max(kmax[2], kmax[3]));
        block_min[blockIdx.y * nchunks + bx] = min(min(kmin[0], kmin[1]), min(kmin[2], kmin[3]));
    }
    }   // chunks
}

void launch_preprocess_fwd(int P, int D, int M, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, float scale_modifier, const float* cov3D_precomp,
                           const float* aux_precomp, ViewSet vs, int W, int H, int32_t* radii,
                           GeomLayout g, InputForm inf, hipStream_t s, int part, int colour_grid, int keep_jacobian,
                           uint32_t* zero_area2, uint32_t zero_words2, int sort_area_untouched) {
    if (P <= 0) return;
    if (part == GGR_PRE_COLOUR && !shs) return;   // precomputed colours: GEOMETRY has written the colour records
    // the depth sort's work area (binning.hip), sized for the V·P keys of all views
    // (sort_area_untouched: no depth sort will read the area — unless the per-tile sort has to be given up, and then the
    //  host clears it — so it is not cleared here: 9 MB of stores at C3; the block maxima are left where the sort expects them)
    const uint32_t sort_words = (uint32_t)ggr_sort_zero_words((size_t)P * vs.V, ggr_sort_segments((size_t)vs.V));
    const uint32_t zero_words = sort_area_untouched ? 0u : sort_words;
    const int threads = GGR_PRE_THREADS;
    const int chunks = (P + threads - 1) / threads;
    // (colour_grid > 0, COLOUR only: that many persistent blocks walk the chunks — see the kernel)
    const int blocks = (part == GGR_PRE_COLOUR && colour_grid > 0) ? std::min(chunks, colour_grid) : chunks;
    const int deg = ggr_sh_degree(D, shs ? M : 25, inf.sh_cap);
    const bool flat = ((3 * M) & 1) && inf.sh_aligned;  // same predicate as the kernel
    const size_t copy_row = inf.sh_channel_major ? (size_t)(3 * M) : (size_t)(3 * (deg + 1) * (deg + 1));
    const size_t rowf = (size_t)(3 * (deg + 1) * (deg + 1));
    const bool compact = (size_t)(3 * M) > rowf && 3 * M <= 128 && (flat || inf.sh_channel_major);
    const size_t row_stride = compact ? (rowf | 1) : flat ? (size_t)(3 * M) : (copy_row | 1);
    // one view at degree 3 / 4: the rows go through LDS a third at a time (see the kernel's header)
    const int kc = (shs && vs.vps == 1 && (deg == 3 || deg == 4)) ? (deg + 1) * (deg + 1) : 0;
    const size_t lds = (!shs || part == GGR_PRE_GEOMETRY) ? 0 : kc ? (size_t)threads * (kc | 1) * sizeof(float)
                                                                   : (size_t)threads * row_stride * sizeof(float);
#define GGR_LAUNCH_PFWD_J(MULTI_, KC_, PART_, JAC_)                                                                       \
    hipLaunchKernelGGL((preprocess_fwd_kernel<MULTI_, KC_, PART_, JAC_>), dim3(blocks, vs.sets), dim3(threads), lds, s, P, D, \
                       M, means3D, shs, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,       \
                       aux_precomp, vs, W, H, radii, g.splat, g.colour, g.sh_jac, (size_t)P * vs.V, g.keys_a, g.rect, g.clamped, g.cov3D,    \
                       g.hist, zero_words, zero_area2, zero_words2, g.hist + sort_words,                                  \
                       g.hist + ggr_sort_block_min_at((size_t)P * vs.V, ggr_sort_segments((size_t)vs.V)), inf)
    const bool jac = keep_jacobian != 0 && shs != nullptr;
#define GGR_LAUNCH_PFWD(MULTI_, KC_, PART_)                                                                               \
    do { if (jac) GGR_LAUNCH_PFWD_J(MULTI_, KC_, PART_, true); else GGR_LAUNCH_PFWD_J(MULTI_, KC_, PART_, false); } while (0)
    if (part == GGR_PRE_GEOMETRY) {   // (no SH rows: no staging variant, no Jacobian)
        if (vs.vps > 1) GGR_LAUNCH_PFWD_J(true, 0, GGR_PRE_GEOMETRY, false);
        else GGR_LAUNCH_PFWD_J(false, 0, GGR_PRE_GEOMETRY, false);
    } else if (part == GGR_PRE_COLOUR) {
        if (vs.vps > 1) GGR_LAUNCH_PFWD(true, 0, GGR_PRE_COLOUR);
        else if (kc == 16) GGR_LAUNCH_PFWD(false, 16, GGR_PRE_COLOUR);
        else if (kc == 25) GGR_LAUNCH_PFWD(false, 25, GGR_PRE_COLOUR);
        else GGR_LAUNCH_PFWD(false, 0, GGR_PRE_COLOUR);
    } else {
        if (vs.vps > 1) GGR_LAUNCH_PFWD(true, 0, GGR_PRE_ALL);
        else if (kc == 16) GGR_LAUNCH_PFWD(false, 16, GGR_PRE_ALL);
        else if (kc == 25) GGR_LAUNCH_PFWD(false, 25, GGR_PRE_ALL);
        else GGR_LAUNCH_PFWD(false, 0, GGR_PRE_ALL);
    }
#undef GGR_LAUNCH_PFWD_J
#undef GGR_LAUNCH_PFWD
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                    const float* __restrict__ V, uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float p0 = means3D[3 * i], p1 = means3D[3 * i + 1], p2 = means3D[3 * i + 2];
    const float t2 = V[2] * p0 + V[6] * p1 + V[10] * p2 + V[14];
    present[i] = t2 > GGR_NEAR_CULL ? 1 : 0;
}

void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}

__global__ void unpack_geom_kernel(int P, const float4* __restrict__ splat, const float4* __restrict__ colour,
                                   const uint2* __restrict__ rect,
                                   const uint32_t* __restrict__ cl, float* depth, float* xy, float* co,
                                   float* rgb, int32_t* tiles, uint8_t* clamped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 s0 = splat[2 * (size_t)i], s1 = splat[2 * (size_t)i + 1], s2 = colour[i];
    if (depth) depth[i] = s1.z;
    if (xy) { xy[2 * i] = s0.x; xy[2 * i + 1] = s0.y; }
    if (co) { co[4 * i] = s0.z; co[4 * i + 1] = s0.w; co[4 * i + 2] = s1.x; co[4 * i + 3] = s1.y; }
    if (rgb) { rgb[3 * i] = s2.x; rgb[3 * i + 1] = s2.y; rgb[3 * i + 2] = s2.z; }
    if (tiles) {   // tiles_touched = area of the packed tile rect (minx | miny << 16, maxx | maxy << 16)
        const uint2 rc = rect[i];
        tiles[i] = (int32_t)(((rc.y & 0xFFFFu) - (rc.x & 0xFFFFu)) * ((rc.y >> 16) - (rc.x >> 16)));
    }
    if (clamped) { const uint32_t c = cl[i]; clamped[3 * i] = c & 1; clamped[3 * i + 1] = (c >> 1) & 1; clamped[3 * i + 2] = (c >> 2) & 1; }
}

void launch_unpack_geom(GeomLayout g, int P, float* depth, float* xy, float* conic_opacity, float* rgb,
                        int32_t* tiles_touched, uint8_t* clamped, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(unpack_geom_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.splat, g.colour, g.rect,
                       g.clamped, depth, xy, conic_opacity, rgb, tiles_touched, clamped);
}

}  // namespace ggr
