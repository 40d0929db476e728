// preprocess_bwd.hip — per-Gaussian backward for gfx950 (one lane per Gaussian, streaming).
//
// Fuses the `computeCov2DCUDA` (backward) and `preprocessCUDA` (backward) stages of the rasterizer
// behind reference cuda_splatting.py:114-125 / train_ggrt_stable.py:143 (SURVEY.md §2.2, Appendix
// A.4-A.5): conic → cov2D → cov3D[6] and mean3D (through J, frozen on frustum-clamped axes),
// mean2D → mean3D through the perspective divide, SH backward (+ view-direction term), optional
// scale/rotation backward.  Optionally accumulates the camera gradients (viewmatrix, projmatrix,
// campos) — an extension beyond the reference.
#include "ggr_common.h"
#include "sh_stage.h"
#include "sh_terms.h"
#include <algorithm>

namespace ggr {

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes, valid in lane 63 (DPP row shifts + row broadcasts; no LDS traffic)
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_take<0x111>(v);        // row_shr:1
    v += dpp_take<0x112>(v);        // row_shr:2
    v += dpp_take<0x114>(v);        // row_shr:4
    v += dpp_take<0x118>(v);        // row_shr:8  → lane 15 of each row = row sum
    v += dpp_take<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
    v += dpp_take<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3
    return v;
}

#define GGR_SH_MAXK 25

// SH basis in the rasterizer's sign convention (same values as preprocess_fwd's); B must hold 25 floats
__device__ __forceinline__ void sh_basis25(int deg, float x, float y, float z, float* B) {
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = bSH_C2[0] * xy; B[5] = bSH_C2[1] * yz; B[6] = bSH_C2[2] * (2.f * zz - xx - yy);
            B[7] = bSH_C2[3] * xz; B[8] = bSH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = bSH_C3[0] * y * (3.f * xx - yy);
                B[10] = bSH_C3[1] * xy * z;
                B[11] = bSH_C3[2] * y * (4.f * zz - xx - yy);
                B[12] = bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                B[13] = bSH_C3[4] * x * (4.f * zz - xx - yy);
                B[14] = bSH_C3[5] * z * (xx - yy);
                B[15] = bSH_C3[6] * x * (xx - 3.f * yy);
                if (deg > 3) {
                    B[16] = bSH_C4[0] * xy * (xx - yy);
                    B[17] = bSH_C4[1] * yz * (3.f * xx - yy);
                    B[18] = bSH_C4[2] * xy * (7.f * zz - 1.f);
                    B[19] = bSH_C4[3] * yz * (7.f * zz - 3.f);
                    B[20] = bSH_C4[4] * (zz * (35.f * zz - 30.f) + 3.f);
                    B[21] = bSH_C4[5] * xz * (7.f * zz - 3.f);
                    B[22] = bSH_C4[6] * (xx - yy) * (7.f * zz - 1.f);
                    B[23] = bSH_C4[7] * xz * (xx - 3.f * yy);
                    B[24] = bSH_C4[8] * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
                }
            }
        }
    }
}

// MULTI = false: one view, the loop over views folds away at compile time (the reference's backward, and its register
// budget); MULTI = true: the launch set's views in a run-time loop
// KC = 16 / 25 (one view, degree 3 / 4, rows of more than 64 floats): the SH rows never sit whole in LDS (76.8 KB per
// block at GGRt's 75 floats = 2 blocks per CU).  They are READ one third at a time (sh_stage.h ShThirds) for the
// view-direction term; the gradient rows — rank one, B_k(direction)·dL/dcolour_c — are then built by their owner
// threads and copied out flat in three ROW ranges (whole 128-B lines; written by column thirds the partial lines cost
// more than the occupancy gives: tools/sh_stage_bench.hip, 0.149 ms whole rows / 0.214 column thirds / 0.129 this).
// CM (KC > 0 only): channel-major rows — a template parameter because with both row forms in one kernel the compiler
// shares the basis gradients across the two branches: 199 VGPRs instead of 162 / 131.
template <bool POSE, bool MULTI, int KC, bool CM>
__global__ void __launch_bounds__(256)
preprocess_bwd_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
                      const float4* __restrict__ sh_jac, size_t jac_plane, int has_colors_precomp, const float* __restrict__ scales,
                      const float* __restrict__ rotations, float scale_modifier,
                      const float* __restrict__ cov3D, ViewSet vs, int W, int H,
                      const int32_t* __restrict__ radii,
                      const uint32_t* __restrict__ clamped, const float* __restrict__ grad2d, int has_dz,
                      float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
                      float* __restrict__ dL_dopacity, float* __restrict__ dL_dsh, float* __restrict__ dL_dcolors_precomp,
                      float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscales,
                      float* __restrict__ dL_drotations, float* __restrict__ dL_daux,
                      float* __restrict__ pose_acc, InputForm inf, int cov_is_input) {
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];  // [256][sh_stride]: SH in, dL/dSH out
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = i < P;
    const int NV = MULTI ? vs.vps : 1;
    // ---- Gaussian set blockIdx.y of the launch set (as in preprocess_fwd): inputs and gradients are rows
    // [set·P, (set+1)·P) of the caller's arrays, the set's views are views [v0, v0 + vps) — every pointer is rebased here
    {
        const int set = (int)blockIdx.y, v0 = set * vs.vps;
        const size_t in_off = (size_t)set * (size_t)P, st_off = (size_t)v0 * (size_t)P;
        means3D += 3 * in_off;
        if (shs) shs += in_off * (size_t)M * 3;
        if (scales) { scales += 3 * in_off; rotations += 4 * in_off; }
        if (cov3D) cov3D += cov_is_input ? (size_t)inf.cov_stride * in_off : 6 * st_off;
        radii += st_off; clamped += st_off; grad2d += GGR_G2D_STRIDE * st_off; sh_jac += st_off;
        dL_dmeans2D += 3 * st_off;
        if (dL_daux) dL_daux += st_off;
        dL_dmeans3D += 3 * in_off; dL_dopacity += in_off;
        if (dL_dsh) dL_dsh += in_off * (size_t)M * 3;
        if (dL_dcolors_precomp) dL_dcolors_precomp += 3 * in_off;
        if (dL_dcov3D) dL_dcov3D += (size_t)(cov_is_input ? inf.cov_stride : 6) * in_off;
        if (dL_dscales) { dL_dscales += 3 * in_off; dL_drotations += 4 * in_off; }
        if (pose_acc) pose_acc += (size_t)v0 * gridDim.x * 64;
        vs.view += 16 * v0; vs.proj += 16 * v0; vs.campos += 3 * v0;
        if (vs.tanfov) vs.tanfov += 2 * v0;
        if (vs.input_scale) vs.input_scale += v0;
    }
    // coalesced staging of the block's SH rows (same reason as in preprocess_fwd)
    // (several views per set: the whole SH backward — direction term and gradient rows — is preprocess_bwd_sh_views_kernel's)
    const bool use_sh = !MULTI && !has_colors_precomp && shs != nullptr;
    const int deg = ggr_sh_degree(D, use_sh ? M : 25, inf.sh_cap);
    const int sh_rowf = 3 * (deg + 1) * (deg + 1);
    const size_t g0 = (size_t)blockIdx.x * blockDim.x;
    const int nG = (int)min((size_t)blockDim.x, (size_t)P - g0);
    const size_t sh_row = (size_t)M * 3;
    // odd row length (GGRt: 3·M = 75): the block's rows are ONE contiguous 16-B aligned region (g0 is a
    // multiple of 256) → flat float4 copy in and out; an odd LDS stride is conflict-free as it is
    const bool sh_flat = (sh_row & 1) != 0 && inf.sh_aligned != 0;
    // (input forms as in preprocess_fwd: channel-major rows are staged — and their gradient written — whole)
    const int copy_row = inf.sh_channel_major ? (int)sh_row : sh_rowf;
    // rows longer than what is used (GGRt with sh_max_degree 3: 25 coefficients, 16 used): only the used 3K floats live in LDS
    const bool sh_compact = (int)sh_row > sh_rowf && sh_row <= 128 && (sh_flat || inf.sh_channel_major);
    const int sh_stride = sh_compact ? (sh_rowf | 1) : sh_flat ? (int)sh_row : (copy_row | 1);
    const int sh_ks = inf.sh_channel_major ? 1 : 3, sh_cs = inf.sh_channel_major ? (sh_compact ? sh_rowf / 3 : M) : 1;
    // this thread's own inputs are requested BEFORE the SH staging, so that their round trip overlaps it
    const size_t il = (size_t)min(i, P - 1);
    const float m0 = ggr_ld(means3D + 3 * il), m1 = ggr_ld(means3D + 3 * il + 1), m2 = ggr_ld(means3D + 3 * il + 2);
    float cin_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cov_is_input) {   // the caller's covariances, in the caller's form; the scale / rotation path reads the
        if (inf.cov_stride == 9) {  // per-view ones preprocess_fwd stored (already scaled) inside the view loop
            const float* c9 = cov3D + 9 * il;
            cin_in[0] = c9[0]; cin_in[1] = c9[1]; cin_in[2] = c9[2]; cin_in[3] = c9[4]; cin_in[4] = c9[5]; cin_in[5] = c9[8];
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) cin_in[k] = ggr_ld(cov3D + 6 * il + k);
        }
    }
    // (the SH rows are NOT read: the forward left ∂colour/∂direction per (view, Gaussian) — sh_jac — and the gradient rows
    //  are rank one, basis × dL/dcolour; the LDS rows below only carry the gradient out in whole lines)

    // sums over the views of this launch set (one view: the reference's backward)
    float dmean[3] = {0.f, 0.f, 0.f};      // w.r.t. the caller's (unscaled) means
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // w.r.t. the caller's covariances (cov_is_input) — else unused
    float dsc[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dop = 0.f, dcp[3] = {0.f, 0.f, 0.f};
    const int K = (deg + 1) * (deg + 1);

    // the per-view loads (gradient record, radius, clamp bits) are requested ONE VIEW AHEAD: with them inside the
    // iteration that uses them every view is a dependent ≈ 2 µs round trip per block
    const float4* recs = reinterpret_cast<const float4*>(grad2d);
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;
    float2 n2 = make_float2(0.f, 0.f);
    int nrad = 0;
    {
        const float4* rec = recs + (GGR_G2D_STRIDE / 4) * il;
        n0 = ggr_ld_f4(reinterpret_cast<const float*>(rec)); n1 = ggr_ld_f4(reinterpret_cast<const float*>(rec + 1));
        n2 = *reinterpret_cast<const float2*>(rec + 2); nrad = radii[il];
    }
#pragma clang loop unroll(disable)
    for (int v = 0; v < NV; v++) {
        // (compiler barrier: without it the 3K SH coefficient reads from LDS below — invariant across views — are
        // hoisted out of the loop into 75 registers, and the kernel needs 256 VGPRs = one block per CU)
        __asm__ volatile("" ::: "memory");
        const size_t o = (size_t)v * P + il;   // this Gaussian's state for view v
        // the blend backward's per-Gaussian record (ggr_common.h GGR_G2D_*): its mean2D and opacity entries are
        // final results; a Gaussian no tile list holds still has its zeroed record
        const float4 r0 = n0;  // r, g, b, mean.x
        const float4 r1 = n1;  // mean.y, conic xx, xy, yy
        const float2 r2 = n2;  // opacity, z
        const bool live = in_range && nrad > 0;
        if (MULTI && v + 1 < NV) {
            const float4* rec = recs + (GGR_G2D_STRIDE / 4) * (o + P);
            n0 = rec[0]; n1 = rec[1]; n2 = *reinterpret_cast<const float2*>(rec + 2); nrad = radii[o + P];
        }
        const float g_z = r2.y;
        if (in_range) {
            ggr_st(dL_dmeans2D + 3 * o, r0.w); ggr_st(dL_dmeans2D + 3 * o + 1, r1.x); ggr_st(dL_dmeans2D + 3 * o + 2, 0.f);
            dop += r2.x;
        }
        float V[16], PM[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { V[k] = vs.view[16 * v + k]; PM[k] = vs.proj[16 * v + k]; }
        const float tanfovx = vs.tanfov ? vs.tanfov[2 * v] : vs.tanfovx;      // device-resident tan(fov/2)
        const float tanfovy = vs.tanfov ? vs.tanfov[2 * v + 1] : vs.tanfovy;
        const float in_s = vs.input_scale ? vs.input_scale[v] : 1.0f;
        float dV[16], dPM[16];
        if (POSE) {
#pragma unroll
            for (int k = 0; k < 16; k++) { dV[k] = 0.f; dPM[k] = 0.f; }
        }
        if (in_range && dL_daux) dL_daux[o] = (live && has_dz) ? g_z : 0.f;  // aux feature: gradient is the blend's

        if (live) {
            const float p0 = in_s * m0, p1 = in_s * m1, p2 = in_s * m2;
            float cov6[6];
            // the caller's covariances come in the caller's form (preprocess_fwd applies the same); on the scale /
            // rotation path they are what preprocess_fwd stored for this view (already scaled)
            const float s2 = cov_is_input ? in_s * in_s : 1.0f;
            if (cov_is_input) {
#pragma unroll
                for (int k = 0; k < 6; k++) cov6[k] = cin_in[k] * s2;
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) cov6[k] = cov3D[6 * o + k];
            }
            const float dcon0 = r1.y, dcon1 = r1.z, dcon2 = r1.w;

            const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
            float t0 = V[0] * p0 + V[4] * p1 + V[8] * p2 + V[12];
            float t1 = V[1] * p0 + V[5] * p1 + V[9] * p2 + V[13];
            const float t2 = V[2] * p0 + V[6] * p1 + V[10] * p2 + V[14];
            const float limx = GGR_FRUSTUM_CLAMP * tanfovx, limy = GGR_FRUSTUM_CLAMP * tanfovy;
            const float txtz = t0 / t2, tytz = t1 / t2;
            const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
            const float ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
            t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
            t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
            const float J00 = fx / t2, J02 = -(fx * t0) / (t2 * t2);
            const float J11 = fy / t2, J12 = -(fy * t1) / (t2 * t2);
            float A0[3], A1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                A0[j] = J00 * V[4 * j + 0] + J02 * V[4 * j + 2];
                A1[j] = J11 * V[4 * j + 1] + J12 * V[4 * j + 2];
            }
            const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
            float SA0[3], SA1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                SA0[j] = A0[0] * S[j] + A0[1] * S[3 + j] + A0[2] * S[6 + j];
                SA1[j] = A1[0] * S[j] + A1[1] * S[3 + j] + A1[2] * S[6 + j];
            }
            const float a = SA0[0] * A0[0] + SA0[1] * A0[1] + SA0[2] * A0[2] + GGR_DILATION;
            const float b = SA0[0] * A1[0] + SA0[1] * A1[1] + SA0[2] * A1[2];
            const float c = SA1[0] * A1[0] + SA1[1] * A1[1] + SA1[2] * A1[2] + GGR_DILATION;
            const float denom = a * c - b * b;
            const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
            float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
            float dcv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // this view's dL/d(scaled covariance)
            if (denom2inv != 0.f) {
                dL_da = denom2inv * (-c * c * dcon0 + 2.f * b * c * dcon1 + (denom - a * c) * dcon2);
                dL_dc = denom2inv * (-a * a * dcon2 + 2.f * a * b * dcon1 + (denom - a * c) * dcon0);
                dL_db = denom2inv * 2.f * (b * c * dcon0 - (denom + 2.f * b * b) * dcon1 + a * b * dcon2);
                dcv[0] = A0[0] * A0[0] * dL_da + A0[0] * A1[0] * dL_db + A1[0] * A1[0] * dL_dc;
                dcv[3] = A0[1] * A0[1] * dL_da + A0[1] * A1[1] * dL_db + A1[1] * A1[1] * dL_dc;
                dcv[5] = A0[2] * A0[2] * dL_da + A0[2] * A1[2] * dL_db + A1[2] * A1[2] * dL_dc;
                dcv[1] = 2.f * A0[0] * A0[1] * dL_da + (A0[0] * A1[1] + A0[1] * A1[0]) * dL_db + 2.f * A1[0] * A1[1] * dL_dc;
                dcv[2] = 2.f * A0[0] * A0[2] * dL_da + (A0[0] * A1[2] + A0[2] * A1[0]) * dL_db + 2.f * A1[0] * A1[2] * dL_dc;
                dcv[4] = 2.f * A0[2] * A0[1] * dL_da + (A0[1] * A1[2] + A0[2] * A1[1]) * dL_db + 2.f * A1[1] * A1[2] * dL_dc;
            }
            float dA0[3], dA1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                dA0[j] = 2.f * SA0[j] * dL_da + SA1[j] * dL_db;
                dA1[j] = 2.f * SA1[j] * dL_dc + SA0[j] * dL_db;
            }
            // A = J·R, R[k][j] = V[4*j+k]
            const float dJ00 = dA0[0] * V[0] + dA0[1] * V[4] + dA0[2] * V[8];
            const float dJ02 = dA0[0] * V[2] + dA0[1] * V[6] + dA0[2] * V[10];
            const float dJ11 = dA1[0] * V[1] + dA1[1] * V[5] + dA1[2] * V[9];
            const float dJ12 = dA1[0] * V[2] + dA1[1] * V[6] + dA1[2] * V[10];
            const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
            const float dtx = xmul * -fx * tz2 * dJ02;
            const float dty = ymul * -fy * tz2 * dJ12;
            const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * t0) * tz3 * dJ02 + (2.f * fy * t1) * tz3 * dJ12;
            // t = [p 1]·V → dL/dp = R^T dt  (R row k = V[4*j+k] over j)
            float dmv[3];  // this view's dL/d(scaled mean)
            dmv[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
            dmv[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
            dmv[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
            if (POSE) {
                // through t: dL/dV[4*j+k] += p_j * dt_k (j<3), dL/dV[12+k] += dt_k
                const float dt[3] = {dtx, dty, dtz};
                const float pp[3] = {p0, p1, p2};
#pragma unroll
                for (int k = 0; k < 3; k++) {
#pragma unroll
                    for (int j = 0; j < 3; j++) dV[4 * j + k] += pp[j] * dt[k];
                    dV[12 + k] += dt[k];
                }
                // through R in A = J·R: dL/dR[k][j] = Σ_i J[i][k] dA[i][j];  R[k][j] = V[4*j+k]
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    dV[4 * j + 0] += J00 * dA0[j];
                    dV[4 * j + 1] += J11 * dA1[j];
                    dV[4 * j + 2] += J02 * dA0[j] + J12 * dA1[j];
                }
            }

            // mean2D (NDC units) → mean3D through the perspective divide
            const float d2x = r0.w, d2y = r1.x;
            const float mh0 = PM[0] * p0 + PM[4] * p1 + PM[8] * p2 + PM[12];
            const float mh1 = PM[1] * p0 + PM[5] * p1 + PM[9] * p2 + PM[13];
            const float mh3 = PM[3] * p0 + PM[7] * p1 + PM[11] * p2 + PM[15];
            const float mw = 1.0f / (mh3 + 0.0000001f);
            const float mul1 = mh0 * mw * mw, mul2 = mh1 * mw * mw;
            dmv[0] += (PM[0] * mw - PM[3] * mul1) * d2x + (PM[1] * mw - PM[3] * mul2) * d2y;
            dmv[1] += (PM[4] * mw - PM[7] * mul1) * d2x + (PM[5] * mw - PM[7] * mul2) * d2y;
            dmv[2] += (PM[8] * mw - PM[11] * mul1) * d2x + (PM[9] * mw - PM[11] * mul2) * d2y;
            if (POSE) {
                // ndc_x = mh0*mw, ndc_y = mh1*mw:  d/dmh0 = mw·d2x, d/dmh1 = mw·d2y, d/dmh3 = -(mul1·d2x + mul2·d2y)
                const float g0_ = mw * d2x, g1_ = mw * d2y, g3_ = -(mul1 * d2x + mul2 * d2y);
                const float pp[4] = {p0, p1, p2, 1.f};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    dPM[4 * j + 0] += pp[j] * g0_;
                    dPM[4 * j + 1] += pp[j] * g1_;
                    dPM[4 * j + 3] += pp[j] * g3_;
                }
            }

            // depth-as-feature gradient: z = t2 = [p 1]·V[:,2]
            if (has_dz && !dL_daux) {
                float gz = g_z;
                if (inf.aux_affine)  // feature = max(a + b·z/s, 0)
                    gz = (inf.aux_a + inf.aux_b * (t2 / in_s) > 0.f) ? gz * (inf.aux_b / in_s) : 0.f;
                dmv[0] += V[2] * gz; dmv[1] += V[6] * gz; dmv[2] += V[10] * gz;
                if (POSE) { dV[2] += p0 * gz; dV[6] += p1 * gz; dV[10] += p2 * gz; dV[14] += gz; }
            }

            // colour: precomputed colours take the blend's gradient as it is; the SH path follows in its own loops
            // over the views below (keeping it inside this loop costs ≈ 100 more live registers)
            if (has_colors_precomp) { dcp[0] += r0.x; dcp[1] += r0.y; dcp[2] += r0.z; }
            // chain through the on-load input forms: means·s, cov·s²
            dmean[0] += in_s * dmv[0]; dmean[1] += in_s * dmv[1]; dmean[2] += in_s * dmv[2];
            if (cov_is_input) {
#pragma unroll
                for (int k = 0; k < 6; k++) dcov[k] += dcv[k] * s2;
            }

            if (scales && dL_dscales) {
                const float r = rotations[4 * i], x = rotations[4 * i + 1], y = rotations[4 * i + 2], z = rotations[4 * i + 3];
                const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                                    2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                                    2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
                const float sm = scale_modifier * in_s;  // d(sc)/d(scale input)
                const float sc[3] = {scale_modifier * (in_s * scales[3 * i]), scale_modifier * (in_s * scales[3 * i + 1]),
                                     scale_modifier * (in_s * scales[3 * i + 2])};
                float Mx[9];
#pragma unroll
                for (int ii = 0; ii < 3; ii++)
#pragma unroll
                    for (int j = 0; j < 3; j++) Mx[3 * ii + j] = R[3 * ii + j] * sc[j];
                const float dS[9] = {dcv[0], 0.5f * dcv[1], 0.5f * dcv[2], 0.5f * dcv[1], dcv[3], 0.5f * dcv[4],
                                     0.5f * dcv[2], 0.5f * dcv[4], dcv[5]};
                float dR[9];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    float acc = 0.f;
#pragma unroll
                    for (int ii = 0; ii < 3; ii++) {
                        float dMij = 0.f;
#pragma unroll
                        for (int k = 0; k < 3; k++) dMij += dS[3 * ii + k] * Mx[3 * k + j];
                        dMij *= 2.f;
                        acc += dMij * R[3 * ii + j];
                        dR[3 * ii + j] = dMij * sc[j];
                    }
                    dsc[j] += acc * sm;
                }
                drot[0] += 2.f * (z * (dR[3] - dR[1]) + y * (dR[2] - dR[6]) + x * (dR[7] - dR[5]));
                drot[1] += 2.f * (y * (dR[1] + dR[3]) + z * (dR[2] + dR[6]) + r * (dR[7] - dR[5])) - 4.f * x * (dR[4] + dR[8]);
                drot[2] += 2.f * (x * (dR[1] + dR[3]) + r * (dR[2] - dR[6]) + z * (dR[5] + dR[7])) - 4.f * y * (dR[0] + dR[8]);
                drot[3] += 2.f * (r * (dR[3] - dR[1]) + x * (dR[2] + dR[6]) + y * (dR[5] + dR[7])) - 4.f * z * (dR[0] + dR[4]);
            }
        }
        if (POSE) {
            // Camera gradient of view v: 35 components summed over ALL Gaussians.  Atomics would put ≈ P/64 × 35 adds on
            // 35 addresses (measured: 2.5 ms at P = 1 M); instead wave (DPP) → block (LDS) reduction, one row of
            // 35 partials per (view, block) written with plain stores, and pose_finish_kernel sums the rows.
            __shared__ float wred[4][40];  // 640 B: keeps the dynamic LDS base 16-B aligned (guide G17)
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float sv = wave_sum_lane63(dV[k]);
                const float sp = wave_sum_lane63(dPM[k]);
                if (lane == 63) { wred[wave][k] = sv; wred[wave][16 + k] = sp; }
            }
            if (lane == 63) { wred[wave][32] = 0.f; wred[wave][33] = 0.f; wred[wave][34] = 0.f; }  // (campos: SH loop below)
            __syncthreads();
            if (threadIdx.x < 35)
                pose_acc[((size_t)v * gridDim.x + blockIdx.x) * 64 + threadIdx.x] =
                    wred[0][threadIdx.x] + wred[1][threadIdx.x] + wred[2][threadIdx.x] + wred[3][threadIdx.x];
            __syncthreads();  // (wred is reused by the next view)
        }
    }

    // ---- outputs: sums over the views -------------------------------------------------------------------------
    if (use_sh && KC > 0) {
        constexpr int KCN = KC > 0 ? KC : 1, DEG = KC == 16 ? 3 : 4;
        // ---- (1) view-direction term: dL/dmean += (∂dir/∂mean)ᵀ Jᵀ·dL/dcolour, J = ∂colour/∂dir from the forward ----
        const float4 c0 = recs[(GGR_G2D_STRIDE / 4) * il];  // r, g, b, –
        const bool live = in_range && radii[il] > 0;
        const uint32_t cl = clamped[il];
        const float4 j0 = ggr_ld_f4(reinterpret_cast<const float*>(sh_jac + il)),
                     j1 = ggr_ld_f4(reinterpret_cast<const float*>(sh_jac + jac_plane + il)),
                     j2 = ggr_ld_f4(reinterpret_cast<const float*>(sh_jac + 2 * jac_plane + il));
        float dc[3] = {live && !(cl & 1u) ? c0.x : 0.f, live && !(cl & 2u) ? c0.y : 0.f, live && !(cl & 4u) ? c0.z : 0.f};
        const float in_s = vs.input_scale ? vs.input_scale[0] : 1.0f;
        const float vx = live ? in_s * m0 - vs.campos[0] : 0.f, vy = live ? in_s * m1 - vs.campos[1] : 0.f,
                    vz = live ? in_s * m2 - vs.campos[2] : 1.f;
        const float len = sqrtf(vx * vx + vy * vy + vz * vz);
        float x = vx / len, y = vy / len, z = vz / len;
        // (a culled Gaussian's Jacobian record is not written by the forward: masked here)
        const float ddx = live ? j0.x * dc[0] + j1.x * dc[1] + j2.x * dc[2] : 0.f;
        const float ddy = live ? j0.y * dc[0] + j1.y * dc[1] + j2.y * dc[2] : 0.f;
        const float ddz = live ? j0.z * dc[0] + j1.z * dc[1] + j2.z * dc[2] : 0.f;
        constexpr bool cm = CM;
        float dcam[3] = {0.f, 0.f, 0.f};
        if (live) {
            const float sum2 = vx * vx + vy * vy + vz * vz;
            const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float gx_ = ((sum2 - vx * vx) * ddx - vy * vx * ddy - vz * vx * ddz) * inv32;
            const float gy_ = (-vx * vy * ddx + (sum2 - vy * vy) * ddy - vz * vy * ddz) * inv32;
            const float gz_ = (-vx * vz * ddx - vy * vz * ddy + (sum2 - vz * vz) * ddz) * inv32;
            dmean[0] += in_s * gx_; dmean[1] += in_s * gy_; dmean[2] += in_s * gz_;
            if (POSE) { dcam[0] = -gx_; dcam[1] = -gy_; dcam[2] = -gz_; }
        }
        if (POSE) {  // dL/dcampos partial joins the row the main loop wrote (same threads: 32..34)
            __shared__ float cred3[4][4];
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float scm = wave_sum_lane63(dcam[k]);
                if (lane == 63) cred3[wave][k] = scm;
            }
            __syncthreads();
            if (threadIdx.x >= 32 && threadIdx.x < 35) {
                const int k = threadIdx.x - 32;
                pose_acc[(size_t)blockIdx.x * 64 + threadIdx.x] += cred3[0][k] + cred3[1][k] + cred3[2][k] + cred3[3][k];
            }
        }
        // ---- (2) dL/dSH rows, three row ranges of the block: owners build, everybody copies out ------------------------
        const int rowlen = (int)sh_row;
        const int pk = cm ? 1 : 3, pc = cm ? M : 1;   // coefficient k, channel c sits at k·pk + c·pc
#pragma unroll 1
        for (int R = 0; R < 3; R++) {
            const int r0 = 84 * R, r1 = R == 2 ? 256 : r0 + 84;   // (84 rows: a multiple of 4 → 16-B aligned ranges)
            __syncthreads();  // LDS free (the thirds / the previous range have been read)
            if ((int)threadIdx.x >= r0 && (int)threadIdx.x < r1 && in_range) {
                float* rowp = sh_lds + ((int)threadIdx.x - r0) * rowlen;
#define SH_ROW(k, Bk, bx, by, bz) { const float b_ = (Bk); rowp[(k) * pk] = b_ * dc[0]; rowp[(k) * pk + pc] = b_ * dc[1]; rowp[(k) * pk + 2 * pc] = b_ * dc[2]; }
                GGR_SH_TERMS(SH_ROW, DEG, )
#undef SH_ROW
                for (int k = KCN; k < M; k++) { rowp[k * pk] = 0.f; rowp[k * pk + pc] = 0.f; rowp[k * pk + 2 * pc] = 0.f; }  // bands not evaluated
            }
            __syncthreads();
            const int nrow = min(r1, nG) - r0;
            if (nrow > 0) {
                const int total = nrow * rowlen, n4 = total >> 2;
                float* dst = dL_dsh + (g0 + r0) * sh_row;
                for (int j = threadIdx.x; j < n4; j += 256)
                    ggr_st_f4(dst + 4 * (size_t)j, reinterpret_cast<const float4*>(sh_lds)[j]);
                for (int j = (n4 << 2) + threadIdx.x; j < total; j += 256) ggr_st(dst + j, sh_lds[j]);
            }
        }
    } else if (use_sh) {
        // dL/dSH[k][c] = Σ_views B_k(direction of the view) · dL/dcolour_c(view), one colour channel at a time: 25
        // accumulators instead of 75 (the basis is recomputed per (channel, view) — the kernel is HBM-bound, not
        // VALU-bound — where 75 accumulators carried through the view loop cost 256 VGPRs + spills).  The Gaussian's
        // LDS row then takes the gradient: the coefficients are no longer needed.
        // (1) the view-direction term of every view: dL/dmean += (∂dir/∂mean)ᵀ Σ_k ∇B_k(dir)·(sh_k · dL/dcolour)
        // (loads one view ahead, as above)
        float4 q0 = recs[(GGR_G2D_STRIDE / 4) * il];
        int qrad = radii[il];
        uint32_t qcl = clamped[il];
#pragma clang loop unroll(disable)
        for (int v = 0; v < NV; v++) {
            __asm__ volatile("" ::: "memory");
            const size_t o = (size_t)v * P + il;
            const float4 r0 = q0;  // r, g, b, –
            const bool live = in_range && qrad > 0;
            const uint32_t cl = qcl;
            if (MULTI && v + 1 < NV) { q0 = recs[(GGR_G2D_STRIDE / 4) * (o + P)]; qrad = radii[o + P]; qcl = clamped[o + P]; }
            float dcam[3] = {0.f, 0.f, 0.f};
            if (live) {
            float dc0 = r0.x, dc1 = r0.y, dc2 = r0.z;
            const float in_s = vs.input_scale ? vs.input_scale[v] : 1.0f;
            const float p0 = in_s * m0, p1 = in_s * m1, p2 = in_s * m2;
            const float* campos = vs.campos + 3 * v;
            float dmv[3] = {0.f, 0.f, 0.f};
            {
            if (cl & 1u) dc0 = 0.f;
            if (cl & 2u) dc1 = 0.f;
            if (cl & 4u) dc2 = 0.f;
            const float vx = p0 - campos[0], vy = p1 - campos[1], vz = p2 - campos[2];
            const float len = sqrtf(vx * vx + vy * vy + vz * vz);
            const float x = vx / len, y = vy / len, z = vz / len;
            float* dsh1 = sh_lds + threadIdx.x * sh_stride;  // one view: the Gaussian's gradient row (written out below)
            // the direction term from the forward's Jacobian (no SH row is read) …
            const float4 j0 = ggr_ld_f4(reinterpret_cast<const float*>(sh_jac + o)),
                         j1 = ggr_ld_f4(reinterpret_cast<const float*>(sh_jac + jac_plane + o)),
                         j2 = ggr_ld_f4(reinterpret_cast<const float*>(sh_jac + 2 * jac_plane + o));
            const float ddx = j0.x * dc0 + j1.x * dc1 + j2.x * dc2;
            const float ddy = j0.y * dc0 + j1.y * dc1 + j2.y * dc2;
            const float ddz = j0.z * dc0 + j1.z * dc1 + j2.z * dc2;
            // … and the gradient row: coefficient k, channel c = B_k(direction) · dL/dcolour_c
#define SH_TERM(k, Bk, bx, by, bz)                                                                     \
{                                                                                                  \
    const int o0 = (k) * sh_ks, o1 = o0 + sh_cs, o2 = o1 + sh_cs;                                  \
    dsh1[o0] = (Bk) * dc0; dsh1[o1] = (Bk) * dc1; dsh1[o2] = (Bk) * dc2;                           \
}
            if (!MULTI) GGR_SH_TERMS(SH_TERM, deg, )
#undef SH_TERM
            const float sum2 = vx * vx + vy * vy + vz * vz;
            const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float gx_ = ((sum2 - vx * vx) * ddx - vy * vx * ddy - vz * vx * ddz) * inv32;
            const float gy_ = (-vx * vy * ddx + (sum2 - vy * vy) * ddy - vz * vy * ddz) * inv32;
            const float gz_ = (-vx * vz * ddx - vy * vz * ddy + (sum2 - vz * vz) * ddz) * inv32;
            dmv[0] += gx_; dmv[1] += gy_; dmv[2] += gz_;
            if (POSE) { dcam[0] -= gx_; dcam[1] -= gy_; dcam[2] -= gz_; }

            }
            dmean[0] += in_s * dmv[0]; dmean[1] += in_s * dmv[1]; dmean[2] += in_s * dmv[2];
            }
            if (POSE) {  // this view's dL/dcampos partial joins the row the main loop wrote (same threads: 32..34)
                __shared__ float cred[4][4];
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float scm = wave_sum_lane63(dcam[k]);
                    if (lane == 63) cred[wave][k] = scm;
                }
                __syncthreads();
                if (threadIdx.x >= 32 && threadIdx.x < 35) {
                    const int k = threadIdx.x - 32;
                    pose_acc[((size_t)v * gridDim.x + blockIdx.x) * 64 + threadIdx.x] += cred[0][k] + cred[1][k] + cred[2][k] + cred[3][k];
                }
                __syncthreads();
            }
        }
        float* dsh = sh_lds + threadIdx.x * sh_stride;
        if (!MULTI) {
            // one view: the live rows already hold their gradient; culled Gaussians get zero rows, coefficients of
            // bands that were not evaluated zero gradient
            if (in_range && !(radii[il] > 0)) {
                for (int k = 0; k < (sh_compact ? sh_rowf : sh_flat ? (int)sh_row : copy_row); k++) dsh[k] = 0.f;
            } else if (in_range && !sh_compact) {
                if (inf.sh_channel_major) {
                    for (int c = 0; c < 3; c++)
                        for (int k = K; k < M; k++) dsh[c * M + k] = 0.f;
                } else if (sh_flat) {
                    for (int k = sh_rowf; k < (int)sh_row; k++) dsh[k] = 0.f;
                }
            }
        }
        __syncthreads();
        // coalesced write-out of dL/dSH: the first 3K floats of every row from LDS, the rest zero
        if (sh_compact) {   // (written as whole float4s in address order instead — every line complete, but (row, column)
            // arithmetic per element — the multi-view kernel was slower: C5', 4 views, 417 → 441 µs; not kept)
            write_sh_rows_compact(dL_dsh, sh_lds, g0, nG, M, sh_rowf / 3, sh_stride, inf.sh_channel_major != 0);
        } else if (sh_flat) {
            const size_t total = (size_t)nG * sh_row;
            float* dst = dL_dsh + g0 * sh_row;
            const int n4 = (int)(total >> 2);
            for (int j = threadIdx.x; j < n4; j += blockDim.x)
                ggr_st_f4(dst + 4 * (size_t)j, reinterpret_cast<const float4*>(sh_lds)[j]);
            for (int j = (n4 << 2) + threadIdx.x; j < (int)total; j += blockDim.x) ggr_st(dst + j, sh_lds[j]);
        } else if ((sh_row & 3) == 0 && (sh_rowf & 3) == 0) {
            const int q_row = (int)(sh_row >> 2), q_used = copy_row >> 2;
            for (int j = threadIdx.x; j < nG * q_row; j += blockDim.x) {
                const int g = j / q_row, q = j - g * q_row;
                float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < q_used) {
                    const float* d = sh_lds + g * sh_stride + 4 * q;
                    v4 = make_float4(d[0], d[1], d[2], d[3]);
                }
                ggr_st_f4(dL_dsh + (g0 + g) * sh_row + 4 * q, v4);
            }
        } else {
            const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll 8
            for (int g = wv; g < nG; g += nw)
                for (int k = ln; k < (int)sh_row; k += 64)
                    ggr_st(dL_dsh + (g0 + g) * sh_row + k, k < copy_row ? sh_lds[g * sh_stride + k] : 0.f);
        }
    }
    if (in_range) {
        ggr_st(dL_dopacity + i, dop);
        ggr_st(dL_dmeans3D + 3 * i, dmean[0]); ggr_st(dL_dmeans3D + 3 * i + 1, dmean[1]); ggr_st(dL_dmeans3D + 3 * i + 2, dmean[2]);
        if (has_colors_precomp) { ggr_st(dL_dcolors_precomp + 3 * i, dcp[0]); ggr_st(dL_dcolors_precomp + 3 * i + 1, dcp[1]); ggr_st(dL_dcolors_precomp + 3 * i + 2, dcp[2]); }
        if (scales && dL_dscales) {
            dL_dscales[3 * i] = dsc[0]; dL_dscales[3 * i + 1] = dsc[1]; dL_dscales[3 * i + 2] = dsc[2];
            dL_drotations[4 * i] = drot[0]; dL_drotations[4 * i + 1] = drot[1]; dL_drotations[4 * i + 2] = drot[2];
            dL_drotations[4 * i + 3] = drot[3];
        }
        // (the upper-triangle gather leaves the lower triangle of a [P,3,3] gradient at zero, as autograd does for
        // the reference's fancy index)
        if (cov_is_input && inf.cov_stride == 9) {
            float* d9 = dL_dcov3D + 9 * (size_t)i;
            d9[0] = dcov[0]; d9[1] = dcov[1]; d9[2] = dcov[2];
            d9[3] = 0.f;     d9[4] = dcov[3]; d9[5] = dcov[4];
            d9[6] = 0.f;     d9[7] = 0.f;     d9[8] = dcov[5];
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) ggr_st(dL_dcov3D + 6 * (size_t)i + k, dcov[k]);
        }
    }
}

// The SH backward of a launch set with SEVERAL views per Gaussian set (round 3), in a kernel of its own:
//   dL/dSH[k][c]  = Σ_views B_k(direction of the view) · dL/dcolour_c(view)                       (gradient rows)
//   dL/dmean     += Σ_views (∂dir/∂mean)ᵀ Σ_k ∇B_k(dir) · (sh_k · dL/dcolour(view))               (direction term)
//   dL/dcampos(view) −= the same term, summed over the Gaussians                                   (POSE)
// Inside preprocess_bwd_kernel<·, MULTI> these phases cost that kernel its occupancy (the SH rows in LDS, 25 accumulators
// + 25 basis values on top of the geometry backward's live state: 216-223 VGPRs, 2 waves per SIMD, 2.4 TB/s where the
// one-view kernels reach 4.9; C5', 4 views: 438 µs → main kernel without them + this kernel, see NOTES.md, old §8).  Runs BEHIND
// the main kernel on the same stream: it adds to the dL/dmeans3D and to the camera rows that kernel wrote.
// Reads per (view, Gaussian) the colour part of the blend's gradient record, the radius and the clamp bits; the SH rows
// once, through LDS; writes the gradient rows through the same LDS in three row ranges of the block (whole 128-B lines).
#ifndef GGR_SHV_WAVES
#define GGR_SHV_WAVES 1
#endif
template <int KMAX, bool POSE>
__global__ void __launch_bounds__(256, GGR_SHV_WAVES)
preprocess_bwd_sh_views_kernel(int P, int M, int deg, const float* __restrict__ means3D, const float* __restrict__ shs,
                               const float4* __restrict__ sh_jac, size_t jac_plane, ViewSet vs, const int32_t* __restrict__ radii, const uint32_t* __restrict__ clamped,
                               const float* __restrict__ grad2d, float* __restrict__ dL_dmeans3D,
                               float* __restrict__ dL_dsh, float* __restrict__ pose_acc, InputForm inf) {
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];  // the block's SH rows, then [88][3M] gradient rows
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = i < P;
    const size_t il = (size_t)min(i, P - 1);
    {   // Gaussian set blockIdx.y (as in preprocess_bwd_kernel)
        const int set = (int)blockIdx.y, v0 = set * vs.vps;
        const size_t in_off = (size_t)set * (size_t)P, st_off = (size_t)v0 * (size_t)P;
        means3D += 3 * in_off; shs += in_off * (size_t)M * 3;
        dL_dmeans3D += 3 * in_off; dL_dsh += in_off * (size_t)M * 3;
        radii += st_off; clamped += st_off; grad2d += GGR_G2D_STRIDE * st_off; sh_jac += st_off;
        if (POSE) pose_acc += (size_t)v0 * gridDim.x * 64;
        vs.campos += 3 * v0;
        if (vs.input_scale) vs.input_scale += v0;
    }
    const float m0 = means3D[3 * il], m1 = means3D[3 * il + 1], m2 = means3D[3 * il + 2];
    const int K = (deg + 1) * (deg + 1);
    // staging of the block's rows: the forms of preprocess_bwd_kernel (flat / compact / repacked)
    const size_t g0 = (size_t)blockIdx.x * blockDim.x;
    const int nG = (int)min((size_t)blockDim.x, (size_t)P - g0);
    const size_t sh_row = (size_t)M * 3;
    const int sh_rowf = 3 * K;
    const bool sh_flat = (sh_row & 1) != 0 && inf.sh_aligned != 0;
    const int copy_row = inf.sh_channel_major ? (int)sh_row : sh_rowf;
    const bool sh_compact = (int)sh_row > sh_rowf && sh_row <= 128 && (sh_flat || inf.sh_channel_major);
    const int sh_stride = sh_compact ? (sh_rowf | 1) : sh_flat ? (int)sh_row : (copy_row | 1);
    const int sh_ks = inf.sh_channel_major ? 1 : 3, sh_cs = inf.sh_channel_major ? (sh_compact ? sh_rowf / 3 : M) : 1;
    // (no SH row is read: the direction term comes from the forward's Jacobian, the gradient rows are basis × dL/dcolour;
    //  the LDS only carries the gradient rows out)
    (void)sh_stride; (void)sh_ks; (void)sh_cs; (void)copy_row; (void)sh_compact; (void)sh_flat;

    float acc[3][KMAX];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < KMAX; k++) acc[c][k] = 0.f;
    float dmean[3] = {0.f, 0.f, 0.f};
    const float4* recs = reinterpret_cast<const float4*>(grad2d);
    float4 q0 = recs[(GGR_G2D_STRIDE / 4) * il];
    int qrad = radii[il];
    uint32_t qcl = clamped[il];
    float4 qj0 = sh_jac[il], qj1 = sh_jac[jac_plane + il], qj2 = sh_jac[2 * jac_plane + il];
#pragma clang loop unroll(disable)
    for (int v = 0; v < vs.vps; v++) {
        __asm__ volatile("" ::: "memory");
        const size_t o = (size_t)v * P + il;
        const float4 r0 = q0, j0 = qj0, j1 = qj1, j2 = qj2;
        const bool live = in_range && qrad > 0;
        const uint32_t cl = qcl;
        if (v + 1 < vs.vps) {   // (the per-view loads one view ahead)
            q0 = recs[(GGR_G2D_STRIDE / 4) * (o + P)]; qrad = radii[o + P]; qcl = clamped[o + P];
            qj0 = sh_jac[o + P]; qj1 = sh_jac[jac_plane + o + P]; qj2 = sh_jac[2 * jac_plane + o + P];
        }
        float dcam[3] = {0.f, 0.f, 0.f};
        if (live) {
            const float dc0 = (cl & 1u) ? 0.f : r0.x, dc1 = (cl & 2u) ? 0.f : r0.y, dc2 = (cl & 4u) ? 0.f : r0.z;
            const float in_s = vs.input_scale ? vs.input_scale[v] : 1.0f;
            const float vx = in_s * m0 - vs.campos[3 * v], vy = in_s * m1 - vs.campos[3 * v + 1], vz = in_s * m2 - vs.campos[3 * v + 2];
            const float len = sqrtf(vx * vx + vy * vy + vz * vz);
            float x = vx / len, y = vy / len, z = vz / len;
            // the rows' accumulators from the basis VALUES; the direction term from the forward's Jacobian
            {
                float B[GGR_SH_MAXK];
                sh_basis25(deg, x, y, z, B);
#pragma unroll
                for (int k = 0; k < KMAX; k++)
                    if (k < K) { acc[0][k] += B[k] * dc0; acc[1][k] += B[k] * dc1; acc[2][k] += B[k] * dc2; }
            }
            const float ddx = j0.x * dc0 + j1.x * dc1 + j2.x * dc2;
            const float ddy = j0.y * dc0 + j1.y * dc1 + j2.y * dc2;
            const float ddz = j0.z * dc0 + j1.z * dc1 + j2.z * dc2;
            const float sum2 = vx * vx + vy * vy + vz * vz;
            const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float gx_ = ((sum2 - vx * vx) * ddx - vy * vx * ddy - vz * vx * ddz) * inv32;
            const float gy_ = (-vx * vy * ddx + (sum2 - vy * vy) * ddy - vz * vy * ddz) * inv32;
            const float gz_ = (-vx * vz * ddx - vy * vz * ddy + (sum2 - vz * vz) * ddz) * inv32;
            dmean[0] += in_s * gx_; dmean[1] += in_s * gy_; dmean[2] += in_s * gz_;
            if (POSE) { dcam[0] = -gx_; dcam[1] = -gy_; dcam[2] = -gz_; }
        }
        if (POSE) {  // this view's dL/dcampos partial joins the row the main kernel wrote (entries 32..34)
            __shared__ float cred[4][4];
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float scm = wave_sum_lane63(dcam[k]);
                if (lane == 63) cred[wave][k] = scm;
            }
            __syncthreads();
            if (threadIdx.x >= 32 && threadIdx.x < 35) {
                const int k = threadIdx.x - 32;
                pose_acc[((size_t)v * gridDim.x + blockIdx.x) * 64 + threadIdx.x] += cred[0][k] + cred[1][k] + cred[2][k] + cred[3][k];
            }
            __syncthreads();
        }
    }
    if (in_range) {   // (the main kernel's geometry part of dL/dmean is already there)
        dL_dmeans3D[3 * i] += dmean[0]; dL_dmeans3D[3 * i + 1] += dmean[1]; dL_dmeans3D[3 * i + 2] += dmean[2];
    }
    // rows out: three row ranges of the block through LDS, flat float4 copies (84 rows: a multiple of 4 → 16-B aligned)
    const int rowlen = 3 * M;
    const bool cm = inf.sh_channel_major != 0;
    const int pk = cm ? 1 : 3, pc = cm ? M : 1;   // coefficient k, channel c sits at k·pk + c·pc
#pragma unroll 1
    for (int R = 0; R < 3; R++) {
        const int r0 = 84 * R, r1 = R == 2 ? 256 : r0 + 84;
        __syncthreads();  // the coefficients / the previous range have been read
        if ((int)threadIdx.x >= r0 && (int)threadIdx.x < r1 && in_range) {
            float* rowp = sh_lds + ((int)threadIdx.x - r0) * rowlen;
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if (k < K) { rowp[k * pk] = acc[0][k]; rowp[k * pk + pc] = acc[1][k]; rowp[k * pk + 2 * pc] = acc[2][k]; }
            for (int k = K; k < M; k++) { rowp[k * pk] = 0.f; rowp[k * pk + pc] = 0.f; rowp[k * pk + 2 * pc] = 0.f; }  // bands not evaluated
        }
        __syncthreads();
        const int nrow = min(r1, nG) - r0;
        if (nrow > 0) {
            const int total = nrow * rowlen;
            float* dst = dL_dsh + (g0 + r0) * (size_t)rowlen;
            if (inf.sh_aligned) {   // (g0 + r0 is a multiple of 4 rows: every range starts 16-B aligned)
                const int n4 = total >> 2;
                for (int j = threadIdx.x; j < n4; j += 256)
                    ggr_st_f4(dst + 4 * (size_t)j, reinterpret_cast<const float4*>(sh_lds)[j]);
                for (int j = (n4 << 2) + threadIdx.x; j < total; j += 256) ggr_st(dst + j, sh_lds[j]);
            } else {
                for (int j = threadIdx.x; j < total; j += 256) ggr_st(dst + j, sh_lds[j]);
            }
        }
    }
}

// dL/d(camera of view v)[k] = Σ_blocks pose_acc[(v·nblocks + b)·64 + k]  (one workgroup per (component, view), fixed
// order → deterministic)
__global__ void __launch_bounds__(256)
pose_finish_kernel(const float* __restrict__ pose_acc, int nblocks, float* __restrict__ dL_dview,
                   float* __restrict__ dL_dproj, float* __restrict__ dL_dcampos) {
    __shared__ float sh[256];
    const int k = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    pose_acc += (size_t)v * nblocks * 64;
    float acc = 0.f;
    // sixteen rows per trip, all loads issued before the first add (clamped index, masked add)
    for (int b0 = 0; b0 < nblocks; b0 += 16 * 256) {
        float vv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) vv[u] = pose_acc[(size_t)min(b0 + u * 256 + tid, nblocks - 1) * 64 + k];
#pragma unroll
        for (int u = 0; u < 16; u++) acc += b0 + u * 256 + tid < nblocks ? vv[u] : 0.f;
    }
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) {  // straight into the caller's tensors (was: 3 tiny device-to-device copies = 3 more launches)
        if (k < 16) dL_dview[16 * v + k] = sh[0];
        else if (k < 32) dL_dproj[16 * v + k - 16] = sh[0];
        else dL_dcampos[3 * v + k - 32] = sh[0];
    }
}

void launch_preprocess_bwd(int P, int D, int M, const float* means3D, const float* shs, const float4* sh_jac,
                           int has_colors_precomp, const float* scales, const float* rotations,
                           float scale_modifier, const float* cov3D, ViewSet vs, int W, int H, const int32_t* radii,
                           const uint32_t* clamped,
                           const float* grad2d, int has_dz, float* dL_dmeans3D, float* dL_dmeans2D,
                           float* dL_dopacity, float* dL_dsh,
                           float* dL_dcolors_precomp, float* dL_dcov3D, float* dL_dscales,
                           float* dL_drotations, float* dL_daux, float* pose_acc, float* dL_dview, float* dL_dproj,
                           float* dL_dcampos, InputForm inf, int cov_is_input, hipStream_t s) {
    if (P <= 0) return;
    const int blocks = (P + 255) / 256;
    const int deg = ggr_sh_degree(D, (!has_colors_precomp && shs) ? M : 25, inf.sh_cap);
    const bool flat = ((3 * M) & 1) && inf.sh_aligned;
    const size_t copy_row = inf.sh_channel_major ? (size_t)(3 * M) : (size_t)(3 * (deg + 1) * (deg + 1));
    const size_t rowf = (size_t)(3 * (deg + 1) * (deg + 1));
    const bool compact = (size_t)(3 * M) > rowf && 3 * M <= 128 && (flat || inf.sh_channel_major);
    const size_t row_stride = compact ? (rowf | 1) : flat ? (size_t)(3 * M) : (copy_row | 1);
    const bool multi = vs.vps > 1;
    const bool sh_split = multi && !has_colors_precomp && shs && dL_dsh;   // several views: the SH backward has its own kernel
    // one view at degree 3 / 4 with long rows: rows read by thirds, gradient rows written by row ranges (kernel header)
    const bool use_sh = !has_colors_precomp && shs && !multi;   // (what the MAIN kernel stages)
    const int kc = (use_sh && (deg == 3 || deg == 4) && 3 * M > 64 &&
                    inf.sh_aligned) ? (deg + 1) * (deg + 1) : 0;
    const size_t lds = !use_sh ? 0 : kc ? sizeof(float) * (size_t)std::max(256 * (kc | 1), 88 * 3 * M)
                                        : (size_t)256 * row_stride * sizeof(float);
#define GGR_LAUNCH_PBWD(POSE_, MULTI_, KC_, CM_)                                                                            \
    hipLaunchKernelGGL((preprocess_bwd_kernel<POSE_, MULTI_, KC_, CM_>), dim3(blocks, vs.sets), dim3(256), lds, s, P, D, M, means3D, shs, \
                       sh_jac, (size_t)P * vs.V, has_colors_precomp, scales, rotations, scale_modifier, cov3D, vs, W, H, radii, clamped, grad2d,     \
                       has_dz, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dsh, dL_dcolors_precomp, dL_dcov3D, dL_dscales,   \
                       dL_drotations, dL_daux, pose_acc, inf, cov_is_input)
#define GGR_LAUNCH_PBWD_P(POSE_)                                                                                         \
    do {                                                                                                                  \
        if (multi) GGR_LAUNCH_PBWD(POSE_, true, 0, false);                                                                \
        else if (kc == 16 && inf.sh_channel_major) GGR_LAUNCH_PBWD(POSE_, false, 16, true);                               \
        else if (kc == 16) GGR_LAUNCH_PBWD(POSE_, false, 16, false);                                                      \
        else if (kc == 25 && inf.sh_channel_major) GGR_LAUNCH_PBWD(POSE_, false, 25, true);                               \
        else if (kc == 25) GGR_LAUNCH_PBWD(POSE_, false, 25, false);                                                      \
        else GGR_LAUNCH_PBWD(POSE_, false, 0, false);                                                                     \
    } while (0)
    auto launch_sh_views = [&](bool pose) {   // several views: the SH backward, behind the main kernel (see the kernel)
        const size_t lds_sh = sizeof(float) * std::max((size_t)256 * row_stride, (size_t)88 * 3 * M);
#define GGR_LAUNCH_SHV(K_, POSE_)                                                                                          \
    hipLaunchKernelGGL((preprocess_bwd_sh_views_kernel<K_, POSE_>), dim3(blocks, vs.sets), dim3(256), lds_sh, s, P, M, deg,  \
                       means3D, shs, sh_jac, (size_t)P * vs.V, vs, radii, clamped, grad2d, dL_dmeans3D, dL_dsh, pose_acc, inf)
        if (deg >= 4) { if (pose) GGR_LAUNCH_SHV(25, true); else GGR_LAUNCH_SHV(25, false); }
        else { if (pose) GGR_LAUNCH_SHV(16, true); else GGR_LAUNCH_SHV(16, false); }
#undef GGR_LAUNCH_SHV
    };
    if (pose_acc) {
        GGR_LAUNCH_PBWD_P(true);
        if (sh_split) launch_sh_views(true);
        hipLaunchKernelGGL(pose_finish_kernel, dim3(35, vs.V), dim3(256), 0, s, pose_acc, blocks, dL_dview, dL_dproj,
                           dL_dcampos);
    } else {
        GGR_LAUNCH_PBWD_P(false);
        if (sh_split) launch_sh_views(false);
    }
#undef GGR_LAUNCH_PBWD_P
#undef GGR_LAUNCH_PBWD
}

}  // namespace ggr
