// blend_common.h — pieces shared by the forward and backward blend kernels (gfx950).
#pragma once
#include "ggr_common.h"

namespace ggr {

#define GGR_BATCH 256

// Workgroup b runs on XCD b mod 8 and every XCD has its own L2.  Two tile ↔ workgroup mappings:
//   interleaved  tile = b: every 8th tile of the row-major sequence per XCD.  Any compact region of the frame — the one
//                live cell of the fine-tune loop's deferred back-propagation, the ground under a sky — is spread over
//                all eight XCDs;
//   ranged       XCD x owns the contiguous tile range [x·len, (x+1)·len): neighbouring tiles share most of their
//                Gaussians' 48-B records, which then meet in one L2.
// Measured (round 3, blend forward / backward, ranged → interleaved): C3 0.174 / 0.394 → 0.173 / 0.389 ms; C3 with the
// upper half of the frame empty 0.145 / 0.330 → 0.118 / 0.246 (rounds 1-2 split the ranges in up to four per XCD for this
// frame: 0.160 / 0.349 at the time); C5′ 0.186 / 0.246 → 0.192-0.195 / 0.246-0.248; C4′ 0.199 / 0.230 → 0.205 / 0.231;
// C6′ 0.433 / 0.841 → 0.440 / 0.825; C5′ with the gradient confined to one cell of 2 × 2: backward 0.155 → 0.103, the
// scissored forward 0.436 → 0.397.  Hence: the backward is always interleaved; the forward is ranged only for a frame of
// fewer than 4096 tiles without a scissor (its one regression, 3-5 %).  Both return -1 for the padding workgroups of a
// grid of xcd_grid(tiles).
__host__ __device__ static inline int xcd_grid(int tiles) { return ((tiles + 7) / 8) * 8; }
__device__ __forceinline__ int xcd_tile(int block, int tiles, bool interleaved) {
    if (interleaved) return block < tiles ? block : -1;
    const int len = (tiles + 7) / 8;
    const int t = (block & 7) * len + (block >> 3);
    return t < tiles ? t : -1;
}
__host__ static inline bool xcd_forward_interleaved(int tiles, bool scissored) { return tiles >= 4096 || scissored; }

// Checkpoints (ggr_common.h, ImageLayout): list positions between two checkpoints of a tile whose list has `len`
// entries — a multiple of the staging batch, large enough that slots 1 … slots−1 cover the whole list.
// Images with ≥ 1024 tiles already bring ≥ 4 waves per SIMD: there a segment is at least two batches, so that
// short lists are not cut into pieces that mostly pay the workgroup prologue.
__device__ __forceinline__ int ckpt_stride(int len, int slots, int ntiles) {
    return GGR_BATCH * max(ntiles >= 1024 ? 2 : 1, (len + slots * GGR_BATCH - 1) / (slots * GGR_BATCH));
}

// One staged list entry in LDS: 3 × 16 B, read back as wave-uniform (broadcast) ds_read_b128.
struct __attribute__((aligned(16))) StagedSplat {
    float4 a;  // x, y, conic.xx, conic.xy
    float4 b;  // conic.yy, opacity, r, g
    float4 c;  // b, z, qmax = 2·ln(255·opacity), id (bits)
};

// [budget: cull]
// Exact minimum of q(d) = cxx·dx² + 2·cxy·dx·dy + cyy·dy² (d = mean − pixel) over the pixel box
// [x0,x1]×[y0,y1], i.e. over d ∈ [dxl,dxh]×[dyl,dyh].  q is convex (the conic is positive definite) with its free
// minimum at the mean (d = 0): the minimum over the box is 0 if the mean lies in it and otherwise sits on a face of the
// box that the mean SEES (moving from any other boundary point towards the mean stays inside the box and lowers q) —
// at most one vertical and one horizontal face, where q is a clamped 1-D quadratic.  With f = the box point nearest to
// the mean (v_med3 per coordinate; 0 where the mean is inside the range) the two candidates are the lines dx = f.x and
// dy = f.y restricted to the box: the faces the mean sees, or — where it sees none — a line through box points, whose
// values cannot undercut the minimum.  No compare, no branch (round 3: four edges, an inside test of four compares at
// 4.6 cycles each and a branch cost ≈ 60 instructions per test; this form ≈ 27 — tools/valu_peak_bench.hip).
__device__ __forceinline__ float box_min_q(float mx, float my, float cxx, float cxy, float cyy, float x0,
                                           float y0, float x1, float y1) {
    const float dxl = mx - x1, dxh = mx - x0, dyl = my - y1, dyh = my - y0;
    const float fx = __builtin_amdgcn_fmed3f(0.f, dxl, dxh), fy = __builtin_amdgcn_fmed3f(0.f, dyl, dyh);
    // v_rcp_f32 (1 ulp) instead of two IEEE divisions: an edge minimiser that is off by 1e-7 relative changes q
    // only to second order, far inside the caller's 1e-3 margin
    const float ry = -cxy * __builtin_amdgcn_rcpf(cyy), rx = -cxy * __builtin_amdgcn_rcpf(cxx);
    auto qf = [&](float dx, float dy) { return cxx * dx * dx + 2.f * cxy * dx * dy + cyy * dy * dy; };
    const float qa = qf(fx, __builtin_amdgcn_fmed3f(ry * fx, dyl, dyh));
    const float qb = qf(__builtin_amdgcn_fmed3f(rx * fy, dxl, dxh), fy);
    return fminf(qa, qb);
}

// Can this entry reach α ≥ 1/255 on any pixel of the box?  α = opacity·exp(−q/2) ≥ 1/255 ⇔
// q ≤ qmax = 2·ln(255·opacity).  Conservative by a 1e-3 relative + absolute margin, i.e. an entry is
// dropped only if every pixel of the box would take the reference's `α < 1/255 → continue` branch.
__device__ __forceinline__ bool box_may_contribute(const float4 a, const float4 b, float qmax, float x0,
                                                   float y0, float x1, float y1) {
    const float qmin = box_min_q(a.x, a.y, a.z, a.w, b.x, x0, y0, x1, y1);
    return qmin * 0.999f <= qmax + 1e-3f;
}

// [budget: evaluate]
// ---- staged form of the conic ---------------------------------------------------------------------------------
// The blend kernels evaluate G = exp(−q/2), q = cxx·dx² + 2·cxy·dx·dy + cyy·dy², once per (entry, pixel).  The
// thread that stages an entry pre-multiplies the conic by k = log2(e)/2, so that the pixel loop needs
//     u = fma(sxx, dx, sxy2·dy);  q2 = fma(syy·dy, dy, u·dx);  G = v_exp_f32(−q2)          (5 ops + exp)
// instead of the reference's −½(cxx·dx² + cyy·dy²) − cxy·dx·dy followed by the ·log2(e) of expf (8 ops + exp).
// `power > 0` (reference: skip) ⇔ q2 < 0.  The record in HBM keeps the plain conic (bit-exact with the oracle).
#define GGR_KQ 0.72134752f         // log2(e) / 2
#define GGR_INV_KQ 1.38629436f     // 2·ln 2
__device__ __forceinline__ void stage_scale_conic(float4& a, float4& b, float4& c) {
    a.z *= GGR_KQ;          // sxx  = k·cxx
    a.w *= 2.f * GGR_KQ;    // sxy2 = 2k·cxy
    b.x *= GGR_KQ;          // syy  = k·cyy
    c.z *= GGR_KQ;          // k·qmax
}
__device__ __forceinline__ float staged_q2(const float4 a, const float4 b, float dx, float dy) {
    const float u = fmaf(a.z, dx, a.w * dy);
    return fmaf(b.x * dy, dy, u * dx);
}
// [budget: cull]
// box_may_contribute on a staged entry (the test is homogeneous in the conic, so it runs on k·q directly)
__device__ __forceinline__ bool staged_box_may_contribute(const float4 a, const float4 b, float kqmax, float x0,
                                                          float y0, float x1, float y1) {
    const float qmin = box_min_q(a.x, a.y, a.z, 0.5f * a.w, b.x, x0, y0, x1, y1);
    return qmin * 0.999f <= kqmax + 1e-3f;
}

// Bounding box of the ACTIVE pixels of a quadrant (lane = 8·row + column; `act` = ballot of the lanes that can still take an
// entry of the batch): entries that reach no active pixel are culled as well — exact, an inactive pixel takes nothing.
// Scalar bit arithmetic on the 64-bit mask (≈ 20 scalar instructions per batch and wave).
__device__ __forceinline__ void active_box(uint64_t act, float qx0, float qy0, float& x0, float& y0, float& x1, float& y1) {
    uint32_t cols = (uint32_t)(act | (act >> 32));
    cols |= cols >> 16; cols |= cols >> 8; cols &= 0xFFu;
    uint64_t rows = act;
    rows |= rows >> 1; rows |= rows >> 2; rows |= rows >> 4; rows &= 0x0101010101010101ull;
    const int c0 = __builtin_ctz(cols), c1 = 31 - __builtin_clz(cols);
    const int r0 = __builtin_ctzll(rows) >> 3, r1 = (63 - __builtin_clzll(rows)) >> 3;
    x0 = qx0 + (float)c0; x1 = qx0 + (float)c1; y0 = qy0 + (float)r0; y1 = qy0 + (float)r1;
}

}  // namespace ggr
