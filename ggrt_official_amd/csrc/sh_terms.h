// sh_terms.h — the real SH basis of degree ≤ 4 and its gradient w.r.t. the (unit) direction, as a term generator.
//
// Used by the per-Gaussian backward (preprocess_bwd.hip: gradient rows from the basis values) and by the forward's colour
// evaluation when it leaves the 3×3 Jacobian ∂colour/∂direction for the backward (preprocess.hip).  Sign convention and
// constants of the rasterizer family behind reference cuda_splatting.py:114-125; band 4: the
// standard real basis (oracle/ggr_oracle.c header).  Every translation unit gets its own copy of the constants (no RDC).
#pragma once
#include <hip/hip_runtime.h>

namespace ggr {

#ifndef SH_C0
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#endif
static __device__ __constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
static __device__ __constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};

static __device__ __constant__ float bSH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                                           -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                                           0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};


// The SH terms of a direction (x, y, z): T(k, B_k, ∂B_k/∂x, ∂B_k/∂y, ∂B_k/∂z) for every coefficient k of degree ≤ deg, in
// the rasterizer's sign convention (band 4: oracle/ggr_oracle.c header, plain polynomial derivatives).  Needs x, y, z
// in scope; unused values fold away.  F: statement placed in front of every band (a scheduling fence, or nothing).
#define GGR_SH_TERMS(T, deg, F)                                                                                            \
    {                                                                                                                     \
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;                                    \
        T(0, SH_C0, 0.f, 0.f, 0.f)                                                                                        \
        if ((deg) > 0) {                                                                                                  \
            F                                                                                                             \
            T(1, -SH_C1 * y, 0.f, -SH_C1, 0.f)                                                                            \
            T(2, SH_C1 * z, 0.f, 0.f, SH_C1)                                                                              \
            T(3, -SH_C1 * x, -SH_C1, 0.f, 0.f)                                                                            \
            if ((deg) > 1) {                                                                                              \
                F                                                                                                         \
                T(4, bSH_C2[0] * xy, bSH_C2[0] * y, bSH_C2[0] * x, 0.f)                                                   \
                T(5, bSH_C2[1] * yz, 0.f, bSH_C2[1] * z, bSH_C2[1] * y)                                                   \
                T(6, bSH_C2[2] * (2.f * zz - xx - yy), bSH_C2[2] * -2.f * x, bSH_C2[2] * -2.f * y, bSH_C2[2] * 4.f * z)  \
                T(7, bSH_C2[3] * xz, bSH_C2[3] * z, 0.f, bSH_C2[3] * x)                                                   \
                T(8, bSH_C2[4] * (xx - yy), bSH_C2[4] * 2.f * x, bSH_C2[4] * -2.f * y, 0.f)                               \
                if ((deg) > 2) {                                                                                          \
                    F                                                                                                     \
                    T(9, bSH_C3[0] * y * (3.f * xx - yy), bSH_C3[0] * 6.f * xy, bSH_C3[0] * 3.f * (xx - yy), 0.f)         \
                    T(10, bSH_C3[1] * xy * z, bSH_C3[1] * yz, bSH_C3[1] * xz, bSH_C3[1] * xy)                             \
                    T(11, bSH_C3[2] * y * (4.f * zz - xx - yy), bSH_C3[2] * -2.f * xy,                                    \
                      bSH_C3[2] * (-3.f * yy + 4.f * zz - xx), bSH_C3[2] * 8.f * yz)                                      \
                    T(12, bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), bSH_C3[3] * -6.f * xz,                        \
                      bSH_C3[3] * -6.f * yz, bSH_C3[3] * 3.f * (2.f * zz - xx - yy))                                      \
                    F                                                                                                     \
                    T(13, bSH_C3[4] * x * (4.f * zz - xx - yy), bSH_C3[4] * (-3.f * xx + 4.f * zz - yy),                  \
                      bSH_C3[4] * -2.f * xy, bSH_C3[4] * 8.f * xz)                                                        \
                    T(14, bSH_C3[5] * z * (xx - yy), bSH_C3[5] * 2.f * xz, bSH_C3[5] * -2.f * yz, bSH_C3[5] * (xx - yy))  \
                    T(15, bSH_C3[6] * x * (xx - 3.f * yy), bSH_C3[6] * 3.f * (xx - yy), bSH_C3[6] * -6.f * xy, 0.f)       \
                    if ((deg) > 3) {                                                                                      \
                        F                                                                                                 \
                        const float a7 = 7.f * zz - 1.f, b7 = 7.f * zz - 3.f; [[maybe_unused]] const float c21 = 21.f * zz - 3.f;                      \
                        const float xmy = xx - yy, x3y = xx - 3.f * yy, y3x = 3.f * xx - yy;                              \
                        T(16, bSH_C4[0] * xy * xmy, bSH_C4[0] * y * y3x, bSH_C4[0] * x * x3y, 0.f)                        \
                        T(17, bSH_C4[1] * yz * y3x, bSH_C4[1] * 6.f * xy * z, bSH_C4[1] * 3.f * z * xmy, bSH_C4[1] * y * y3x) \
                        T(18, bSH_C4[2] * xy * a7, bSH_C4[2] * y * a7, bSH_C4[2] * x * a7, bSH_C4[2] * 14.f * xy * z)     \
                        T(19, bSH_C4[3] * yz * b7, 0.f, bSH_C4[3] * z * b7, bSH_C4[3] * y * c21)                          \
                        T(20, bSH_C4[4] * (zz * (35.f * zz - 30.f) + 3.f), 0.f, 0.f, bSH_C4[4] * z * (140.f * zz - 60.f)) \
                        F                                                                                                 \
                        T(21, bSH_C4[5] * xz * b7, bSH_C4[5] * z * b7, 0.f, bSH_C4[5] * x * c21)                          \
                        T(22, bSH_C4[6] * xmy * a7, bSH_C4[6] * 2.f * x * a7, bSH_C4[6] * -2.f * y * a7, bSH_C4[6] * 14.f * z * xmy) \
                        T(23, bSH_C4[7] * xz * x3y, bSH_C4[7] * 3.f * z * xmy, bSH_C4[7] * -6.f * xy * z, bSH_C4[7] * x * x3y) \
                        T(24, bSH_C4[8] * (xx * x3y - yy * y3x), bSH_C4[8] * 4.f * x * x3y, bSH_C4[8] * -4.f * y * y3x, 0.f) \
                    }                                                                                                     \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
    }


}  // namespace ggr
