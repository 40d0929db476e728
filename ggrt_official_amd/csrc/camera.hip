// camera.hip — per-view camera quantities of the call site in ONE small launch (gfx950).
//
// Reference cuda_splatting.py:66-73 (1/near renormalisation), :82-84 + ggrt/geometry/projection.py:233-247
// (fov from the normalised intrinsics), :18-46 (GGRt's projection matrix — built from intrinsics[0] for EVERY
// view), :86-89 (view = inverse(extrinsics)ᵀ, full = view @ projectionᵀ) run ≈ 40 tiny torch kernels plus two
// blocking copies (four host→device constants, tan(fov/2) back to the host) before the first rasterizer call.
// Here: one thread per view, fp64 inside, results rounded to fp32 once; tan(fov/2) and 1/near stay on the
// device (GgrSettings.tanfov_dev, GgrForwardIn.input_scale), so the whole call site can run without a host sync.
#include "ggr_common.h"

namespace ggr {

__device__ __forceinline__ bool invert4(const double* m, double* inv) {
    // Gauss-Jordan with partial pivoting on [m | I]
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { a[i][j] = m[4 * i + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; j++) { const double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        const double d = 1.0 / a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] *= d;
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.0) for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) inv[4 * i + j] = a[i][4 + j];
    return true;
}

__device__ __forceinline__ void k_inv_ray(const double* K, double u, double v, double* d) {
    // d = normalise(K⁻¹ · (u, v, 1)), K 3×3 (adjugate form; any non-singular K)
    const double c00 = K[4] * K[8] - K[5] * K[7], c01 = K[2] * K[7] - K[1] * K[8], c02 = K[1] * K[5] - K[2] * K[4];
    const double c10 = K[5] * K[6] - K[3] * K[8], c11 = K[0] * K[8] - K[2] * K[6], c12 = K[2] * K[3] - K[0] * K[5];
    const double c20 = K[3] * K[7] - K[4] * K[6], c21 = K[1] * K[6] - K[0] * K[7], c22 = K[0] * K[4] - K[1] * K[3];
    const double det = K[0] * c00 + K[1] * c10 + K[2] * c20;
    double x = (c00 * u + c01 * v + c02) / det, y = (c10 * u + c11 * v + c12) / det, z = (c20 * u + c21 * v + c22) / det;
    const double n = sqrt(x * x + y * y + z * z);
    d[0] = x / n; d[1] = y / n; d[2] = z / n;
}

__global__ void camera_setup_kernel(int n, const float* __restrict__ extrinsics /*[n,4,4] camera-to-world*/,
                                    const float* __restrict__ intrinsics /*[n,3,3] normalised*/,
                                    const float* __restrict__ near, const float* __restrict__ far,
                                    int scale_invariant, float* __restrict__ view /*[n,16]*/,
                                    float* __restrict__ full /*[n,16]*/, float* __restrict__ campos /*[n,3]*/,
                                    float* __restrict__ tanfov /*[n,2]*/, float* __restrict__ scale /*[n]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the reference multiplies in fp32 (scale = 1/near; t·scale; near·scale; far·scale): keep those roundings
    const float s = scale_invariant ? 1.0f / near[i] : 1.0f;
    scale[i] = s;
    const float nr = scale_invariant ? near[i] * s : near[i], fr = scale_invariant ? far[i] * s : far[i];
    double E[16], Ei[16];
    for (int k = 0; k < 16; k++) E[k] = (double)extrinsics[16 * (size_t)i + k];
    for (int r = 0; r < 3; r++) {
        const float t = scale_invariant ? extrinsics[16 * (size_t)i + 4 * r + 3] * s : extrinsics[16 * (size_t)i + 4 * r + 3];
        E[4 * r + 3] = (double)t;
        campos[3 * i + r] = t;
    }
    if (!invert4(E, Ei)) for (int k = 0; k < 16; k++) Ei[k] = nan("");
    // view = inverse(extrinsics)ᵀ  (row-vector convention of the rasterizer)
    double V[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) V[4 * r + c] = Ei[4 * c + r];
    for (int k = 0; k < 16; k++) view[16 * (size_t)i + k] = (float)V[k];
    // fov: angle between the rays through the mid-points of opposite image edges
    double K[9], l[3], r_[3], t_[3], b_[3];
    for (int k = 0; k < 9; k++) K[k] = (double)intrinsics[9 * (size_t)i + k];
    k_inv_ray(K, 0.0, 0.5, l); k_inv_ray(K, 1.0, 0.5, r_); k_inv_ray(K, 0.5, 0.0, t_); k_inv_ray(K, 0.5, 1.0, b_);
    const double fovx = acos(fmin(1.0, fmax(-1.0, l[0] * r_[0] + l[1] * r_[1] + l[2] * r_[2])));
    const double fovy = acos(fmin(1.0, fmax(-1.0, t_[0] * b_[0] + t_[1] * b_[1] + t_[2] * b_[2])));
    tanfov[2 * i] = (float)tan(0.5 * fovx);
    tanfov[2 * i + 1] = (float)tan(0.5 * fovy);
    // GGRt's projection (cuda_splatting.py:18-46): X/Y rows from intrinsics[0] for every view, fp32 products
    const float k00 = intrinsics[0], k11 = intrinsics[4], k02 = intrinsics[2], k12 = intrinsics[5];
    double Pm[16] = {0};
    Pm[0] = (double)(2.0f * nr * k00);
    Pm[5] = (double)(2.0f * nr * k11);
    Pm[2] = (double)(2.0f * k02 - 1.0f);
    Pm[6] = (double)(2.0f * k12 - 1.0f);
    Pm[14] = 1.0;
    Pm[10] = (double)(fr / (fr - nr));
    Pm[11] = (double)(-(fr * nr) / (fr - nr));
    // full = view @ Pmᵀ
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double a = 0.0;
            for (int k = 0; k < 4; k++) a += V[4 * r + k] * Pm[4 * c + k];
            full[16 * (size_t)i + 4 * r + c] = (float)a;
        }
}

void launch_camera_setup(int n, const float* extrinsics, const float* intrinsics, const float* near, const float* far,
                         int scale_invariant, float* view, float* full, float* campos, float* tanfov, float* scale,
                         hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(camera_setup_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, extrinsics, intrinsics, near, far,
                       scale_invariant, view, full, campos, tanfov, scale);
}

}  // namespace ggr
