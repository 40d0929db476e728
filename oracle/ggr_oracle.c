/*
 * ggr_oracle.c — TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline).
 *
 * A plain-C, sequential, scalar restatement of the differentiable 3D-Gaussian
 * tile rasterizer that GGRt calls at
 *   /root/reference/ggrt/model/pixelsplat/decoder/cuda_splatting.py:101-125
 * (`GaussianRasterizer(settings)(means3D=…, means2D=…, shs=…, opacities=…,
 *   cov3D_precomp=…)`).
 *
 * PARITY UNPINNED.  The arithmetic of that call lives in a third-party CUDA
 * extension (`diff_gaussian_rasterization`, README.md:17-18 of the reference:
 * dcharatan/diff-gaussian-rasterization-modified, a fork of
 * graphdeco-inria/diff-gaussian-rasterization) which is NOT vendored, NOT
 * version-pinned, and absent from /root/reference; the reference holds no
 * tests or golden vectors for it.  This file restates the
 * published algorithm of that rasterizer family:
 *   A.1 preprocess   (projection, EWA cov2D, conic, radius, tile rect, SH→RGB)
 *   A.2 binning      (64-bit key = tile<<32 | depth bits, stable sort, ranges)
 *   A.3 forward      (per-pixel front-to-back alpha compositing, exact skip /
 *                     stop rules)
 *   A.4 backward     (back-to-front replay, analytic gradients, incl. the
 *                     documented quirks A.5: unclamped-α gradient, frozen
 *                     frustum clamp, 1e-7 epsilons)
 * and is itself pinned by (a) known-answer tests, (b) an independent PyTorch
 * autograd restatement (oracle/torch_raster.py) and (c) finite differences —
 * see tests/test_oracle_*.py.
 *
 * Nothing under ggrt_official_amd/ (the product) may include, link, import or
 * call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * SH degree: the graphdeco rasterizer and its "w-depth" forks evaluate bands 0..3 only.  GGRt calls with
 * sh_degree = 4 and 25 coefficients (cuda_splatting.py:75-77, encoder config d_sh = 25).  Its README names
 * dcharatan's fork, but the LIVE call site unpacks a 3-tuple and builds the settings without a `debug` field
 * (cuda_splatting.py:101-118) — the signature of the graphdeco-era w-depth family, not of dcharatan's
 * (2-tuple, `debug` required) — so `sh_cap` = 3 (coefficients 16.. ignored, zero gradient) is the default.
 * `sh_cap` = 4 (band 4 evaluated when D ≥ 4 and M ≥ 25) is restated as well, for a host whose rasterizer
 * does evaluate it; neither can be verified against the installed extension here.  The
 * degree-4 basis is the standard real SH polynomial set (PlenOctree/svox2 `SH_C4`), its gradient the plain
 * polynomial derivative.
 *
 * Build: see oracle/Makefile   (gcc -O2 -ffp-contract=off -fopenmp → libggr_oracle.so)
 * Arithmetic: fp32 with explicit operation order, no FMA contraction; gradient
 * sums over (pixel, Gaussian) pairs are accumulated in fp64 so the oracle is
 * the *more* accurate side of every comparison.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GGO_TILE 16
#define GGO_NEAR_CULL 0.2f
#define GGO_DILATION 0.3f
#define GGO_FRUSTUM_CLAMP 1.3f
#define GGO_ALPHA_MIN (1.0f / 255.0f)
#define GGO_ALPHA_MAX 0.99f
#define GGO_T_MIN 0.0001f

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static const float SH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                               -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                               0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

/* bands actually evaluated: min(D, cap), and never more than the row holds */
static int sh_eff_degree(int D, int M, int cap) {
    int deg = D < cap ? D : cap;
    if (deg < 0) deg = 0;
    while (deg > 0 && (deg + 1) * (deg + 1) > M) deg--;
    return deg;
}

/* Row-vector convention (Appendix A.0): p' = [x y z 1] @ M, M row-major 4x4. */
static void xform4x3(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static void get_rect(float px, float py, int radius, int gx, int gy, int* rmin, int* rmax) {
    rmin[0] = imin(gx, imax(0, (int)((px - (float)radius) / (float)GGO_TILE)));
    rmin[1] = imin(gy, imax(0, (int)((py - (float)radius) / (float)GGO_TILE)));
    rmax[0] = imin(gx, imax(0, (int)((px + (float)radius + (float)(GGO_TILE - 1)) / (float)GGO_TILE)));
    rmax[1] = imin(gy, imax(0, (int)((py + (float)radius + (float)(GGO_TILE - 1)) / (float)GGO_TILE)));
}

/* Σ = R S Sᵀ Rᵀ from scale + (un-normalised) quaternion (r,x,y,z); A.1 scale/rot path. */
static void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* cov6) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                  2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                  2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
    float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
    /* M = R * diag(sc); Sigma = M Mᵀ */
    float Mx[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Mx[3 * i + j] = R[3 * i + j] * sc[j];
    float S[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float a = 0.f;
            for (int k = 0; k < 3; k++) a += Mx[3 * i + k] * Mx[3 * j + k];
            S[3 * i + j] = a;
        }
    cov6[0] = S[0]; cov6[1] = S[1]; cov6[2] = S[2]; cov6[3] = S[4]; cov6[4] = S[5]; cov6[5] = S[8];
}

/* EWA projection pieces shared by forward and backward (A.1 step 3). */
typedef struct {
    float t[3];      /* view-space mean with clamped x,y */
    float A[6];      /* 2x3 matrix J·R, row-major */
    float a, b, c;   /* dilated 2D covariance */
    float xmul, ymul;/* 0 where the frustum clamp bit (A.5 item 2) */
} Ewa;

static void ewa_project(const float* mean, const float* cov6, const float* V, float fx, float fy,
                        float tanfovx, float tanfovy, Ewa* e) {
    float t[3];
    xform4x3(mean, V, t);
    const float limx = GGO_FRUSTUM_CLAMP * tanfovx, limy = GGO_FRUSTUM_CLAMP * tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    e->xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e->ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    e->t[0] = t[0]; e->t[1] = t[1]; e->t[2] = t[2];
    /* J (2x3) */
    const float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* R = world->view rotation, R[i][j] = V[4*j+i] */
    float R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = V[4 * j + i];
    for (int j = 0; j < 3; j++) {
        e->A[j] = J00 * R[j] + J02 * R[6 + j];
        e->A[3 + j] = J11 * R[3 + j] + J12 * R[6 + j];
    }
    const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
    float AS[6];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            float acc = 0.f;
            for (int k = 0; k < 3; k++) acc += e->A[3 * i + k] * S[3 * k + j];
            AS[3 * i + j] = acc;
        }
    float c00 = 0.f, c01 = 0.f, c11 = 0.f;
    for (int k = 0; k < 3; k++) {
        c00 += AS[k] * e->A[k];
        c01 += AS[k] * e->A[3 + k];
        c11 += AS[3 + k] * e->A[3 + k];
    }
    e->a = c00 + GGO_DILATION;
    e->b = c01;
    e->c = c11 + GGO_DILATION;
}

/* SH basis values for unit direction d, K = (deg+1)^2 entries, deg ≤ 4 (A.1 step 7). */
static void sh_basis(int deg, const float* d, float* B) {
    float x = d[0], y = d[1], z = d[2];
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3[0] * y * (3.0f * xx - yy);
                B[10] = SH_C3[1] * xy * z;
                B[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
                B[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
                B[14] = SH_C3[5] * z * (xx - yy);
                B[15] = SH_C3[6] * x * (xx - 3.0f * yy);
                if (deg > 3) {
                    B[16] = SH_C4[0] * xy * (xx - yy);
                    B[17] = SH_C4[1] * yz * (3.0f * xx - yy);
                    B[18] = SH_C4[2] * xy * (7.0f * zz - 1.0f);
                    B[19] = SH_C4[3] * yz * (7.0f * zz - 3.0f);
                    B[20] = SH_C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
                    B[21] = SH_C4[5] * xz * (7.0f * zz - 3.0f);
                    B[22] = SH_C4[6] * (xx - yy) * (7.0f * zz - 1.0f);
                    B[23] = SH_C4[7] * xz * (xx - 3.0f * yy);
                    B[24] = SH_C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
                }
            }
        }
    }
}
/* ∂B_k/∂(x,y,z) for the same basis. */
static void sh_basis_grad(int deg, const float* d, float* Bx, float* By, float* Bz) {
    float x = d[0], y = d[1], z = d[2];
    for (int k = 0; k < 25; k++) Bx[k] = By[k] = Bz[k] = 0.f;
    if (deg > 0) {
        By[1] = -SH_C1; Bz[2] = SH_C1; Bx[3] = -SH_C1;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Bx[4] = SH_C2[0] * y; By[4] = SH_C2[0] * x;
            By[5] = SH_C2[1] * z; Bz[5] = SH_C2[1] * y;
            Bx[6] = SH_C2[2] * -2.f * x; By[6] = SH_C2[2] * -2.f * y; Bz[6] = SH_C2[2] * 4.f * z;
            Bx[7] = SH_C2[3] * z; Bz[7] = SH_C2[3] * x;
            Bx[8] = SH_C2[4] * 2.f * x; By[8] = SH_C2[4] * -2.f * y;
            if (deg > 2) {
                Bx[9] = SH_C3[0] * 6.f * xy;            By[9] = SH_C3[0] * 3.f * (xx - yy);
                Bx[10] = SH_C3[1] * yz;                 By[10] = SH_C3[1] * xz;  Bz[10] = SH_C3[1] * xy;
                Bx[11] = SH_C3[2] * -2.f * xy;          By[11] = SH_C3[2] * (-3.f * yy + 4.f * zz - xx);
                Bz[11] = SH_C3[2] * 8.f * yz;
                Bx[12] = SH_C3[3] * -6.f * xz;          By[12] = SH_C3[3] * -6.f * yz;
                Bz[12] = SH_C3[3] * 3.f * (2.f * zz - xx - yy);
                Bx[13] = SH_C3[4] * (-3.f * xx + 4.f * zz - yy); By[13] = SH_C3[4] * -2.f * xy;
                Bz[13] = SH_C3[4] * 8.f * xz;
                Bx[14] = SH_C3[5] * 2.f * xz;           By[14] = SH_C3[5] * -2.f * yz;
                Bz[14] = SH_C3[5] * (xx - yy);
                Bx[15] = SH_C3[6] * 3.f * (xx - yy);    By[15] = SH_C3[6] * -6.f * xy;
                if (deg > 3) {
                    Bx[16] = SH_C4[0] * y * (3.f * xx - yy);   By[16] = SH_C4[0] * x * (xx - 3.f * yy);
                    Bx[17] = SH_C4[1] * 6.f * xy * z;          By[17] = SH_C4[1] * 3.f * z * (xx - yy);
                    Bz[17] = SH_C4[1] * y * (3.f * xx - yy);
                    Bx[18] = SH_C4[2] * y * (7.f * zz - 1.f);  By[18] = SH_C4[2] * x * (7.f * zz - 1.f);
                    Bz[18] = SH_C4[2] * 14.f * xy * z;
                    By[19] = SH_C4[3] * z * (7.f * zz - 3.f);  Bz[19] = SH_C4[3] * y * (21.f * zz - 3.f);
                    Bz[20] = SH_C4[4] * z * (140.f * zz - 60.f);
                    Bx[21] = SH_C4[5] * z * (7.f * zz - 3.f);  Bz[21] = SH_C4[5] * x * (21.f * zz - 3.f);
                    Bx[22] = SH_C4[6] * 2.f * x * (7.f * zz - 1.f); By[22] = SH_C4[6] * -2.f * y * (7.f * zz - 1.f);
                    Bz[22] = SH_C4[6] * 14.f * z * (xx - yy);
                    Bx[23] = SH_C4[7] * 3.f * z * (xx - yy);   By[23] = SH_C4[7] * -6.f * xy * z;
                    Bz[23] = SH_C4[7] * x * (xx - 3.f * yy);
                    Bx[24] = SH_C4[8] * 4.f * x * (xx - 3.f * yy); By[24] = SH_C4[8] * -4.f * y * (3.f * xx - yy);
                }
            }
        }
    }
}

typedef struct {
    uint64_t key;
    uint32_t val;
} KV;

/* Stable merge sort on the 64-bit key (A.2): ties keep emission order.  The merges of one width are independent of
 * each other: OpenMP over them (the result does not depend on the thread count; the last widths have fewer merges than
 * threads and run on one or two cores).  The two arrays alternate as source and destination. */
static void kv_merge_sort(KV* a, KV* tmp, int64_t n) {
    KV *src = a, *dst = tmp;
    for (int64_t w = 1; w < n; w *= 2) {
        const int64_t merges = (n + 2 * w - 1) / (2 * w);
#pragma omp parallel for schedule(static) if (n > 65536 && merges >= 2)
        for (int64_t m = 0; m < merges; m++) {
            const int64_t lo = m * 2 * w;
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = (src[j].key < src[i].key) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        KV* t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, (size_t)n * sizeof(KV));
}

/*
 * TIGHT tile rects (an option of the BUILD, not of the reference; ggo_preprocess `tight_rects`).  The reference lists a
 * Gaussian in every tile of the square of radius ceil(3·sqrt(λmax)) around its mean; a pixel can only take it if
 * α = opacity·exp(−q/2) ≥ 1/255, i.e. inside the ellipse q ≤ 2·ln(255·opacity), whose axis-aligned bounding box has the
 * half widths sqrt(qmax·cov_xx), sqrt(qmax·cov_yy).  Intersecting the reference's rect with the tiles of that box (with
 * half a pixel and 1 % to spare) drops only (Gaussian, tile) pairs that every pixel of the tile would `continue` past:
 * images, final_T, radii and all gradients are unchanged bit for bit (tests/test_oracle.py), the lists are sub-lists of
 * the reference's in the same order.  The bound on ln is formed from the float's exponent and a cubic in its mantissa —
 * exact-order fp32 operations only, so that the C and the HIP side agree to the bit on every rect (libm's logf and the
 * device's differ in the last place).
 */
static float qmax_upper(float opacity) {   /* ≥ 2·ln(255·opacity); negative: below 1/255 everywhere */
    const float u = 255.0f * opacity;
    if (!(u >= 1.0f)) return -1.0f;
    uint32_t bits;
    memcpy(&bits, &u, 4);
    const int e = (int)(bits >> 23) - 127;
    const uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &mb, 4);
    const float x = m - 1.0f, t = x * x;
    const float lnm = (x - 0.5f * t) + 0.33333334f * (t * x);   /* ≥ ln(1 + x) on [0, 1): alternating series */
    return 2.0f * ((float)e * 0.69314718f + lnm) + 0.02f;
}
static void tighten_rect(float px, float py, float cov_xx, float cov_yy, float opacity, int* rmin, int* rmax) {
    const float qmax = qmax_upper(opacity);
    if (qmax < 0.f) { rmax[0] = rmin[0]; rmax[1] = rmin[1]; return; }
    const float hx = sqrtf(qmax * cov_xx) * 1.01f + 0.5f, hy = sqrtf(qmax * cov_yy) * 1.01f + 0.5f;
    rmin[0] = imax(rmin[0], (int)floorf((px - hx) / (float)GGO_TILE));
    rmin[1] = imax(rmin[1], (int)floorf((py - hy) / (float)GGO_TILE));
    rmax[0] = imin(rmax[0], (int)floorf((px + hx) / (float)GGO_TILE) + 1);
    rmax[1] = imin(rmax[1], (int)floorf((py + hy) / (float)GGO_TILE) + 1);
    if (rmax[0] < rmin[0]) rmax[0] = rmin[0];
    if (rmax[1] < rmin[1]) rmax[1] = rmin[1];
}

/*
 * Preprocess (A.1).  Any output pointer except radii/tiles_touched may be NULL.
 * cov3D_precomp XOR (scales, rotations); shs XOR colors_precomp.
 * cov3D_out[P,6] receives the covariance actually used.
 * Returns Σ tiles_touched.
 */
int64_t ggo_preprocess(int P, int D, int M, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales,
                       const float* rotations, float scale_modifier, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos, int W,
                       int H, float tanfovx, float tanfovy, float* depth, int32_t* radii, float* xy,
                       float* conic_opacity, float* rgb, uint8_t* clamped, int32_t* tiles_touched,
                       float* cov3D_out, int sh_cap, int32_t* rect_out /*[P,4] (x0,y0,x1,y1) or NULL*/,
                       int tight_rects /*0: the reference's rects*/) {
    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    const int gx = (W + GGO_TILE - 1) / GGO_TILE, gy = (H + GGO_TILE - 1) / GGO_TILE;
    const int deg = sh_eff_degree(D, shs ? M : 25, sh_cap);
    int64_t total = 0;
    /* Gaussians are independent: the loop order does not enter any result (total is an integer sum) */
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int i = 0; i < P; i++) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        if (rect_out) rect_out[4 * i] = rect_out[4 * i + 1] = rect_out[4 * i + 2] = rect_out[4 * i + 3] = 0;
        if (depth) depth[i] = 0.f;
        if (xy) xy[2 * i] = xy[2 * i + 1] = 0.f;
        if (conic_opacity) for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = 0.f;
        if (rgb) for (int k = 0; k < 3; k++) rgb[3 * i + k] = 0.f;
        if (clamped) for (int k = 0; k < 3; k++) clamped[3 * i + k] = 0;
        const float* p = means3D + 3 * i;
        float cov6[6];
        if (cov3D_precomp) memcpy(cov6, cov3D_precomp + 6 * i, sizeof cov6);
        else cov3d_from_scale_rot(scales + 3 * i, scale_modifier, rotations + 4 * i, cov6);
        if (cov3D_out) memcpy(cov3D_out + 6 * i, cov6, sizeof cov6);
        float pv[3];
        xform4x3(p, viewmatrix, pv);
        if (pv[2] <= GGO_NEAR_CULL) continue;
        float ph[4];
        xform4x4(p, projmatrix, ph);
        const float pw = 1.0f / (ph[3] + 0.0000001f);
        const float ppx = ph[0] * pw, ppy = ph[1] * pw;
        Ewa e;
        ewa_project(p, cov6, viewmatrix, fx, fy, tanfovx, tanfovy, &e);
        const float det = e.a * e.c - e.b * e.b;
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float con[3] = {e.c * det_inv, -e.b * det_inv, e.a * det_inv};
        const float mid = 0.5f * (e.a + e.c);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l1 = mid + sq, l2 = mid - sq;
        const float radf = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        const float px = ndc2pix(ppx, W), py = ndc2pix(ppy, H);
        /* NON-FINITE INPUTS — the build's contract (the reference has none: a NaN mean is culled by its `z <= 0.2` test or
         * not, a NaN covariance reaches `(int)ceil(NaN)`): a Gaussian whose projected geometry, opacity or colour is not
         * finite — a NaN / Inf in its mean, covariance (scale, rotation), opacity, evaluated SH coefficients or precomputed
         * colour — takes no part in the frame: radius 0, no list entry, zero gradient.  So does one whose radius exceeds
         * 2^30 px (the int conversion would overflow).  Same test, same place in csrc/preprocess.hip. */
        if (!(isfinite(px) && isfinite(py) && isfinite(con[0]) && isfinite(con[1]) && isfinite(con[2]) &&
              isfinite(opacities[i]) && isfinite(e.a) && isfinite(e.c) && isfinite(pv[2]) && radf < 1073741824.f))
            continue;
        const int rad = (int)radf;
        int rmin[2], rmax[2];
        get_rect(px, py, rad, gx, gy, rmin, rmax);
        int area = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
        if (area == 0) continue;   /* visibility (radii, colours) follows the REFERENCE's rect in both modes */
        if (tight_rects) {
            tighten_rect(px, py, e.a, e.c, opacities[i], rmin, rmax);
            area = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
        }
        int colour_finite = 1;   /* (the contract above, colour part) */
        if (rgb) {
            if (colors_precomp) {
                for (int k = 0; k < 3; k++) rgb[3 * i + k] = colors_precomp[3 * i + k];
            } else {
                float dir[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
                const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                dir[0] /= len; dir[1] /= len; dir[2] /= len;
                float B[25];
                sh_basis(deg, dir, B);
                const int K = (deg + 1) * (deg + 1);
                const float* sh = shs + (size_t)i * M * 3;
                for (int ch = 0; ch < 3; ch++) {
                    float r = 0.f;
                    for (int k = 0; k < K; k++) r += B[k] * sh[3 * k + ch];
                    r += 0.5f;
                    if (clamped) clamped[3 * i + ch] = (r < 0.f);
                    rgb[3 * i + ch] = fmaxf(r, 0.f);
                    if (!isfinite(r)) colour_finite = 0;
                }
            }
            for (int k = 0; k < 3; k++) if (!isfinite(rgb[3 * i + k])) colour_finite = 0;
            if (!colour_finite) {
                for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0.f; if (clamped) clamped[3 * i + k] = 0; }
                continue;
            }
        }
        if (rect_out) { rect_out[4 * i] = rmin[0]; rect_out[4 * i + 1] = rmin[1]; rect_out[4 * i + 2] = rmax[0]; rect_out[4 * i + 3] = rmax[1]; }
        if (depth) depth[i] = pv[2];
        radii[i] = rad;
        if (xy) { xy[2 * i] = px; xy[2 * i + 1] = py; }
        if (conic_opacity) {
            conic_opacity[4 * i] = con[0]; conic_opacity[4 * i + 1] = con[1];
            conic_opacity[4 * i + 2] = con[2]; conic_opacity[4 * i + 3] = opacities[i];
        }
        tiles_touched[i] = area;
        total += area;
    }
    return total;
}

/*
 * Binning (A.2): emit keys, stable sort, tile ranges.
 * point_list[N], ranges[tiles*2], optional keys_sorted[N].
 */
void ggo_bin(int P, int W, int H, const float* depth, const int32_t* radii, const float* xy,
             int64_t N, uint32_t* point_list, uint64_t* keys_sorted, int32_t* ranges,
             const int32_t* rect /*[P,4] from ggo_preprocess, or NULL: the reference's rects, recomputed*/) {
    const int gx = (W + GGO_TILE - 1) / GGO_TILE, gy = (H + GGO_TILE - 1) / GGO_TILE;
    KV* kv = (KV*)malloc(sizeof(KV) * (size_t)(N > 0 ? N : 1));
    KV* tmp = (KV*)malloc(sizeof(KV) * (size_t)(N > 0 ? N : 1));
    int64_t off = 0;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        int rmin[2], rmax[2];
        if (rect) { rmin[0] = rect[4 * i]; rmin[1] = rect[4 * i + 1]; rmax[0] = rect[4 * i + 2]; rmax[1] = rect[4 * i + 3]; }
        else get_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, rmin, rmax);
        uint32_t dbits;
        memcpy(&dbits, &depth[i], 4);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                kv[off].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                kv[off].val = (uint32_t)i;
                off++;
            }
    }
    kv_merge_sort(kv, tmp, N);
    for (int t = 0; t < gx * gy; t++) ranges[2 * t] = ranges[2 * t + 1] = 0;
    for (int64_t k = 0; k < N; k++) {
        point_list[k] = kv[k].val;
        if (keys_sorted) keys_sorted[k] = kv[k].key;
        const uint32_t tile = (uint32_t)(kv[k].key >> 32);
        if (k == 0) ranges[2 * tile] = 0;
        else {
            const uint32_t prev = (uint32_t)(kv[k - 1].key >> 32);
            if (prev != tile) { ranges[2 * prev + 1] = (int32_t)k; ranges[2 * tile] = (int32_t)k; }
        }
        if (k == N - 1) ranges[2 * tile + 1] = (int32_t)N;
    }
    free(kv);
    free(tmp);
}

/*
 * Forward blend (A.3).  features = rgb[P,3]; depth optional (third output of the
 * "w-depth" fork family: out_depth = Σ z α T).
 */
void ggo_blend_forward(int W, int H, const int32_t* ranges, const uint32_t* point_list,
                       const float* xy, const float* conic_opacity, const float* rgb,
                       const float* depth, const float* bg, float* out_color, float* final_T,
                       int32_t* n_contrib, float* out_depth) {
    const int gx = (W + GGO_TILE - 1) / GGO_TILE;
#pragma omp parallel for schedule(dynamic, 16)
    for (int py = 0; py < H; py++) {
        for (int px = 0; px < W; px++) {
            const int tile = (py / GGO_TILE) * gx + (px / GGO_TILE);
            const int r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pixx = (float)px, pixy = (float)py;
            float T = 1.0f, C[3] = {0.f, 0.f, 0.f}, Dz = 0.f;
            int contributor = 0, last = 0;
            for (int k = r0; k < r1; k++) {
                contributor++;
                const uint32_t g = point_list[k];
                const float dx = xy[2 * g] - pixx, dy = xy[2 * g + 1] - pixy;
                const float* co = conic_opacity + 4 * g;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf(GGO_ALPHA_MAX, co[3] * expf(power));
                if (alpha < GGO_ALPHA_MIN) continue;
                const float test_T = T * (1.f - alpha);
                if (test_T < GGO_T_MIN) break;
                for (int ch = 0; ch < 3; ch++) C[ch] += rgb[3 * g + ch] * alpha * T;
                if (depth) Dz += depth[g] * alpha * T;
                T = test_T;
                last = contributor;
            }
            const int pid = py * W + px;
            final_T[pid] = T;
            n_contrib[pid] = last;
            for (int ch = 0; ch < 3; ch++) out_color[ch * H * W + pid] = C[ch] + T * bg[ch];
            if (out_depth) out_depth[pid] = Dz;
        }
    }
}

/*
 * Backward blend (A.4, first half).  Accumulates in fp64:
 *   dL_dmean2D[P,2] (NDC units, i.e. already × W/2, H/2), dL_dconic[P,3]
 *   (xx, xy [half convention of upstream], yy), dL_dopacity[P], dL_drgb[P,3].
 * dL_dpix = [3,H,W].
 *
 * The per-pixel replay is upstream's, statement for statement.  Tiles are independent except for the sums
 * into the per-Gaussian accumulators, so tiles run in parallel (OpenMP): each keeps fp64 partial sums per entry
 * of its own list and adds them to the shared fp64 arrays with atomic updates at the end — only the ORDER of an
 * fp64 summation depends on the thread count (≈1e-16 relative), nothing else.  A pixel whose upstream gradient
 * is exactly zero contributes exactly zero to every sum (dL_dalpha = 0, dL_dG = 0) and is skipped: this is what
 * makes windowed full-size comparisons (dL zero outside a few windows) affordable.
 */
void ggo_blend_backward(int P, int W, int H, const int32_t* ranges, const uint32_t* point_list,
                        const float* xy, const float* conic_opacity, const float* rgb,
                        const float* bg, const float* final_T, const int32_t* n_contrib,
                        const float* dL_dpix, double* dL_dmean2D, double* dL_dconic,
                        double* dL_dopacity, double* dL_drgb) {
    const int gx = (W + GGO_TILE - 1) / GGO_TILE, gy = (H + GGO_TILE - 1) / GGO_TILE;
    memset(dL_dmean2D, 0, sizeof(double) * 2 * (size_t)P);
    memset(dL_dconic, 0, sizeof(double) * 3 * (size_t)P);
    memset(dL_dopacity, 0, sizeof(double) * (size_t)P);
    memset(dL_drgb, 0, sizeof(double) * 3 * (size_t)P);
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
#pragma omp parallel
    {
        double* loc = NULL;  /* [entries replayed in this tile][9]: mean2D 0-1, conic 2-4, opacity 5, rgb 6-8 */
        size_t cap = 0;
#pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < gx * gy; tile++) {
            const int r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            if (r1 <= r0) continue;
            const int tx = tile % gx, ty = tile / gx;
            const int x_end = imin(W, (tx + 1) * GGO_TILE), y_end = imin(H, (ty + 1) * GGO_TILE);
            int top = 0;
            for (int py = ty * GGO_TILE; py < y_end; py++)
                for (int px = tx * GGO_TILE; px < x_end; px++) {
                    const int pid = py * W + px;
                    if (dL_dpix[pid] == 0.f && dL_dpix[H * W + pid] == 0.f && dL_dpix[2 * H * W + pid] == 0.f) continue;
                    top = imax(top, n_contrib[pid]);
                }
            if (top == 0) continue;
            if ((size_t)top * 9 > cap) {
                cap = (size_t)top * 9;
                free(loc);
                loc = (double*)malloc(sizeof(double) * cap);
            }
            memset(loc, 0, sizeof(double) * 9 * (size_t)top);
            for (int py = ty * GGO_TILE; py < y_end; py++)
                for (int px = tx * GGO_TILE; px < x_end; px++) {
                    const int pid = py * W + px;
                    const float dpix[3] = {dL_dpix[pid], dL_dpix[H * W + pid], dL_dpix[2 * H * W + pid]};
                    if (dpix[0] == 0.f && dpix[1] == 0.f && dpix[2] == 0.f) continue;
                    const float pixx = (float)px, pixy = (float)py;
                    const float T_final = final_T[pid];
                    float T = T_final;
                    const int last = n_contrib[pid];
                    float accum[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f}, last_alpha = 0.f;
                    const float bg_dot = bg[0] * dpix[0] + bg[1] * dpix[1] + bg[2] * dpix[2];
                    for (int k = r0 + last - 1; k >= r0; k--) {
                        const uint32_t g = point_list[k];
                        const float dx = xy[2 * g] - pixx, dy = xy[2 * g + 1] - pixy;
                        const float* co = conic_opacity + 4 * g;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        const float alpha = fminf(GGO_ALPHA_MAX, co[3] * G);
                        if (alpha < GGO_ALPHA_MIN) continue;
                        T = T / (1.f - alpha);
                        const float dchannel_dcolor = alpha * T;
                        double* a = loc + 9 * (size_t)(k - r0);
                        float dL_dalpha = 0.f;
                        for (int ch = 0; ch < 3; ch++) {
                            const float c = rgb[3 * g + ch];
                            accum[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum[ch];
                            last_color[ch] = c;
                            dL_dalpha += (c - accum[ch]) * dpix[ch];
                            a[6 + ch] += (double)(dchannel_dcolor * dpix[ch]);
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                        const float dL_dG = co[3] * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const float dG_ddely = -gdy * co[2] - gdx * co[1];
                        a[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                        a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                        a[2] += (double)(-0.5f * gdx * dx * dL_dG);
                        a[3] += (double)(-0.5f * gdx * dy * dL_dG);
                        a[4] += (double)(-0.5f * gdy * dy * dL_dG);
                        a[5] += (double)(G * dL_dalpha);
                    }
                }
            for (int k = 0; k < top; k++) {
                const double* a = loc + 9 * (size_t)k;
                const size_t g = point_list[r0 + k];
                double* dst[9] = {dL_dmean2D + 2 * g, dL_dmean2D + 2 * g + 1, dL_dconic + 3 * g, dL_dconic + 3 * g + 1,
                                  dL_dconic + 3 * g + 2, dL_dopacity + g, dL_drgb + 3 * g, dL_drgb + 3 * g + 1,
                                  dL_drgb + 3 * g + 2};
                for (int v = 0; v < 9; v++)
                    if (a[v] != 0.0) {
#pragma omp atomic
                        *dst[v] += a[v];
                    }
            }
        }
        free(loc);
    }
}

static void scale_rot_backward(const float* s, float mod, const float* q, const float* dL_dcov6,
                               float* dL_ds, float* dL_dq) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                  2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                  2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
    float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float Mx[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Mx[3 * i + j] = R[3 * i + j] * sc[j];
    /* dL/dSigma as a full symmetric matrix: the 6-vector's off-diagonals already carry both
       symmetric contributions, so each off-diagonal entry gets half. */
    float dS[9] = {dL_dcov6[0], 0.5f * dL_dcov6[1], 0.5f * dL_dcov6[2],
                   0.5f * dL_dcov6[1], dL_dcov6[3], 0.5f * dL_dcov6[4],
                   0.5f * dL_dcov6[2], 0.5f * dL_dcov6[4], dL_dcov6[5]};
    /* Sigma = M Mᵀ → dL/dM = 2 dS M */
    float dM[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float a = 0.f;
            for (int k = 0; k < 3; k++) a += dS[3 * i + k] * Mx[3 * k + j];
            dM[3 * i + j] = 2.f * a;
        }
    /* M_ij = R_ij sc_j */
    float dR[9];
    for (int j = 0; j < 3; j++) {
        float a = 0.f;
        for (int i = 0; i < 3; i++) { a += dM[3 * i + j] * R[3 * i + j]; dR[3 * i + j] = dM[3 * i + j] * sc[j]; }
        dL_ds[j] = a * mod;
    }
    /* R(q) derivative (q un-normalised, as upstream) */
    dL_dq[0] = 2.f * (z * (dR[3] - dR[1]) + y * (dR[2] - dR[6]) + x * (dR[7] - dR[5]));
    dL_dq[1] = 2.f * (y * (dR[1] + dR[3]) + z * (dR[2] + dR[6]) + r * (dR[7] - dR[5])) - 4.f * x * (dR[4] + dR[8]);
    dL_dq[2] = 2.f * (x * (dR[1] + dR[3]) + r * (dR[2] - dR[6]) + z * (dR[5] + dR[7])) - 4.f * y * (dR[0] + dR[8]);
    dL_dq[3] = 2.f * (r * (dR[3] - dR[1]) + x * (dR[2] + dR[6]) + y * (dR[5] + dR[7])) - 4.f * z * (dR[0] + dR[4]);
}

/*
 * Per-Gaussian backward (A.4, second half): conic→cov2D→cov3D and mean3D
 * (through J and through the perspective divide), SH backward.
 * Inputs are the fp64 sums from ggo_blend_backward (cast to fp32 first, as the
 * device holds them in fp32).  Outputs zero for culled Gaussians.
 */
void ggo_preprocess_backward(int P, int D, int M, const float* means3D, const float* shs,
                             int has_colors_precomp, const float* scales, const float* rotations,
                             float scale_modifier, const float* cov3D_used,
                             const float* viewmatrix, const float* projmatrix, const float* campos,
                             int W, int H, float tanfovx, float tanfovy, const int32_t* radii,
                             const uint8_t* clamped, const double* dL_dmean2D_acc,
                             const double* dL_dconic_acc, const double* dL_drgb_acc,
                             float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dsh,
                             float* dL_dcolors_precomp, float* dL_dcov3D, float* dL_dscales,
                             float* dL_drotations, int sh_cap) {
    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    const int deg = sh_eff_degree(D, shs ? M : 25, sh_cap);
    const int K = (deg + 1) * (deg + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        for (int k = 0; k < 3; k++) { dL_dmeans3D[3 * i + k] = 0.f; dL_dmeans2D[3 * i + k] = 0.f; }
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = 0.f;
        if (dL_dsh) for (int k = 0; k < 3 * M; k++) dL_dsh[(size_t)i * 3 * M + k] = 0.f;
        if (dL_dcolors_precomp) for (int k = 0; k < 3; k++) dL_dcolors_precomp[3 * i + k] = 0.f;
        if (dL_dscales) for (int k = 0; k < 3; k++) dL_dscales[3 * i + k] = 0.f;
        if (dL_drotations) for (int k = 0; k < 4; k++) dL_drotations[4 * i + k] = 0.f;
        if (!(radii[i] > 0)) continue;
        const float* p = means3D + 3 * i;
        const float* cov6 = cov3D_used + 6 * i;
        const float dcon[3] = {(float)dL_dconic_acc[3 * i], (float)dL_dconic_acc[3 * i + 1],
                               (float)dL_dconic_acc[3 * i + 2]};
        Ewa e;
        ewa_project(p, cov6, viewmatrix, fx, fy, tanfovx, tanfovy, &e);
        const float a = e.a, b = e.b, c = e.c;
        const float denom = a * c - b * b;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float* A0 = e.A;
        const float* A1 = e.A + 3;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * dcon[0] + 2.f * b * c * dcon[1] + (denom - a * c) * dcon[2]);
            dL_dc = denom2inv * (-a * a * dcon[2] + 2.f * a * b * dcon[1] + (denom - a * c) * dcon[0]);
            dL_db = denom2inv * 2.f * (b * c * dcon[0] - (denom + 2.f * b * b) * dcon[1] + a * b * dcon[2]);
            float* g = dL_dcov3D + 6 * i;
            g[0] = A0[0] * A0[0] * dL_da + A0[0] * A1[0] * dL_db + A1[0] * A1[0] * dL_dc;
            g[3] = A0[1] * A0[1] * dL_da + A0[1] * A1[1] * dL_db + A1[1] * A1[1] * dL_dc;
            g[5] = A0[2] * A0[2] * dL_da + A0[2] * A1[2] * dL_db + A1[2] * A1[2] * dL_dc;
            g[1] = 2.f * A0[0] * A0[1] * dL_da + (A0[0] * A1[1] + A0[1] * A1[0]) * dL_db + 2.f * A1[0] * A1[1] * dL_dc;
            g[2] = 2.f * A0[0] * A0[2] * dL_da + (A0[0] * A1[2] + A0[2] * A1[0]) * dL_db + 2.f * A1[0] * A1[2] * dL_dc;
            g[4] = 2.f * A0[2] * A0[1] * dL_da + (A0[1] * A1[2] + A0[2] * A1[1]) * dL_db + 2.f * A1[1] * A1[2] * dL_dc;
        }
        /* dL/dA (2x3): a = A0 Σ A0ᵀ, b = A0 Σ A1ᵀ, c = A1 Σ A1ᵀ */
        const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
        float SA0[3], SA1[3];
        for (int j = 0; j < 3; j++) {
            SA0[j] = A0[0] * S[j] + A0[1] * S[3 + j] + A0[2] * S[6 + j];
            SA1[j] = A1[0] * S[j] + A1[1] * S[3 + j] + A1[2] * S[6 + j];
        }
        float dA0[3], dA1[3];
        for (int j = 0; j < 3; j++) {
            dA0[j] = 2.f * SA0[j] * dL_da + SA1[j] * dL_db;
            dA1[j] = 2.f * SA1[j] * dL_dc + SA0[j] * dL_db;
        }
        /* A = J R → dL/dJ_ik = Σ_j dA_ij R_kj, R[k][j] = V[4*j+k] */
        float R[9];
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 3; j++) R[3 * k + j] = viewmatrix[4 * j + k];
        const float dJ00 = dA0[0] * R[0] + dA0[1] * R[1] + dA0[2] * R[2];
        const float dJ02 = dA0[0] * R[6] + dA0[1] * R[7] + dA0[2] * R[8];
        const float dJ11 = dA1[0] * R[3] + dA1[1] * R[4] + dA1[2] * R[5];
        const float dJ12 = dA1[0] * R[6] + dA1[1] * R[7] + dA1[2] * R[8];
        const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = e.xmul * -fx * tz2 * dJ02;
        const float dty = e.ymul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * e.t[0]) * tz3 * dJ02 +
                          (2.f * fy * e.t[1]) * tz3 * dJ12;
        /* t = R p + T → dL/dp = Rᵀ dt */
        float dmean[3];
        for (int j = 0; j < 3; j++) dmean[j] = R[j] * dtx + R[3 + j] * dty + R[6 + j] * dtz;

        /* mean2D (NDC) → mean3D through the perspective divide */
        const float d2x = (float)dL_dmean2D_acc[2 * i], d2y = (float)dL_dmean2D_acc[2 * i + 1];
        dL_dmeans2D[3 * i] = d2x; dL_dmeans2D[3 * i + 1] = d2y;
        const float* pm = projmatrix;
        float mh[4];
        xform4x4(p, pm, mh);
        const float mw = 1.0f / (mh[3] + 0.0000001f);
        const float mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
        dmean[0] += (pm[0] * mw - pm[3] * mul1) * d2x + (pm[1] * mw - pm[3] * mul2) * d2y;
        dmean[1] += (pm[4] * mw - pm[7] * mul1) * d2x + (pm[5] * mw - pm[7] * mul2) * d2y;
        dmean[2] += (pm[8] * mw - pm[11] * mul1) * d2x + (pm[9] * mw - pm[11] * mul2) * d2y;

        /* colour */
        float dcol[3] = {(float)dL_drgb_acc[3 * i], (float)dL_drgb_acc[3 * i + 1], (float)dL_drgb_acc[3 * i + 2]};
        if (has_colors_precomp) {
            for (int k = 0; k < 3; k++) dL_dcolors_precomp[3 * i + k] = dcol[k];
        } else {
            for (int ch = 0; ch < 3; ch++) if (clamped[3 * i + ch]) dcol[ch] = 0.f;
            float dirO[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
            const float len = sqrtf(dirO[0] * dirO[0] + dirO[1] * dirO[1] + dirO[2] * dirO[2]);
            const float dir[3] = {dirO[0] / len, dirO[1] / len, dirO[2] / len};
            float B[25], Bx[25], By[25], Bz[25];
            sh_basis(deg, dir, B);
            sh_basis_grad(deg, dir, Bx, By, Bz);
            const float* sh = shs + (size_t)i * M * 3;
            float* dsh = dL_dsh + (size_t)i * M * 3;
            float ddir[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < K; k++)
                for (int ch = 0; ch < 3; ch++) {
                    dsh[3 * k + ch] = B[k] * dcol[ch];
                    ddir[0] += Bx[k] * sh[3 * k + ch] * dcol[ch];
                    ddir[1] += By[k] * sh[3 * k + ch] * dcol[ch];
                    ddir[2] += Bz[k] * sh[3 * k + ch] * dcol[ch];
                }
            /* d normalize(v)/dv */
            const float sum2 = dirO[0] * dirO[0] + dirO[1] * dirO[1] + dirO[2] * dirO[2];
            const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float vx = dirO[0], vy = dirO[1], vz = dirO[2];
            dmean[0] += ((sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * inv32;
            dmean[1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * inv32;
            dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * inv32;
        }
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = dmean[k];
        if (scales && dL_dscales && dL_drotations)
            scale_rot_backward(scales + 3 * i, scale_modifier, rotations + 4 * i, dL_dcov3D + 6 * i,
                               dL_dscales + 3 * i, dL_drotations + 4 * i);
    }
}

int ggo_abi_version(void) { return 2; }
int ggo_num_threads(void) { return omp_get_max_threads(); }
