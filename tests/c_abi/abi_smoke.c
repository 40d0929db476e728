/* abi_smoke.c — drives the C ABI from plain C (no Python, no torch): hipMalloc'd buffers, one forward and one
 * backward — and the same through the several-views entry points — checked against the closed form of a single centred
 * Gaussian:
 *   colour(centre pixel) = c·alpha + (1-alpha)·bg,  alpha = min(0.99, opacity·exp(-½ dᵀ conic d)).
 * Build: hipcc -x c tests/c_abi/abi_smoke.c -Iinclude -Lggrt_official_amd -lggr_raster -lamdhip64 -lm -o abi_smoke
 * This is the binding a non-Python host would write. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ggr_raster.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)

typedef struct { void* p[2]; int n; } Two;
static void* two_alloc(void* ctx, size_t bytes) {
    Two* t = (Two*)ctx;
    void* p = NULL;
    if (t->n >= 2 || hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return NULL;
    t->p[t->n++] = p;
    return p;
}

static float* upload(const float* h, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

int main(void) {
    if (ggr_abi_version() != GGR_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    enum { W = 33, H = 17, P = 2 };
    /* identity camera looking down +z, fov 90°: tan = 1; projection as cuda_splatting.py:18-46 with near 1 far 100 */
    const float tanx = 1.0f, tany = (float)H / (float)W;
    const float fxn = 0.5f / tanx, fyn = 0.5f / tany, zn = 1.f, zf = 100.f;
    float view[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
    /* row-vector convention: proj = view @ P^T, P rows: (2n fx, 0, 2cx-1, 0) (0, 2n fy, 2cy-1, 0) (0,0,f/(f-n),-fn/(f-n)) (0,0,1,0) */
    float proj[16] = {2*zn*fxn,0,0,0,  0,2*zn*fyn,0,0,  0,0,zf/(zf-zn),1,  0,0,-(zf*zn)/(zf-zn),0};
    float campos[3] = {0,0,0}, bg[3] = {0.25f, 0.5f, 0.75f};
    /* Gaussian 0: on the optical axis at z = 4, isotropic sigma 0.3, opacity 0.6, colour (0.9,0.1,0.4);
       Gaussian 1: behind the camera (culled) */
    float means[P*3] = {0,0,4,  0,0,-3};
    float cov[P*6] = {0.09f,0,0,0.09f,0,0.09f,  0.09f,0,0,0.09f,0,0.09f};
    float colors[P*3] = {0.9f,0.1f,0.4f,  1,1,1};
    float opac[P] = {0.6f, 0.9f};
    float *d_view = upload(view,16), *d_proj = upload(proj,16), *d_cam = upload(campos,3), *d_bg = upload(bg,3);
    float *d_means = upload(means,P*3), *d_cov = upload(cov,P*6), *d_col = upload(colors,P*3), *d_op = upload(opac,P);
    float *d_color, *d_depth; int32_t* d_radii; void *d_geom, *d_img;
    CHECK(hipMalloc((void**)&d_color, 3*W*H*4)); CHECK(hipMalloc((void**)&d_depth, W*H*4));
    CHECK(hipMalloc((void**)&d_radii, P*4));
    CHECK(hipMalloc(&d_geom, ggr_geom_bytes(P))); CHECK(hipMalloc(&d_img, ggr_image_bytes(W, H)));

    GgrSettings st; memset(&st, 0, sizeof st);
    st.image_height = H; st.image_width = W; st.sh_degree = 0; st.sh_stride = 0; st.num_points = P;
    st.tanfovx = tanx; st.tanfovy = tany; st.scale_modifier = 1.f;
    st.bg = d_bg; st.viewmatrix = d_view; st.projmatrix = d_proj; st.campos = d_cam;
    GgrForwardIn in; memset(&in, 0, sizeof in);
    in.means3D = d_means; in.colors_precomp = d_col; in.opacities = d_op; in.cov3D_precomp = d_cov;
    GgrForwardOut out; memset(&out, 0, sizeof out);
    out.out_color = d_color; out.radii = d_radii; out.out_depth = d_depth; out.geom_buffer = d_geom; out.image_buffer = d_img;
    Two mem; memset(&mem, 0, sizeof mem);
    if (ggr_forward(&st, &in, &out, two_alloc, &mem, NULL) != GGR_OK) { fprintf(stderr, "forward: %s\n", ggr_last_error()); return 1; }
    CHECK(hipDeviceSynchronize());

    float h_color[3*W*H]; int32_t h_radii[P];
    CHECK(hipMemcpy(h_color, d_color, sizeof h_color, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h_radii, d_radii, sizeof h_radii, hipMemcpyDeviceToHost));
    /* expected: focal (pixels) fx = W/(2 tan) ; cov2D = (fx/z)^2·0.09 + 0.3 ; centre pixel (16, 8) is at the mean:
       ndc 0 → pixel ((0+1)·W-1)/2 = 16, ((0+1)·H-1)/2 = 8 → d = 0 → alpha = opacity */
    const int cx = 16, cy = 8;
    const float alpha = 0.6f;
    int bad = 0;
    for (int c = 0; c < 3; c++) {
        const float want = colors[c] * alpha + (1.f - alpha) * bg[c];
        const float got = h_color[c*W*H + cy*W + cx];
        if (fabsf(got - want) > 1e-5f) { fprintf(stderr, "channel %d: got %f want %f\n", c, got, want); bad = 1; }
    }
    const float fx = (float)W / (2.f * tanx), var = (fx / 4.f) * (fx / 4.f) * 0.09f + 0.3f;
    const int want_radius = (int)ceilf(3.f * sqrtf(var));
    if (h_radii[0] != want_radius || h_radii[1] != 0) { fprintf(stderr, "radii %d %d, want %d 0\n", h_radii[0], h_radii[1], want_radius); bad = 1; }
    if (out.num_rendered <= 0) { fprintf(stderr, "num_rendered %lld\n", (long long)out.num_rendered); bad = 1; }
    /* far corner: background only */
    for (int c = 0; c < 3; c++) if (fabsf(h_color[c*W*H + 0] - bg[c]) > 1e-6f) { fprintf(stderr, "corner pixel not background\n"); bad = 1; }

    /* backward with dL/dcolour = 1 at the centre pixel only: dL/dcolour_precomp[0] = alpha·T = alpha */
    float h_dL[3*W*H]; memset(h_dL, 0, sizeof h_dL);
    for (int c = 0; c < 3; c++) h_dL[c*W*H + cy*W + cx] = 1.f;
    float* d_dL = upload(h_dL, 3*W*H);
    void* d_scratch; CHECK(hipMalloc(&d_scratch, ggr_backward_scratch_bytes(P)));
    float *g_means, *g_m2d, *g_col, *g_op, *g_cov;
    CHECK(hipMalloc((void**)&g_means, P*3*4)); CHECK(hipMalloc((void**)&g_m2d, P*3*4)); CHECK(hipMalloc((void**)&g_col, P*3*4));
    CHECK(hipMalloc((void**)&g_op, P*4)); CHECK(hipMalloc((void**)&g_cov, P*6*4));
    GgrBackwardIn bi; memset(&bi, 0, sizeof bi);
    bi.fwd = in; bi.radii = d_radii; bi.geom_buffer = d_geom; bi.image_buffer = d_img; bi.binning_buffer = out.binning_buffer;
    bi.num_rendered = out.num_rendered; bi.dL_dout_color = d_dL; bi.scratch = d_scratch;
    GgrBackwardOut bo; memset(&bo, 0, sizeof bo);
    bo.dL_dmeans3D = g_means; bo.dL_dmeans2D = g_m2d; bo.dL_dcolors_precomp = g_col; bo.dL_dopacities = g_op; bo.dL_dcov3D = g_cov;
    if (ggr_backward(&st, &bi, &bo, NULL) != GGR_OK) { fprintf(stderr, "backward: %s\n", ggr_last_error()); return 1; }
    CHECK(hipDeviceSynchronize());
    float h_gcol[P*3], h_gop[P];
    CHECK(hipMemcpy(h_gcol, g_col, sizeof h_gcol, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h_gop, g_op, sizeof h_gop, hipMemcpyDeviceToHost));
    for (int c = 0; c < 3; c++) if (fabsf(h_gcol[c] - alpha) > 1e-6f) { fprintf(stderr, "dL/dcolour[%d] = %f, want %f\n", c, h_gcol[c], alpha); bad = 1; }
    /* dL/dopacity = G · Σ_c (colour_c - bg_c) with G = 1 at the centre */
    const float want_gop = (colors[0] - bg[0]) + (colors[1] - bg[1]) + (colors[2] - bg[2]);
    if (fabsf(h_gop[0] - want_gop) > 1e-5f || h_gop[1] != 0.f) { fprintf(stderr, "dL/dopacity = %f %f, want %f 0\n", h_gop[0], h_gop[1], want_gop); bad = 1; }
    for (int c = 3; c < 6; c++) if (h_gcol[c] != 0.f) { fprintf(stderr, "culled Gaussian has a colour gradient\n"); bad = 1; }

    /* ---- two views of the same Gaussians in ONE launch set (ggr_forward_views / ggr_backward_views): view 0 is the camera
       above, view 1 the same camera with another background — each view's centre pixel has its own closed form, the
       Gaussian gradients come back summed over the views, the forward clears the backward's scratch on the side ---- */
    {
        enum { V = 2 };
        float views2[V*16], projs2[V*16], cams2[V*3] = {0,0,0, 0,0,0}, bgs2[V*3] = {0.25f,0.5f,0.75f, 0.f,1.f,0.f};
        float tans2[V*2] = {tanx, tany, tanx, tany};
        memcpy(views2, view, sizeof view); memcpy(views2 + 16, view, sizeof view);
        memcpy(projs2, proj, sizeof proj); memcpy(projs2 + 16, proj, sizeof proj);
        GgrViews vw; memset(&vw, 0, sizeof vw);
        vw.num_views = V; vw.viewmatrix = upload(views2, V*16); vw.projmatrix = upload(projs2, V*16);
        vw.campos = upload(cams2, V*3); vw.bg = upload(bgs2, V*3); vw.tanfov = upload(tans2, V*2);
        float *v_color, *v_depth; int32_t* v_radii; void *v_geom, *v_img, *v_scratch;
        CHECK(hipMalloc((void**)&v_color, V*3*W*H*4)); CHECK(hipMalloc((void**)&v_depth, V*W*H*4));
        CHECK(hipMalloc((void**)&v_radii, V*P*4));
        CHECK(hipMalloc(&v_geom, ggr_geom_bytes_views(P, V))); CHECK(hipMalloc(&v_img, ggr_image_bytes_views(W, H, V)));
        CHECK(hipMalloc(&v_scratch, ggr_backward_scratch_bytes_views(P, V)));
        GgrForwardOut vo; memset(&vo, 0, sizeof vo);
        vo.out_color = v_color; vo.radii = v_radii; vo.out_depth = v_depth; vo.geom_buffer = v_geom; vo.image_buffer = v_img;
        vo.backward_scratch = v_scratch;
        Two vmem; memset(&vmem, 0, sizeof vmem);
        if (ggr_forward_views(&st, &vw, &in, &vo, two_alloc, &vmem, NULL) != GGR_OK) { fprintf(stderr, "forward_views: %s\n", ggr_last_error()); return 1; }
        CHECK(hipDeviceSynchronize());
        static float hv[V*3*W*H];
        CHECK(hipMemcpy(hv, v_color, sizeof hv, hipMemcpyDeviceToHost));
        for (int v = 0; v < V; v++)
            for (int c = 0; c < 3; c++) {
                const float want = colors[c] * alpha + (1.f - alpha) * bgs2[3*v + c];
                const float got = hv[(v*3 + c)*W*H + cy*W + cx];
                if (fabsf(got - want) > 1e-5f) { fprintf(stderr, "view %d channel %d: got %f want %f\n", v, c, got, want); bad = 1; }
            }
        if (vo.num_rendered != 2 * out.num_rendered) { fprintf(stderr, "views num_rendered %lld\n", (long long)vo.num_rendered); bad = 1; }
        static float hdL2[V*3*W*H]; memset(hdL2, 0, sizeof hdL2);
        for (int v = 0; v < V; v++) for (int c = 0; c < 3; c++) hdL2[(v*3 + c)*W*H + cy*W + cx] = 1.f;
        float* d_dL2 = upload(hdL2, V*3*W*H);
        float* g_m2d2; CHECK(hipMalloc((void**)&g_m2d2, V*P*3*4));
        GgrBackwardIn vbi; memset(&vbi, 0, sizeof vbi);
        vbi.fwd = in; vbi.radii = v_radii; vbi.geom_buffer = v_geom; vbi.image_buffer = v_img; vbi.binning_buffer = vo.binning_buffer;
        vbi.num_rendered = vo.num_rendered; vbi.dL_dout_color = d_dL2; vbi.scratch = v_scratch; vbi.scratch_zeroed = 1;
        GgrBackwardOut vbo = bo; vbo.dL_dmeans2D = g_m2d2;
        if (ggr_backward_views(&st, &vw, &vbi, &vbo, NULL) != GGR_OK) { fprintf(stderr, "backward_views: %s\n", ggr_last_error()); return 1; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h_gcol, g_col, sizeof h_gcol, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h_gop, g_op, sizeof h_gop, hipMemcpyDeviceToHost));
        /* summed over the two views: dL/dcolour = 2·alpha; dL/dopacity = Σ_views Σ_c (colour_c − bg_c) */
        for (int c = 0; c < 3; c++) if (fabsf(h_gcol[c] - 2.f * alpha) > 1e-6f) { fprintf(stderr, "views dL/dcolour[%d] = %f\n", c, h_gcol[c]); bad = 1; }
        float want2 = 0.f;
        for (int v = 0; v < V; v++) for (int c = 0; c < 3; c++) want2 += colors[c] - bgs2[3*v + c];
        if (fabsf(h_gop[0] - want2) > 1e-5f) { fprintf(stderr, "views dL/dopacity = %f, want %f\n", h_gop[0], want2); bad = 1; }
        hipFree(vmem.p[0]); hipFree(vmem.p[1]);

        /* ---- the same two views as two Gaussian SETS (GgrViews.num_sets = 2: inputs [2, P, …], view v renders set v —
           the reference's (b v) flattening, decoder_splatting_cuda.py:40-60): set 1 has another colour and opacity, so
           each view's centre pixel follows its own set's closed form and the gradients come back per set ---- */
        float means2[2*P*3], cov2[2*P*6], colors2[2*P*3] = {0.9f,0.1f,0.4f, 1,1,1,  0.2f,0.7f,0.3f, 1,1,1};
        float opac2[2*P] = {0.6f, 0.9f, 0.35f, 0.9f};
        memcpy(means2, means, sizeof means); memcpy(means2 + P*3, means, sizeof means);
        memcpy(cov2, cov, sizeof cov); memcpy(cov2 + P*6, cov, sizeof cov);
        GgrForwardIn in2 = in;
        in2.means3D = upload(means2, 2*P*3); in2.cov3D_precomp = upload(cov2, 2*P*6);
        in2.colors_precomp = upload(colors2, 2*P*3); in2.opacities = upload(opac2, 2*P);
        vw.num_sets = 2;
        Two smem; memset(&smem, 0, sizeof smem);
        if (ggr_forward_views(&st, &vw, &in2, &vo, two_alloc, &smem, NULL) != GGR_OK) { fprintf(stderr, "forward_views (sets): %s\n", ggr_last_error()); return 1; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(hv, v_color, sizeof hv, hipMemcpyDeviceToHost));
        for (int v = 0; v < V; v++) {
            const float al = fminf(0.99f, opac2[v*P]);
            for (int c = 0; c < 3; c++) {
                const float want = colors2[v*P*3 + c] * al + (1.f - al) * bgs2[3*v + c];
                const float got = hv[(v*3 + c)*W*H + cy*W + cx];
                if (fabsf(got - want) > 1e-5f) { fprintf(stderr, "set %d channel %d: got %f want %f\n", v, c, got, want); bad = 1; }
            }
        }
        float *s_col, *s_op, *s_m3, *s_cov;   /* per-set gradients: [2, P, …] */
        CHECK(hipMalloc((void**)&s_col, 2*P*3*4)); CHECK(hipMalloc((void**)&s_op, 2*P*4));
        CHECK(hipMalloc((void**)&s_m3, 2*P*3*4)); CHECK(hipMalloc((void**)&s_cov, 2*P*6*4));
        GgrBackwardIn sbi = vbi; sbi.fwd = in2; sbi.binning_buffer = vo.binning_buffer; sbi.num_rendered = vo.num_rendered;
        GgrBackwardOut sbo = vbo; sbo.dL_dcolors_precomp = s_col; sbo.dL_dopacities = s_op; sbo.dL_dmeans3D = s_m3; sbo.dL_dcov3D = s_cov;
        if (ggr_backward_views(&st, &vw, &sbi, &sbo, NULL) != GGR_OK) { fprintf(stderr, "backward_views (sets): %s\n", ggr_last_error()); return 1; }
        CHECK(hipDeviceSynchronize());
        float h_scol[2*P*3];
        CHECK(hipMemcpy(h_scol, s_col, sizeof h_scol, hipMemcpyDeviceToHost));
        for (int v = 0; v < V; v++)
            for (int c = 0; c < 3; c++)
                if (fabsf(h_scol[v*P*3 + c] - fminf(0.99f, opac2[v*P])) > 1e-6f) { fprintf(stderr, "set %d dL/dcolour[%d] = %f\n", v, c, h_scol[v*P*3 + c]); bad = 1; }
        hipFree(smem.p[0]); hipFree(smem.p[1]);
        vw.num_sets = 0;
    }

    /* ---- scissored forward (GgrSettings.scissor, the deferred back-propagation cell of finetune_ggrt_stable.py:126-142):
       a window around the centre renders the centre pixel exactly as before; a window in the top-left corner does not
       contain the Gaussian — every pixel is background and nothing is visible ---- */
    {
        GgrSettings sc = st;
        sc.scissor[0] = cx - 3; sc.scissor[1] = cy - 2; sc.scissor[2] = cx + 4; sc.scissor[3] = cy + 3;
        Two m2; memset(&m2, 0, sizeof m2);
        if (ggr_forward(&sc, &in, &out, two_alloc, &m2, NULL) != GGR_OK) { fprintf(stderr, "scissored forward: %s\n", ggr_last_error()); return 1; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h_color, d_color, sizeof h_color, hipMemcpyDeviceToHost));
        for (int c = 0; c < 3; c++) {
            const float want = colors[c] * alpha + (1.f - alpha) * bg[c];
            if (fabsf(h_color[c*W*H + cy*W + cx] - want) > 1e-5f) { fprintf(stderr, "scissor: centre channel %d = %f, want %f\n", c, h_color[c*W*H + cy*W + cx], want); bad = 1; }
        }
        hipFree(m2.p[0]); hipFree(m2.p[1]);
        sc.scissor[0] = 0; sc.scissor[1] = 0; sc.scissor[2] = 4; sc.scissor[3] = 4;
        memset(&m2, 0, sizeof m2);
        if (ggr_forward(&sc, &in, &out, two_alloc, &m2, NULL) != GGR_OK) { fprintf(stderr, "scissored forward 2: %s\n", ggr_last_error()); return 1; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h_color, d_color, sizeof h_color, hipMemcpyDeviceToHost));
        for (int c = 0; c < 3; c++)
            if (h_color[c*W*H + cy*W + cx] != bg[c]) { fprintf(stderr, "scissor: centre outside the window is not background\n"); bad = 1; }
        /* (the Gaussian's rect reaches into tile (0,0), so it is still listed there: one entry, not two) */
        if (out.num_rendered != 1) { fprintf(stderr, "scissor: %lld entries, want 1 (tile 0 of the two the Gaussian touches)\n", (long long)out.num_rendered); bad = 1; }
        hipFree(m2.p[0]); hipFree(m2.p[1]);
    }

    hipFree(mem.p[0]); hipFree(mem.p[1]);
    printf(bad ? "C ABI SMOKE FAILED\n" : "C ABI SMOKE OK (radius %d)\n", h_radii[0]);
    return bad;
}
