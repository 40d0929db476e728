/*
 * ggr_raster.h — C ABI of the MI355X-native differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary for the ONE native call GGRt makes on its render hot path:
 *
 *   reference  ggrt/model/pixelsplat/decoder/cuda_splatting.py:6-9      (import of the extension)
 *   reference  ggrt/model/pixelsplat/decoder/cuda_splatting.py:101-113  (GaussianRasterizationSettings)
 *   reference  ggrt/model/pixelsplat/decoder/cuda_splatting.py:114-125  (GaussianRasterizer.forward)
 *   reference  train_ggrt_stable.py:143                                 (loss.backward() → rasterizer backward)
 *
 * In the reference those lines bind (through pybind11) to the third-party CUDA extension
 * `diff_gaussian_rasterization._C` with the two entry points `rasterize_gaussians` and
 * `rasterize_gaussians_backward` (+ `mark_visible`, never called by GGRt).  The functions below
 * replace exactly those entry points; everything is plain C (pointers, sizes, POD structs), no
 * torch / C++ types, no exceptions across the boundary, no global mutable state (autograd calls
 * backward from a different thread — SURVEY.md §8b).
 *
 * Ownership: every buffer is owned by the caller (in the PyTorch binding: torch tensors).  The
 * library never hipMalloc's.  The only dynamically sized buffer (binning) is obtained through the
 * caller's GgrAllocFn after the 4-byte num_rendered readback (the single host sync of forward).
 *
 * All device pointers must be valid on the device that `stream` belongs to; fp32 unless noted;
 * arrays are dense row-major.  All work is enqueued on `stream` (a hipStream_t passed as void*).
 *
 * Return value: 0 on success, a GGR_E_* code otherwise; ggr_last_error() returns a thread-local
 * message.
 *
 * Non-finite inputs (a contract of this build; the reference has none — there a NaN mean is culled by the near-plane test or
 * not, a NaN covariance reaches `(int)ceil(NaN)`, a NaN colour poisons every pixel it touches): a Gaussian with a NaN or an
 * infinity in its mean, covariance (scale / rotation), opacity, aux feature... in anything its projected geometry is computed
 * from, or in an EVALUATED SH coefficient / its precomputed colour, takes no part in the frame — radius 0, no contribution to
 * any pixel, zero gradient in every one of its inputs; so does one whose screen radius exceeds 2^30 px.  The other Gaussians
 * render as if it were absent; nothing faults or hangs.  (SH coefficients of bands that are not evaluated are never read.)
 * A non-finite camera matrix or upstream gradient is the caller's error: it reaches every Gaussian.
 */
#ifndef GGR_RASTER_H
#define GGR_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGR_ABI_VERSION 11

enum {
    GGR_OK = 0,
    GGR_E_INVALID = 1,   /* bad argument combination (e.g. both/neither of shs & colors_precomp) */
    GGR_E_HIP = 2,       /* a HIP runtime call or kernel launch failed */
    GGR_E_ALLOC = 3,     /* the allocator callback returned NULL */
    GGR_E_LIMIT = 4,     /* size beyond what the kernels index: P or N ≥ 2^31, > 2^24 tiles, width or height > 65 535
                            tiles (1 048 560 px: the packed tile rects hold 16-bit tile coordinates); the message of
                            ggr_last_error() names the limit that was hit.  (Until ABI 8 frames wider than 12 288 px were
                            refused: rows of more than 768 tiles are now counted in column windows.) */
    GGR_E_CAPACITY = 5   /* GgrForwardOut.capacity_is_hint: num_rendered exceeds the buffer that was brought along AND the
                            allocator could not supply the exact one (or returned NULL) — the outputs of this call are
                            void, repeat it in exact mode (or with a larger buffer) */
};

/* Mirrors the NamedTuple built at cuda_splatting.py:101-113 (field meaning identical). */
typedef struct GgrSettings {
    int32_t image_height;
    int32_t image_width;
    int32_t sh_degree;      /* D; bands 0..min(D, sh_max_degree) are evaluated, and never more than sh_stride holds
                               (GGRt passes D=4, M=25) */
    int32_t sh_stride;      /* M = coefficients per Gaussian in `shs` (0 with colors_precomp) */
    int32_t num_points;     /* P */
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    const float* bg;         /* device [3] */
    const float* viewmatrix; /* device [4,4]  = (extrinsics^-1)^T, row-vector convention */
    const float* projmatrix; /* device [4,4]  = viewmatrix @ P^T */
    const float* campos;     /* device [3] */
    int32_t prefiltered;     /* accepted, unused (as at the call site: False) */
    int32_t debug;           /* 1: synchronise + check after every kernel */
    const float* tanfov_dev; /* device float[2] or NULL.  When given it overrides tanfovx / tanfovy, so that a host
                                that derived them on the device (ggr_camera_setup) never has to read them back */
    int32_t sh_max_degree;   /* 0 = default (3).  3: bands 0..3 only, as the graphdeco rasterizer and its "w-depth"
                                forks do — coefficients 16.. are ignored and get zero gradient.  This is the family
                                the LIVE call site's signature belongs to (3-tuple return, no `debug` field:
                                cuda_splatting.py:101-118), hence the default.  4: the nine
                                degree-4 terms are evaluated and differentiated when D >= 4 and M >= 25 — for a
                                host whose installed rasterizer does evaluate band 4 (not verifiable in this build). */
    int32_t scissor[4];      /* x0, y0, x1, y1 in pixels, half-open; all zero = the whole image (upstream).  Extension for
                                the reference's deferred back-propagation loop (finetune_ggrt_stable.py:126-142), which
                                re-renders the WHOLE frame per crop cell and keeps one cell: only the 16x16 tiles that
                                overlap the window are binned and blended — inside them every output equals the
                                full-frame render bit for bit; pixels of other tiles come out as background
                                (final_T = 1, no contributors), and `radii` / visibility refer to the window (a
                                Gaussian that touches no window tile gets radius 0 and no gradient).  The backward
                                needs no scissor: it skips tiles and quadrants whose upstream gradient is all zero. */
    int32_t reference_rects; /* 0 (default): TIGHT tile rects — a Gaussian is listed only in the tiles that the bounding box of
                                its alpha >= 1/255 ellipse reaches (inside the reference's 3-sigma square).  Every output
                                (images, radii, all gradients) is bit-identical to the reference-rect build — the dropped
                                (Gaussian, tile) pairs are exactly pairs every pixel would `continue` past — only the
                                internal lists are shorter (num_rendered, n_contrib positions).  1: the reference's rects,
                                i.e. lists identical to the reference's 64-bit sort, entry for entry. */
    int32_t depth_sort;      /* ABI 10.  How the per-tile lists get their (depth, index) order — the same lists either way, bit
                                for bit (the reference: ONE 64-bit radix sort over all (tile, depth) keys).
                                GGR_DEPTH_SORT_GLOBAL (1): the P Gaussians are sorted by depth once, in front of the tile-list
                                build, which then walks them in that order.  GGR_DEPTH_SORT_PER_TILE (2): the tile lists are
                                built in index order and every tile's list is then sorted by depth, stably, in LDS — no global
                                dependency on the forward's critical path; a list of more than 8192 entries cannot be sorted
                                that way: ggr_forward then rebuilds the lists with the global sort inside the call (exact mode
                                and capacity_is_hint), or raises the overflow flag of ggr_forward_status (sync-free mode).
                                GGR_DEPTH_SORT_AUTO (0): per tile when the frame holds at most 256 (view, Gaussian) pairs per tile on
                                average, the caller's max_list_len guess (if any) is at most 4096, and the call is not
                                sync-free (a 1080p frame with 1 M Gaussians: 123 per tile, longest list 1 100: 3 % faster
                                fwd+bwd; 200 k Gaussians at 504 x 378: 26 %); global otherwise (GGRt's own 660-tile frames
                                hold lists of 4 000-6 000 entries: the global sort is 10 % faster there). */
} GgrSettings;
enum { GGR_DEPTH_SORT_AUTO = 0, GGR_DEPTH_SORT_GLOBAL = 1, GGR_DEPTH_SORT_PER_TILE = 2,
       /* ABI 11.  The global sort has two forms with the same result: three stable radix passes over the P keys, or ONE stable
          partition pass into <= 1024 equally full depth buckets + every bucket sorted in LDS by one workgroup (about half the
          time: a million keys in 50 instead of 100 us).  The bucket form is what GGR_DEPTH_SORT_GLOBAL (and AUTO, where it
          resolves to global) runs when the call has a read-back (exact mode, capacity_is_hint) and a segment holds at most
          2 M keys.  It can meet a bucket it cannot sort — more than 8192 DIFFERENT keys inside 1/4096 of the frame's depth
          range (keys that are all equal, a plane of constant depth, are fine) — and the call then builds the lists again with
          the three passes: same frame, about one binning (0.15 ms at 1 M Gaussians) later. */
       GGR_DEPTH_SORT_NO_BUCKETS = 0x100,       /* IN flag, OR-ed into any of the three above: never the bucket form */
       GGR_DEPTH_SORT_GLOBAL_3PASS = 0x101,     /* IN: GLOBAL | NO_BUCKETS.  OUT (depth_sort_used): three passes built the lists */
       GGR_DEPTH_SORT_GLOBAL_SLOW = 0x401,      /* OUT only: the bucket form built the lists, but the frame's depths are so concentrated
                                                   in a small part of its depth range (a few far outliers, the rest inside an octave)
                                                   that most buckets were overfull fine bins, sorted by a launch of few workgroups:
                                                   correct, and slower than the three passes — same advice as for _FELL_BACK */
       GGR_DEPTH_SORT_GLOBAL_FELL_BACK = 0x201  /* OUT only: the bucket form met such a bucket and the call sorted again in three
                                                   passes (complete and correct); a host that sees this for a shape does better
                                                   setting NO_BUCKETS for it for a while */ };

/* Inputs of GaussianRasterizer.forward (cuda_splatting.py:118-125).
 * Exactly one of {shs, colors_precomp} and one of {cov3D_precomp, (scales, rotations)}. */
typedef struct GgrForwardIn {
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or NULL */
    const float* colors_precomp; /* [P,3]   or NULL */
    const float* opacities;      /* [P] (the [P,1] tensor of the call site) */
    const float* scales;         /* [P,3]   or NULL */
    const float* rotations;      /* [P,4] (r,x,y,z) or NULL */
    const float* cov3D_precomp;  /* [P,6] (00,01,02,11,12,22) or NULL */
    const float* aux_precomp;    /* [P] or NULL — extension: a 4th per-Gaussian feature blended like a colour
                                    channel into out_depth (Σ aux·α·T).  NULL ⇒ the feature is view-space z.
                                    Lets a host get GGRt's depth pass (cuda_splatting.py:227-269) out of the
                                    SAME rasterization as the colour pass. */
    /* ---- input forms (all zero / NULL = upstream's forms above).  They replace the torch operations reference
     * render_cuda runs over the P-sized tensors before every call (cuda_splatting.py:66-77,116,124): applied on
     * load, chained through in backward, results equal to the unfused call site up to fp32 rounding. ---- */
    const float* input_scale;    /* device scalar s or NULL (= 1): means3D·s, cov3D·s², scales·s  (the 1/near
                                    renormalisation, :66-73).  Gradients are w.r.t. the UNSCALED inputs. */
    int32_t cov3D_full;          /* 1: cov3D_precomp is [P,3,3] row-major; entries (0,1,2,4,5,8) are used and
                                    dL_dcov3D is [P,3,3] with a zero lower triangle (the triu gather, :116,124) */
    int32_t sh_channel_major;    /* 1: shs (and dL_dshs) are [P,3,M] — GGRt's harmonics layout (:77 transposes it) */
    int32_t aux_affine;          /* 1 (with aux_precomp NULL): blended feature = max(aux_a + aux_b·z/s, 0), z = view
                                    depth — GGRt's depth-as-colour pass (:240-269) without a per-Gaussian tensor */
    float aux_a, aux_b;
} GgrForwardIn;

typedef struct GgrForwardOut {
    float* out_color;      /* [3,H,W] */
    int32_t* radii;        /* [P] */
    float* out_depth;      /* [H,W]  Σ f·α·T with f = view z (or aux_precomp): third value of the 3-tuple
                              unpacked at :118; may be NULL */
    void* geom_buffer;     /* ggr_geom_bytes(P) bytes, caller-allocated, kept for backward */
    void* image_buffer;    /* ggr_image_bytes(W,H) bytes, caller-allocated, kept for backward: tile ranges,
                              final T, contributor counts and — for images below 4096 tiles — the forward's
                              per-pixel checkpoints for the segmented backward (320 B per pixel) */
    void* binning_buffer;  /* OUT: what the allocator returned (kept by the caller for backward).
                              IN (sync-free mode): the caller's own list buffer, see binning_capacity */
    int64_t num_rendered;  /* OUT: Σ tiles touched = length of the sorted (tile, Gaussian) list; -1 in sync-free mode */
    float* stage_ms;       /* HOST float[GGR_FWD_STAGES] or NULL.  When given, every stage is bracketed
                              with hipEvents on `stream`, the call synchronises at the end and ADDS the
                              elapsed milliseconds per stage (profiling only; costs a sync). */
    int64_t binning_capacity; /* IN.  0: exact mode — one 4-byte read-back + host sync, then the allocator is asked for
                              exactly num_rendered entries.  > 0 together with a non-NULL binning_buffer of
                              ggr_binning_bytes(capacity) bytes: SYNC-FREE mode — no read-back, no host sync, no second
                              allocator call, so forward + backward can be captured in a hipGraph.  Lists that do
                              not fit are cut at the buffer's end and an overflow flag is raised on the device;
                              ggr_forward_status() reads count and flag whenever the caller chooses to sync. */
    int32_t no_backward;   /* IN.  1: this forward will never be followed by ggr_backward (inference / torch.no_grad()):
                              the per-pixel checkpoints of the segmented backward (images below 4096 tiles: 320 B per
                              pixel) are neither written nor needed, and image_buffer may be the smaller
                              ggr_image_bytes_inference() bytes; the per-pixel final transmittance / last contributor and
                              last contributor are not produced either (the blend kernel runs without the bookkeeping
                              a backward needs) and the per-tile replay bound is written as 0: a ggr_backward handed
                              such an image_buffer by mistake replays no list entry (zero blend gradients) rather than
                              reading uninitialised state.  0: as before. */
    void* backward_scratch; /* IN, optional.  The ggr_backward_scratch_bytes() buffer the caller will hand to this frame's
                              ggr_backward: the forward clears it on the side (inside the forward blend kernel, whose
                              memory pipe is idle) and the backward, told so by GgrBackwardIn.scratch_zeroed, skips its own
                              64-byte-per-Gaussian memset.  NULL: the backward clears it itself. */
    int32_t capacity_is_hint; /* IN, with binning_capacity > 0.  0: sync-free mode as described above.  1: EXACT mode with a
                              guess — the caller expects num_rendered ≤ binning_capacity (e.g. 1.25 × the previous frame's) and
                              brought a list buffer of that size: scatter and blend are enqueued behind the count WITHOUT
                              waiting for it, then the call waits for num_rendered alone (the exact mode's early read-back;
                              the device is busy with scatter and blend meanwhile) and returns it.  Fits: every output is what
                              the exact mode gives, and the host's latency after the read-back — alloc, two launches — is off
                              the device's critical path.  Does not fit: the call REPAIRS itself (ABI 9) — the allocator
                              is asked for ggr_binning_bytes(num_rendered) (its second call, as in the exact mode), scatter
                              and blend run once more on that buffer, and on return binning_buffer / binning_capacity name
                              the new buffer (the one to keep for ggr_backward) and this field reads 2 (IN/OUT).  Only if the
                              allocator returns NULL: GGR_E_CAPACITY, outputs void. */
    int32_t max_list_len;  /* ABI 10.  OUT: the longest tile list of the frame (-1 in sync-free mode: known on the device only).
                              IN, with capacity_is_hint = 1 and the per-tile depth sort: the longest list the caller expects (e.g.
                              1.25 x the previous frame's; 0 = no idea) — decides whether the launch for lists of 2049..8192
                              entries is enqueued up front.  A guess that was too small is repaired inside the call (those
                              lists are sorted and the frame blended once more). */
    int32_t depth_sort_used; /* ABI 10.  OUT: what built this frame's lists — GGR_DEPTH_SORT_PER_TILE, GGR_DEPTH_SORT_GLOBAL (ABI 11: its
                              bucket form), GGR_DEPTH_SORT_GLOBAL_3PASS or GGR_DEPTH_SORT_GLOBAL_FELL_BACK (ABI 11, below the enum) */
} GgrForwardOut;

/* stage indices for GgrForwardOut.stage_ms / GgrBackwardOut.stage_ms */
enum {
    GGR_FWD_PREPROCESS = 0, GGR_FWD_DEPTH_SORT = 1, GGR_FWD_TILE_COUNT = 2 /* counts + scans + the N readback */,
    GGR_FWD_TILE_SCATTER = 3, GGR_FWD_BLEND = 4,
    GGR_FWD_COLOUR = 5 /* ABI 9: the SH colour kernel's own duration when the forward runs it on its side stream BESIDE
                          stages 1-3 (then stage 0 is the geometry half alone and stage 4 contains whatever wait for the
                          colours was left); 0 when the per-Gaussian stage ran as one kernel.  Not a term of the forward's
                          duration: stages 0-4 add up to it */,
    GGR_FWD_TILE_SORT = 6 /* ABI 10: the per-tile depth sort (GgrSettings.depth_sort), behind the scatter; stage 1 is then 0 */,
    GGR_FWD_STAGES = 7
};
enum { GGR_BWD_CLEAR = 0, GGR_BWD_BLEND = 1, GGR_BWD_PREPROCESS = 2, GGR_BWD_STAGES = 3 };

/* Called TWICE per forward, in this order; must return device memory (256-byte aligned) or NULL:
 *   1st call  ggr_work_bytes(P,W,H) bytes: transient work area of the tile-list builder — the caller may
 *             release it as soon as ggr_forward has returned;
 *   2nd call  ggr_binning_bytes(num_rendered,…) bytes, after num_rendered is known: the tile lists, kept
 *             by the caller for backward (returned in GgrForwardOut.binning_buffer). */
typedef void* (*GgrAllocFn)(void* ctx, size_t bytes);

typedef struct GgrBackwardIn {
    GgrForwardIn fwd;            /* the same input pointers forward saw */
    const int32_t* radii;        /* [P] from forward */
    const void* geom_buffer;
    const void* image_buffer;
    const void* binning_buffer;
    int64_t num_rendered;
    const float* dL_dout_color;  /* [3,H,W] */
    const float* dL_dout_depth;  /* [H,W] or NULL (GGRt discards out_depth) */
    void* scratch;               /* ggr_backward_scratch_bytes(P) bytes, caller-allocated */
    int32_t scratch_zeroed;      /* 1: `scratch` was this frame's GgrForwardOut.backward_scratch and has not been used by
                                    a backward since (a second backward over the same forward must pass 0) */
} GgrBackwardIn;

/* Gradients in the order autograd returns them.  Buffers are overwritten
 * (no need to pre-zero).  NULL = not wanted / not applicable. */
typedef struct GgrBackwardOut {
    float* dL_dmeans3D;        /* [P,3] */
    float* dL_dmeans2D;        /* [P,3] (x,y in NDC units, z = 0) — the `mean_gradients` sink of :95 */
    float* dL_dshs;            /* [P,M,3] (zero for coefficients of bands that were not evaluated) or NULL */
    float* dL_dcolors_precomp; /* [P,3] or NULL */
    float* dL_dopacities;      /* [P] */
    float* dL_dcov3D;          /* [P,6]; always required (scratch for the scale/rot path too) */
    float* dL_dscales;         /* [P,3] or NULL */
    float* dL_drotations;      /* [P,4] or NULL */
    float* dL_daux;            /* [P]; required iff aux_precomp was given and dL_dout_depth != NULL */
    /* Extension beyond the reference: camera gradients.  The reference passes
     * the matrices inside a NamedTuple, which autograd does not differentiate; these three let
     * the host chain dL/d(extrinsics) = f(dL/dviewmatrix, dL/dprojmatrix, dL/dcampos).
     * All three NULL, or all three non-NULL. */
    float* dL_dviewmatrix;     /* [4,4] or NULL */
    float* dL_dprojmatrix;     /* [4,4] or NULL */
    float* dL_dcampos;         /* [3]   or NULL */
    float* stage_ms;           /* HOST float[GGR_BWD_STAGES] or NULL (see GgrForwardOut.stage_ms) */
} GgrBackwardOut;

int ggr_abi_version(void);
/* ABI 8: sha256 (64 hex digits) of the kernel sources + compiler flags this library was built from.  The Python
 * binding compares it with the csrc/ tree next to it and refuses a library built from anything else (a stale .so can
 * neither pass for a build nor be measured by accident).  No counterpart in the reference's extension. */
const char* ggr_source_hash(void);
const char* ggr_last_error(void);

size_t ggr_geom_bytes(int32_t num_points);
size_t ggr_image_bytes(int32_t width, int32_t height);
size_t ggr_binning_bytes(int64_t num_rendered, int32_t width, int32_t height);
size_t ggr_work_bytes(int32_t num_points, int32_t width, int32_t height);
size_t ggr_backward_scratch_bytes(int32_t num_points);

/* replaces diff_gaussian_rasterization._C.rasterize_gaussians */
int ggr_forward(const GgrSettings* settings, const GgrForwardIn* in, GgrForwardOut* out,
                GgrAllocFn alloc, void* alloc_ctx, void* stream);

/* replaces diff_gaussian_rasterization._C.rasterize_gaussians_backward */
int ggr_backward(const GgrSettings* settings, const GgrBackwardIn* in, GgrBackwardOut* out,
                 void* stream);

/* ---- V views of the SAME Gaussians in one launch set ------------------------------------------
 * The reference renders the views of a sample in a Python loop — one rasterizer call per view over a v× repeated copy
 * of the Gaussian tensors (decoder_splatting_cuda.py:40-60, cuda_splatting.py:93-127), each call with its own
 * preprocess, sort and blend launches, and autograd then adds the V per-view gradient tensors.  Here the P Gaussians
 * are read ONCE for all views (their SH rows stay in LDS while the cameras change), the V·P depth keys go through ONE
 * sort, the tiles of the views are stacked into one tile-list build and one blend launch (a 480×352 frame has 660
 * tiles — four views fill the chip where one cannot), and the backward returns the gradients already summed over
 * the views.  Results per view equal ggr_forward / ggr_backward's: same lists, bit-identical images. */
typedef struct GgrViews {
    int32_t num_views;           /* V >= 1;  V·P < 2^31, V·ceil(H/16) <= 65535, V·tiles <= 2^24 */
    const float* viewmatrix;     /* device [V,4,4] */
    const float* projmatrix;     /* device [V,4,4] */
    const float* campos;         /* device [V,3] */
    const float* bg;             /* device [V,3] */
    const float* tanfov;         /* device [V,2] (tanfovx, tanfovy) or NULL: settings->tanfovx/y for every view */
    const float* input_scale;    /* device [V] or NULL (see GgrForwardIn.input_scale, which is ignored here) */
    int32_t num_sets;            /* B: 0 / 1 = every view renders the same P Gaussians.  B > 1: the V views are B groups of
                                    V/B consecutive views and group b renders Gaussian SET b — every per-Gaussian input
                                    (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp) is then
                                    [B, P, …] with P = settings->num_points, and so is every per-Gaussian gradient of
                                    ggr_backward_views (summed over the views of the set).  This is the reference's
                                    `(b v)` flattening with per-batch-element Gaussians (decoder_splatting_cuda.py:40-60)
                                    as ONE launch set: one preprocess launch, one segmented sort, one tile-list build,
                                    one blend launch.  Everything per view ([V, …]: radii, images, aux_precomp, dL_dmeans2D,
                                    camera gradients) is unchanged. */
} GgrViews;

/* image_buffer size of a forward with GgrForwardOut.no_backward = 1 (no checkpoint area); num_views = 1 for ggr_forward */
size_t ggr_image_bytes_inference(int32_t width, int32_t height, int32_t num_views);
/* ABI 10: geom_buffer size of a forward with GgrForwardOut.no_backward = 1 — without the 48 B per (view, Gaussian) of the SH
 * colour's Jacobian that a training forward leaves for its backward (55 MB at GGRt's 1.15 M-Gaussian eval shape).  num_views = 1
 * for ggr_forward.  The full ggr_geom_bytes buffer is accepted as well. */
size_t ggr_geom_bytes_inference(int32_t num_points, int32_t num_views);

size_t ggr_geom_bytes_views(int32_t num_points, int32_t num_views);
size_t ggr_image_bytes_views(int32_t width, int32_t height, int32_t num_views);
size_t ggr_work_bytes_views(int32_t num_points, int32_t width, int32_t height, int32_t num_views);
size_t ggr_backward_scratch_bytes_views(int32_t num_points, int32_t num_views);

/* As ggr_forward with the camera fields of `settings` (bg, viewmatrix, projmatrix, campos, tanfov_dev) ignored in
 * favour of `views`.  Shapes: out_color [V,3,H,W], radii [V,P], out_depth [V,H,W]; aux_precomp, when given, [V,P];
 * buffers sized with the *_views queries; num_rendered counts the entries of all views. */
int ggr_forward_views(const GgrSettings* settings, const GgrViews* views, const GgrForwardIn* in, GgrForwardOut* out,
                      GgrAllocFn alloc, void* alloc_ctx, void* stream);

/* As ggr_backward.  dL_dout_color [V,3,H,W], dL_dout_depth [V,H,W] or NULL, radii [V,P].  Gradients w.r.t. the
 * Gaussians come out SUMMED over the views ([P,…]); dL_dmeans2D and dL_daux are per view ([V,P,3], [V,P]); the camera
 * gradients are per view ([V,4,4], [V,4,4], [V,3]). */
int ggr_backward_views(const GgrSettings* settings, const GgrViews* views, const GgrBackwardIn* in, GgrBackwardOut* out,
                       void* stream);

/* The per-view camera quantities of the call site in one launch (cuda_splatting.py:18-46,66-73,82-89 and
 * ggrt/geometry/projection.py:233-247): for each of n views  scale = scale_invariant ? 1/near : 1,
 * view = inverse(extrinsics with its translation·scale)^T, full = view @ P^T with GGRt's projection P (built from
 * intrinsics[0] for every view, near·scale, far·scale), campos, tan(fov/2) from the normalised intrinsics.
 * Everything stays on the device: feed tanfov to GgrSettings.tanfov_dev and scale to GgrForwardIn.input_scale. */
int ggr_camera_setup(int32_t n, const float* extrinsics /*[n,4,4] camera-to-world*/,
                     const float* intrinsics /*[n,3,3] normalised*/, const float* near /*[n]*/,
                     const float* far /*[n]*/, int32_t scale_invariant, float* viewmatrix /*[n,4,4]*/,
                     float* projmatrix /*[n,4,4]*/, float* campos /*[n,3]*/, float* tanfov /*[n,2]*/,
                     float* scale /*[n]*/, void* stream);

/* Sync-free mode: num_rendered and the overflow flag of the forward that filled `geom_buffer` (synchronises).
 * `num_points` is what sized that geom buffer: P for ggr_forward, P·V for ggr_forward_views.
 * Returns GGR_E_HIP if a look-back spin of the depth sort ran into its bound (GPU preempted or halted: the frame
 * is invalid), GGR_E_LIMIT if a sort key beyond 30 bits reached the sort.  The exact mode reports the same two
 * conditions from ggr_forward itself, with the same codes.
 * No counterpart in the reference: upstream always reads num_rendered back inside rasterize_gaussians.
 *
 * Depth order: the sort key is the float bits of the view depth less those of the near cull (0.2), 30 bits — any
 * depth below 6.8e37 keeps its exact order (ties by ascending index, as the reference's stable 64-bit sort);
 * Gaussians at or beyond 6.8e37 (incl. +inf) share the last key and are ordered by index among themselves.
 * A launch set of up to 64 views sorts one segment per view; beyond 64 views the views share one segment (same
 * lists, slower sort). */
int ggr_forward_status(const void* geom_buffer, int32_t num_points, int64_t* num_rendered, int32_t* overflow,
                       void* stream);

/* How the per-tile depth sort of the forward that filled `geom_buffer` went — WITHOUT a sync: queues a 16-byte copy on
 * `stream` into `host_words` (4 words of page-locked host memory, valid once everything queued on the stream so far has run):
 *   [0] num_rendered   [1] status bits (as ggr_forward_status)   [2] the longest tile list
 *   [3] the list entries of the tiles whose depths cluster in few key buckets — those tiles take the sort kernel's slow
 *       route (csrc/tile_sort.h, route 2).  0 after a forward that sorted globally.
 * A frame with most of its entries in [3] renders faster with depth_sort = GGR_DEPTH_SORT_GLOBAL (measured: NOTES r6,
 * "clustered depths"); AUTO cannot know that before the frame has been sorted once, so the host looks at a frame now and
 * then and chooses for the next ones of the same shape (ggrt_official_amd/rasterizer.py does: first and second call of a
 * shape, then every 64th).  No counterpart in the reference. */
int ggr_sort_stats_async(const void* geom_buffer, int32_t num_points, uint32_t* host_words, void* stream);

/* replaces diff_gaussian_rasterization._C.mark_visible: present[P] (uint8) = view z > 0.2 */
int ggr_mark_visible(int32_t num_points, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present, void* stream);

/* Introspection for tests: copies of forward intermediates out of the opaque buffers
 * (device → device on `stream`).  Any destination may be NULL. */
int ggr_debug_unpack_geom(const void* geom_buffer, int32_t num_points, float* depth /*[P]*/,
                          float* xy /*[P,2]*/, float* conic_opacity /*[P,4]*/, float* rgb /*[P,3]*/,
                          int32_t* tiles_touched /*[P]*/, uint8_t* clamped /*[P,3]*/, void* stream);
int ggr_debug_unpack_binning(const void* binning_buffer, const void* image_buffer, int64_t num_rendered,
                             int32_t width, int32_t height, uint32_t* point_list /*[N]*/,
                             int32_t* ranges /*[tiles,2]*/, float* final_T /*[H,W]*/,
                             int32_t* n_contrib /*[H,W]*/, void* stream);


/* Self-test of the exact mode's num_rendered wait (no GPU work, no counterpart in the reference): runs the host-side
 * wait loop of ggr_forward on a private word with an injected event-query result.
 * scenario 0: the query reports "not ready" and the word receives 1234 after a few polls → GGR_OK (*value = 1234);
 * scenario 1: the query reports an error status (a stream in error / a lost device) → GGR_E_HIP at once;
 * scenario 2: the query never becomes ready and the word never changes (a hung GPU) → GGR_E_HIP after timeout_s;
 * scenario 3: the query reports "done" but the word was never written → GGR_E_HIP;
 * scenario 4: the stream is busy with EARLIER work for 3·timeout_s before the tile-list kernels get their turn, then
 *             the word receives 4321 → GGR_OK: the time bound only runs from those kernels' turn. */
int ggr_debug_readback_wait(int32_t scenario, double timeout_s, uint32_t* value);

/* The streaming yardstick (no counterpart in the reference): a float4 copy of `bytes` (multiple of 16; both pointers
 * 16-byte aligned) from `src` to `dst` in `blocks` workgroups of 256 threads (0 = default).  bench.py times it to quote
 * what a pure HBM streaming kernel reaches on this part next to the 8 TB/s spec. */
int ggr_debug_copy(const void* src, void* dst, size_t bytes, int32_t blocks, void* stream);

/* Work counters of the two blend kernels on the current device since the last reset (no counterpart in the reference; dev builds
 * only — a library built without -DGGR_DEV_COUNTERS returns GGR_E_INVALID and zeros).  out[0..3]: forward — survivors the
 * quadrant culls listed, survivors walked, (survivor, pixel) pairs composited, batches culled; out[4..7]: backward — (quadrant,
 * entry) slots that survived, slots without a single valid pixel, valid (slot, pixel) pairs, batches culled.  Synchronises the device. */
int ggr_debug_counters(uint64_t* out /*[8]*/, int32_t reset);

/* Host-side slots the library has EVER allocated in this process (no counterpart in the reference): read-back slots (a pinned
 * line + two events each) and side-stream slots (a stream + four events each).  A host thread owns one of each per device it
 * renders on and returns them to a process-wide pool when it ends, so the numbers follow the largest number of threads that
 * rendered AT THE SAME TIME, not the number of threads that ever did.  Either pointer may be NULL.  No GPU work. */
int ggr_debug_host_slots(int32_t* readback_slots, int32_t* side_streams);

#ifdef __cplusplus
}
#endif
#endif /* GGR_RASTER_H */
