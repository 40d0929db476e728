"""Call-site layer: what GGRt wraps around the rasterizer.

Mirrors, with the same names / argument meaning / results, the reference's
``ggrt/model/pixelsplat/decoder/cuda_splatting.py`` (``get_projection_matrix`` :18-46,
``render_cuda`` :49-128, ``render_depth_cuda`` :227-269) and
``decoder/decoder_splatting_cuda.py`` (``DecoderSplattingCUDA`` :19-85), so that GGRt's ``PixelSplat``
(``pixelsplat.py:144,230``) can use this module unchanged.  Written from the behaviour of those
functions (pinned by the golden vectors under ``tests/golden/``), not from their text:
no einops / jaxtyping, all per-view quantities computed batched, no ``.item()`` syncs in the loop
(``tan(fov/2)`` is derived on the host side once for the whole batch).

Reference quirks kept on purpose (drop-in parity):
  * the projection matrix uses ``intrinsics[0]`` for EVERY batch element (``cuda_splatting.py:39-42``);
  * ``scale_invariant`` divides translations / means by ``near`` and covariances by ``near²`` (:66-73);
  * the depth pass feeds depth as a degree-0 SH coefficient, so the rasterizer returns
    ``0.5 + C0·z`` per channel and the result is the channel mean (:256-269);
  * ``sh_degree = isqrt(d_sh) - 1`` (GGRt: d_sh = 25 → 4); bands 0..min(sh_degree, cap) are evaluated.  The cap is
    per call (``sh_max_degree=`` of every function here; ``DecoderSplattingCUDA(sh_max_degree=…)`` per INSTANCE) and
    otherwise this layer's default ``SH_MAX_DEGREE`` (``set_sh_max_degree`` / ``GGR_SH_MAX_DEGREE``) — the default the
    ``diff_gaussian_rasterization`` import shim uses too, so that GGRt's own ``cuda_splatting.py`` on the shim and this
    module render one checkpoint identically (``tests/test_gpu_one_checkpoint_one_answer.py``).  The default
    is 4 for THIS layer unless chosen otherwise: the package GGRt's README installs is
    pixelSplat's rasterizer fork, GGRt's encoder emits, masks and Wigner-rotates all of bands 0..4
    (``encoder/common/gaussian_adapter.py:45-46,90``) and passes ``sh_degree = 4`` on purpose; 3 reproduces the
    graphdeco / w-depth family (coefficients 16.. ignored); the raw ``GaussianRasterizer`` keeps "not chosen → 3 with
    one warning".  What the choice moves on a GGRt-like scene is measured in INTEGRATION.md §7.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from math import isqrt
from typing import Literal, Optional

import torch
from torch import Tensor, nn

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

DepthRenderingMode = Literal["depth", "disparity", "relative_disparity", "log"]

# Highest SH band the rasterizer evaluates for this call site (GaussianRasterizationSettings.sh_max_degree).
# 4 (this layer's default): band 4 is evaluated and differentiated when the call passes sh_degree >= 4
# with >= 25 coefficients, as GGRt does; 3: coefficients 16.. are ignored (graphdeco / w-depth family); 0 hands the
# decision to the raw rasterizer ("not chosen": 3, with one warning the first time coefficients 16.. go unused).
SH_MAX_DEGREE = int(os.environ.get("GGR_SH_MAX_DEGREE", "4") or 4)


def set_sh_max_degree(cap: int) -> int:
    """Chooses the DEFAULT (process-wide: this call-site layer and the ``diff_gaussian_rasterization`` import shim) for the
    highest SH band the rasterizer evaluates: 3 or 4 (0 = leave it to the raw rasterizer: "not chosen", 3 with one warning).
    A per-call / per-instance ``sh_max_degree`` takes precedence.  Returns the previous setting."""
    global SH_MAX_DEGREE
    if int(cap) not in (0, 3, 4):
        raise ValueError("sh_max_degree must be 3 or 4")
    prev, SH_MAX_DEGREE = SH_MAX_DEGREE, int(cap)
    return prev


def resolve_sh_max_degree(cap: Optional[int] = None) -> int:
    """The cap a call uses: its own explicit choice, else this layer's default as it is NOW."""
    if cap is None:
        return SH_MAX_DEGREE
    if int(cap) not in (0, 3, 4):
        raise ValueError("sh_max_degree must be 3 or 4")
    return int(cap)


@dataclass
class Gaussians:
    """Same fields as reference ``ggrt/model/pixelsplat/types.py:7-12``."""
    means: Tensor        # [b, g, 3]
    covariances: Tensor  # [b, g, 3, 3]
    harmonics: Tensor    # [b, g, 3, d_sh]
    opacities: Tensor    # [b, g]
    # fused-adapter form (§8f-4): set ``covariances=None`` and give the ellipsoids as world-space
    # (scales, (w,x,y,z) quaternions) from ``adapter_scale_rotation``
    scales: Optional[Tensor] = None     # [b, g, 3]
    rotations: Optional[Tensor] = None  # [b, g, 4]


@dataclass
class DecoderOutput:
    color: Tensor            # [b, v, 3, h, w]
    depth: Optional[Tensor]  # [b, v, h, w]


def get_fov(intrinsics: Tensor) -> Tensor:
    """[b,3,3] normalised intrinsics → [b,2] (fov_x, fov_y): angle between the rays through the
    mid-points of opposite image edges (reference ``ggrt/geometry/projection.py:233-247``)."""
    inv = torch.linalg.inv(intrinsics)

    def ray(u, v):
        p = torch.tensor([u, v, 1.0], dtype=torch.float32, device=intrinsics.device)
        d = inv @ p
        return d / d.norm(dim=-1, keepdim=True)

    fov_x = (ray(0.0, 0.5) * ray(1.0, 0.5)).sum(-1).acos()
    fov_y = (ray(0.5, 0.0) * ray(0.5, 1.0)).sum(-1).acos()
    return torch.stack((fov_x, fov_y), dim=-1)


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor, intrinsics: Tensor) -> Tensor:
    """GGRt-modified perspective matrix [b,4,4] (reference ``cuda_splatting.py:18-46``): X/Y → (-1,1)
    with an off-centre principal point, Z → (0,1).  ``fov_*`` are accepted for signature parity but,
    as in the reference, only ``intrinsics[0]`` determines the X/Y rows."""
    b = near.shape[0]
    P = torch.zeros((b, 4, 4), dtype=torch.float32, device=near.device)
    k0 = intrinsics[0]
    P[:, 0, 0] = 2 * near * k0[0, 0]
    P[:, 1, 1] = 2 * near * k0[1, 1]
    P[:, 0, 2] = 2 * k0[0, 2] - 1
    P[:, 1, 2] = 2 * k0[1, 2] - 1
    P[:, 3, 2] = 1
    P[:, 2, 2] = far / (far - near)
    P[:, 2, 3] = -(far * near) / (far - near)
    return P


_TRIU = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))


def quaternion_to_matrix(q_xyzw: Tensor, eps: float = 1e-8) -> Tensor:
    """(x,y,z,w) quaternions (any norm) → rotation matrices (reference ``encoder/common/gaussians.py:8-31``)."""
    i, j, k, r = q_xyzw.unbind(-1)
    s = 2 / ((q_xyzw * q_xyzw).sum(-1) + eps)
    m = torch.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                     s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                     s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)], -1)
    return m.reshape(*q_xyzw.shape[:-1], 3, 3)


def adapter_covariances(scales: Tensor, rotations_xyzw: Tensor, c2w_rotations: Tensor) -> Tensor:
    """World-space covariance the way the reference's adapter builds it: ``C·R·S·Sᵀ·Rᵀ·Cᵀ``
    (``encoder/common/gaussians.py:33-44`` + ``gaussian_adapter.py:79-81``) → [...,3,3]."""
    L = c2w_rotations @ quaternion_to_matrix(rotations_xyzw) * scales[..., None, :]
    return L @ L.transpose(-1, -2)


def matrix_to_quaternion_wxyz(m: Tensor) -> Tensor:
    """Rotation matrices [...,3,3] → unit (w,x,y,z); picks the best-conditioned of the four branches."""
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    cand = torch.stack([
        torch.stack([1 + m00 + m11 + m22, m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]], -1),
        torch.stack([m[..., 2, 1] - m[..., 1, 2], 1 + m00 - m11 - m22, m[..., 0, 1] + m[..., 1, 0], m[..., 0, 2] + m[..., 2, 0]], -1),
        torch.stack([m[..., 0, 2] - m[..., 2, 0], m[..., 0, 1] + m[..., 1, 0], 1 - m00 + m11 - m22, m[..., 1, 2] + m[..., 2, 1]], -1),
        torch.stack([m[..., 1, 0] - m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1], 1 - m00 - m11 + m22], -1),
    ], -2)
    pick = torch.stack([m00 + m11 + m22, m00, m11, m22], -1).argmax(-1)
    q = torch.gather(cand, -2, pick[..., None, None].expand(*pick.shape, 1, 4)).squeeze(-2)
    return q / q.norm(dim=-1, keepdim=True)


def adapter_scale_rotation(scales: Tensor, rotations_xyzw: Tensor, c2w_rotations: Tensor, eps: float = 1e-8):
    """what the rasterizer needs INSTEAD of ``adapter_covariances`` — the same ellipsoid as
    (scales[...,3], world-space unit quaternion (w,x,y,z)[...,4]), 28 B per Gaussian instead of a 36-B matrix
    plus the [.,3,3] temporaries of three batched matmuls.  The camera-to-world rotation is composed onto
    the Gaussian's quaternion (Hamilton product); ``R S Sᵀ Rᵀ`` itself is then evaluated inside the HIP
    preprocess kernel (``scales``/``rotations`` inputs) and differentiated by its backward."""
    qc = matrix_to_quaternion_wxyz(c2w_rotations)
    x, y, z, w = (rotations_xyzw / (rotations_xyzw.norm(dim=-1, keepdim=True) + eps)).unbind(-1)
    cw, cx, cy, cz = qc.unbind(-1)
    q = torch.stack([cw * w - cx * x - cy * y - cz * z,
                     cw * x + cx * w + cy * z - cz * y,
                     cw * y - cx * z + cy * w + cz * x,
                     cw * z + cx * y - cy * x + cz * w], -1)
    return scales.broadcast_to(q.shape[:-1] + (3,)), q


def boundary_arguments(extrinsics, intrinsics, near, far, image_shape, background_color, gaussian_means,
                       gaussian_covariances, gaussian_sh_coefficients, gaussian_opacities, scale_invariant=True,
                       use_sh=True, gaussian_scales=None, gaussian_rotations=None, scissor=None, sh_max_degree=None):
    """Everything ``render_cuda`` hands to the rasterizer, batched: a list of
    (GaussianRasterizationSettings, kwargs) per view.  Split out so the golden-vector tests can
    compare it with what the reference's call site produces.

    ``gaussian_covariances=None`` selects the fused-adapter form (§8f-4): ``gaussian_scales[b,g,3]`` +
    world-space ``gaussian_rotations[b,g,4]`` (w,x,y,z) go to the rasterizer's ``scales``/``rotations``
    inputs; the scale-invariant renormalisation then multiplies the scales by 1/near (≡ cov × 1/near²)."""
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    fused_adapter = gaussian_covariances is None
    if fused_adapter and (gaussian_scales is None or gaussian_rotations is None):
        raise ValueError("pass gaussian_covariances, or gaussian_scales together with gaussian_rotations")
    if scale_invariant:
        scale = 1 / near
        extrinsics = extrinsics.clone()
        extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
        if fused_adapter:
            gaussian_scales = gaussian_scales * scale[:, None, None]
        else:
            gaussian_covariances = gaussian_covariances * (scale[:, None, None, None] ** 2)
        gaussian_means = gaussian_means * scale[:, None, None]
        near = near * scale
        far = far * scale
    d_sh = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(d_sh) - 1
    shs = gaussian_sh_coefficients.permute(0, 1, 3, 2).contiguous()  # [b, g, d_sh, 3]
    b = extrinsics.shape[0]
    h, w = image_shape
    fov = get_fov(intrinsics)
    tan_half = (0.5 * fov).tan()
    tan_host = tan_half.detach().cpu().tolist()  # ONE device→host copy for the whole batch
    proj = get_projection_matrix(near, far, fov[:, 0], fov[:, 1], intrinsics).transpose(1, 2)
    view = torch.linalg.inv(extrinsics).transpose(1, 2)
    full = view @ proj
    if not fused_adapter:
        cov6 = torch.stack([gaussian_covariances[:, :, i, j] for i, j in _TRIU], dim=-1)  # [b, g, 6]
    out = []
    for i in range(b):
        settings = GaussianRasterizationSettings(
            image_height=h, image_width=w, tanfovx=tan_host[i][0], tanfovy=tan_host[i][1],
            bg=background_color[i], scale_modifier=1.0, viewmatrix=view[i], projmatrix=full[i],
            sh_degree=degree, campos=extrinsics[i, :3, 3], prefiltered=False,
            sh_max_degree=resolve_sh_max_degree(sh_max_degree), **({} if scissor is None else {"scissor": tuple(scissor)}))
        kwargs = dict(means3D=gaussian_means[i], shs=shs[i] if use_sh else None,
                      colors_precomp=None if use_sh else shs[i, :, 0, :],
                      opacities=gaussian_opacities[i, ..., None])
        if fused_adapter:
            kwargs.update(scales=gaussian_scales[i], rotations=gaussian_rotations[i])
        else:
            kwargs.update(cov3D_precomp=cov6[i])
        out.append((settings, kwargs))
    return out


def _rasterize_views(calls, aux=None):
    """Runs the per-view rasterizer calls of a batch, serially like the reference's loop
    (``cuda_splatting.py:93-127``) but without its two ``.item()`` syncs per view.

    (measured and dropped in round 1: spreading the views over HIP streams gained nothing —
    4 views at 480×352: 2.79 ms serial vs 2.87 ms on 4 streams, 3.44 ms with one host thread per stream —
    because at GGRt's sizes a view is host-bound (≈ 270 µs of launches + the `num_rendered` read-back per
    forward); the multi-stream autograd path also needed stream-lifetime care that is not worth carrying for
    no gain.  The lever is a sync-free forward with fewer launches, NOTES.md (old §8).)"""
    outs = []
    for i, (settings, kw) in enumerate(calls):
        mean_gradients = torch.zeros_like(kw["means3D"], requires_grad=True)  # the `means2D` gradient sink
        extra = {} if aux is None else {"aux_precomp": aux[i]}
        outs.append(GaussianRasterizer(settings)(means2D=mean_gradients, **kw, **extra))
    return outs


def render_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape, background_color: Tensor,
                gaussian_means: Tensor, gaussian_covariances: Tensor, gaussian_sh_coefficients: Tensor,
                gaussian_opacities: Tensor, scale_invariant: bool = True, use_sh: bool = True,
                gaussian_scales: Optional[Tensor] = None, gaussian_rotations: Optional[Tensor] = None,
                scissor=None, sh_max_degree: Optional[int] = None) -> Tensor:
    """[batch] views → [batch,3,h,w] (reference ``cuda_splatting.py:49-128``).  With
    ``gaussian_covariances=None`` the ellipsoids come as scales + world quaternions (§8f-4).

    ``scissor=(x0, y0, x1, y1)`` (extension): render only the tiles overlapping that pixel window — for the
    fine-tune loop's deferred back-propagation (``finetune_ggrt_stable.py:126-142``), which renders the whole frame
    per crop cell and slices one cell out.  Inside the window the image equals the full render bit for bit."""
    calls = boundary_arguments(extrinsics, intrinsics, near, far, image_shape, background_color, gaussian_means,
                               gaussian_covariances, gaussian_sh_coefficients, gaussian_opacities, scale_invariant,
                               use_sh, gaussian_scales, gaussian_rotations, scissor, sh_max_degree)
    return torch.stack([o[0] for o in _rasterize_views(calls)])


def depth_to_relative_disparity(depth, near, far, eps: float = 1e-10):
    """0 at near, 1 at far (reference ``encoder/epipolar/conversions.py:17-27``)."""
    disp_near, disp_far, disp = 1 / (near + eps), 1 / (far + eps), 1 / (depth + eps)
    return 1 - (disp - disp_far) / (disp_near - disp_far + eps)


def depth_feature(extrinsics: Tensor, gaussian_means: Tensor, near: Tensor, far: Tensor, mode: DepthRenderingMode):
    """Camera-space z of every Gaussian, mapped as reference ``cuda_splatting.py:240-252`` maps it."""
    w2c = torch.linalg.inv(extrinsics)
    z = (gaussian_means @ w2c[:, 2, :3, None]).squeeze(-1) + w2c[:, 2, 3, None]
    if mode == "disparity":
        z = 1 / z
    elif mode == "relative_disparity":
        z = depth_to_relative_disparity(z, near[:, None], far[:, None])
    elif mode == "log":
        z = z.minimum(near[:, None]).maximum(far[:, None]).log()
    return z


def render_depth_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape,
                      gaussian_means: Tensor, gaussian_covariances: Tensor, gaussian_opacities: Tensor,
                      scale_invariant: bool = True, mode: DepthRenderingMode = "depth") -> Tensor:
    """Depth as colour, black background, channel mean → [batch,h,w] (reference ``cuda_splatting.py:227-269``)."""
    fake_color = depth_feature(extrinsics, gaussian_means, near, far, mode)
    b = fake_color.shape[0]
    result = render_cuda(extrinsics, intrinsics, near, far, image_shape,
                         torch.zeros((b, 3), dtype=fake_color.dtype, device=fake_color.device), gaussian_means,
                         gaussian_covariances, fake_color[:, :, None, None].expand(-1, -1, 3, 1), gaussian_opacities,
                         scale_invariant=scale_invariant)
    return result.mean(dim=1)


SH_C0 = 0.28209479177387814


def render_color_and_depth(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape,
                           background_color: Tensor, gaussian_means: Tensor, gaussian_covariances: Tensor,
                           gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor,
                           depth_mode: DepthRenderingMode = "depth", scale_invariant: bool = True,
                           use_sh: bool = True, gaussian_scales: Optional[Tensor] = None,
                           gaussian_rotations: Optional[Tensor] = None, sh_max_degree: Optional[int] = None):
    """ONE rasterization per view for what the reference obtains from two:
    ``render_cuda`` (colour, :49-128) + ``render_depth_cuda`` (:227-269).

    The reference's depth pass re-runs preprocess + sort + blend with the depth feature as a degree-0 SH
    coefficient, i.e. every Gaussian contributes ``max(0.5 + C0·f(z), 0)`` per channel over a black
    background, and the three identical channels are averaged.  Here that per-Gaussian value is handed to
    the rasterizer as its 4th blended feature (``aux_precomp``), so the depth image is the aux image of the
    SAME pass: identical values and gradients, half the work.  Returns ([b,3,h,w], [b,h,w])."""
    feat = depth_feature(extrinsics, gaussian_means, near, far, depth_mode)  # unscaled, as the reference
    aux = (0.5 + SH_C0 * feat).clamp(min=0.0)
    calls = boundary_arguments(extrinsics, intrinsics, near, far, image_shape, background_color, gaussian_means,
                               gaussian_covariances, gaussian_sh_coefficients, gaussian_opacities, scale_invariant,
                               use_sh, gaussian_scales, gaussian_rotations, None, sh_max_degree)
    outs = _rasterize_views(calls, aux=aux)
    return torch.stack([o[0] for o in outs]), torch.stack([o[2] for o in outs])


def render_views_fused(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape,
                       background_color: Tensor, gaussians: Gaussians, view_to_batch,
                       depth_mode: Optional[DepthRenderingMode] = None, scale_invariant: bool = True,
                       device_camera: bool = True, list_capacity: int = 0, batched: bool = True, scissor=None,
                       sh_max_degree: Optional[int] = None):
    """The call site with NO torch operation on a Gaussian-sized tensor (SURVEY.md §8 a2 "where time goes"):

    * ``device_camera``: view / projection matrices, camera position, tan(fov/2) and 1/near of all views come
      from one library kernel and stay on the device (no ``.item()`` / ``.cpu()`` — reference :104-105);
      with ``list_capacity > 0`` (sync-free forward) the whole call then runs without any host sync;
    * the Gaussians are not repeated per view (reference ``decoder_splatting_cuda.py:47-50``): view n reads
      batch element ``view_to_batch[n]`` of ``gaussians`` directly;
    * the 1/near renormalisation of means and covariances (``cuda_splatting.py:66-73``) travels as a device
      scalar (``input_scale``) and is applied when the kernel loads them;
    * ``harmonics`` stay ``[g,3,d_sh]`` (``sh_channel_major``) — no transpose copy (:77) and no copy back in
      backward; ``covariances`` stay ``[g,3,3]`` — the upper-triangle gather (:116,124) happens on load;
    * ``depth_mode="depth"``: the depth-as-colour feature ``max(0.5 + C0·z, 0)`` (:240-269) is formed inside the
      kernel from the view depth (``aux_affine``); the other modes still build it with torch.

    * ``batched``: the views that share a batch element go through ONE launch set (``rasterize_views``, SURVEY.md
      §8f-2): the Gaussians are read once for all of them, one depth sort, one tile-list build, one blend launch,
      gradients summed over the views inside the backward kernel — instead of one rasterizer call per view and
      autograd adding the per-view gradient tensors.  A batch element with a single view takes the per-view call.

    Same images and gradients as ``render_color_and_depth`` / ``render_cuda`` up to fp32 rounding
    (``tests/test_callsite_fused.py``).  extrinsics/intrinsics/near/far/background: one row per view.
    Returns (color [n,3,h,w], depth [n,h,w] | None)."""
    n = extrinsics.shape[0]
    h, w = image_shape
    d_sh = gaussians.harmonics.shape[-1]
    degree = isqrt(d_sh) - 1
    sh_cap = resolve_sh_max_degree(sh_max_degree)
    ext_orig = extrinsics
    # the per-view camera quantities: one library kernel, everything (incl. tan(fov/2) and 1/near) stays on the
    # device — or, for poses that carry gradients / CPU golden tests, the reference's torch formulation
    on_device = device_camera and extrinsics.is_cuda and not (extrinsics.requires_grad or intrinsics.requires_grad)
    if on_device:
        from .rasterizer import camera_setup
        view, full, campos, tanfov, scale = camera_setup(extrinsics, intrinsics, near, far, scale_invariant)
        tan_host = None
        if not scale_invariant:
            scale = None
    else:
        if scale_invariant:
            scale = 1 / near
            extrinsics = extrinsics.clone()
            extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
            near_s, far_s = near * scale, far * scale
        else:
            scale, near_s, far_s = None, near, far
        fov = get_fov(intrinsics)
        tan_host = (0.5 * fov).tan().detach().cpu().tolist()
        proj = get_projection_matrix(near_s, far_s, fov[:, 0], fov[:, 1], intrinsics).transpose(1, 2)
        view = torch.linalg.inv(extrinsics).transpose(1, 2)
        full = view @ proj
        campos, tanfov = extrinsics[:, :3, 3], None
    fused_cov = gaussians.covariances is not None
    # ---- every batch element in ONE launch set (GgrViews.num_sets): the reference's `(b v)` flattening with
    # per-batch-element Gaussians (decoder_splatting_cuda.py:40-60) without its per-view loop, its v× repeat, or a
    # Python loop over batch elements — view n renders Gaussian set n // v of the [b, g, …] tensors as they are
    nb = gaussians.means.shape[0]
    vpb = n // nb if nb and n % nb == 0 else 0
    if (batched and nb > 1 and vpb >= 1 and extrinsics.is_cuda and nb <= 64 and
            list(int(x) for x in view_to_batch) == [i // vpb for i in range(n)]):
        from .rasterizer import rasterize_views
        aux, aux_affine = None, None
        if depth_mode == "depth":
            aux_affine = (0.5, SH_C0)
        elif depth_mode is not None:
            feat = depth_feature(ext_orig, gaussians.means.repeat_interleave(vpb, 0), near, far, depth_mode)  # [n, g]
            aux = (0.5 + SH_C0 * feat).clamp(min=0.0)
        tf = tanfov if tanfov is not None else torch.tensor(tan_host, dtype=torch.float32, device=view.device)
        settings = GaussianRasterizationSettings(
            image_height=h, image_width=w, tanfovx=0.0, tanfovy=0.0, bg=background_color[0], scale_modifier=1.0,
            viewmatrix=view[0], projmatrix=full[0], sh_degree=degree, campos=campos[0], prefiltered=False,
            list_capacity=list_capacity * n, sh_channel_major=True, aux_affine=aux_affine,
            sh_max_degree=sh_cap, scissor=None if scissor is None else tuple(scissor))
        kw = dict(cov3D_precomp=gaussians.covariances) if fused_cov else dict(scales=gaussians.scales,
                                                                              rotations=gaussians.rotations)
        col, _, dep = rasterize_views(gaussians.means, gaussians.opacities, view, full, campos, background_color, tf,
                                      settings, shs=gaussians.harmonics, aux_precomp=aux, input_scale=scale, **kw)
        return col, (dep if depth_mode is not None else None)
    # batch element b of every Gaussian tensor WITHOUT `t[b]`: select's backward zero-fills a full [B,…] tensor
    # and copies the slice in, per view (0.2 ms per view for 1 M × 25 SH coefficients).  One unbind per tensor
    # (backward = one stack) — or a free reshape when there is a single batch element, GGRt's case.
    def per_batch(t: Optional[Tensor]):
        if t is None:
            return None
        return [t.reshape(t.shape[1:])] if t.shape[0] == 1 else list(t.unbind(0))
    g_means, g_cov, g_sh, g_op = (per_batch(gaussians.means), per_batch(gaussians.covariances),
                                  per_batch(gaussians.harmonics), per_batch(gaussians.opacities))
    g_scales, g_rot = per_batch(gaussians.scales), per_batch(gaussians.rotations)
    colors, depths = [None] * n, [None] * n
    groups = {}
    for i in range(n):
        groups.setdefault(int(view_to_batch[i]), []).append(i)
    single = []
    for b, idx in groups.items():
        if not (batched and len(idx) > 1 and extrinsics.is_cuda):
            single += idx
            continue
        # ---- all views of batch element b in one launch set ----
        from .rasterizer import rasterize_views
        ii = torch.as_tensor(idx, device=view.device)
        contiguous = idx == list(range(idx[0], idx[0] + len(idx)))
        take = (lambda t: t[idx[0]:idx[0] + len(idx)]) if contiguous else (lambda t: t.index_select(0, ii))
        aux, aux_affine = None, None
        if depth_mode == "depth":
            aux_affine = (0.5, SH_C0)
        elif depth_mode is not None:  # per-Gaussian feature per view, built with torch: [V,P]
            feat = depth_feature(take(ext_orig), g_means[b][None].expand(len(idx), -1, -1), take(near), take(far), depth_mode)
            aux = (0.5 + SH_C0 * feat).clamp(min=0.0)
        tf = take(tanfov) if tanfov is not None else torch.tensor([tan_host[i] for i in idx], dtype=torch.float32,
                                                                  device=view.device)
        settings = GaussianRasterizationSettings(
            image_height=h, image_width=w, tanfovx=0.0, tanfovy=0.0, bg=background_color[idx[0]], scale_modifier=1.0,
            viewmatrix=view[idx[0]], projmatrix=full[idx[0]], sh_degree=degree, campos=campos[idx[0]],
            prefiltered=False, list_capacity=list_capacity * len(idx), sh_channel_major=True, aux_affine=aux_affine,
            sh_max_degree=sh_cap, scissor=None if scissor is None else tuple(scissor))
        kw = dict(cov3D_precomp=g_cov[b]) if fused_cov else dict(scales=g_scales[b], rotations=g_rot[b])
        col, _, dep = rasterize_views(g_means[b], g_op[b][..., None], take(view), take(full), take(campos),
                                      take(background_color), tf, settings, shs=g_sh[b], aux_precomp=aux,
                                      input_scale=None if scale is None else take(scale), **kw)
        if len(idx) == n and contiguous:  # every view in this one launch set: hand its outputs on as they are
            return col, (dep if depth_mode is not None else None)
        for k, i in enumerate(idx):
            colors[i], depths[i] = col[k], dep[k]
    for i in single:
        b = int(view_to_batch[i])
        aux, aux_affine = None, None
        if depth_mode == "depth":
            aux_affine = (0.5, SH_C0)
        elif depth_mode is not None:  # disparity / relative_disparity / log: per-Gaussian feature built with torch
            feat = depth_feature(ext_orig[i:i + 1], g_means[b][None], near[i:i + 1], far[i:i + 1], depth_mode)
            aux = (0.5 + SH_C0 * feat[0]).clamp(min=0.0)
        settings = GaussianRasterizationSettings(
            image_height=h, image_width=w, tanfovx=tan_host[i][0] if tan_host else 0.0,
            tanfovy=tan_host[i][1] if tan_host else 0.0, bg=background_color[i],
            scale_modifier=1.0, viewmatrix=view[i], projmatrix=full[i], sh_degree=degree,
            campos=campos[i], prefiltered=False, list_capacity=list_capacity,
            input_scale=None if scale is None else scale[i:i + 1], sh_channel_major=True, aux_affine=aux_affine,
            tanfov=None if tanfov is None else tanfov[i], sh_max_degree=sh_cap,
            scissor=None if scissor is None else tuple(scissor))
        means = g_means[b]
        kw = dict(cov3D_precomp=g_cov[b]) if fused_cov else dict(scales=g_scales[b], rotations=g_rot[b])
        # means2D is only a gradient sink (`cuda_splatting.py:95-99`): its values are never read
        sink = torch.empty_like(means).requires_grad_()
        out = GaussianRasterizer(settings)(means3D=means, means2D=sink, opacities=g_op[b][..., None], shs=g_sh[b],
                                           aux_precomp=aux, **kw)
        colors[i], depths[i] = out[0], out[2]
    # one view (GGRt's usual call): a view of the rasterizer's output instead of a stack — no copy kernel forward,
    # none backward
    stack = lambda ts: ts[0].unsqueeze(0) if len(ts) == 1 else torch.stack(ts)
    return stack(colors), (stack(depths) if depth_mode is not None else None)


class DecoderSplattingCUDA(nn.Module):
    """Same call contract as reference ``decoder_splatting_cuda.py:19-85``:
    ``forward(gaussians, extrinsics[b,v,4,4], intrinsics[b,v,3,3], near[b,v], far[b,v], image_shape,
    depth_mode) -> DecoderOutput(color[b,v,3,h,w], depth[b,v,h,w] | None)``."""

    def __init__(self, cfg=None, fused_depth: bool = True, fused_inputs: bool = True, list_capacity: int = 0,
                 sh_max_degree: Optional[int] = None):
        super().__init__()
        self.cfg = cfg
        # 3 / 4: the explicit choice of INTEGRATION.md §7, for THIS decoder (two decoders of one process may differ); None =
        # this layer's default at the time of each call (set_sh_max_degree / GGR_SH_MAX_DEGREE)
        self.sh_max_degree = None if sh_max_degree is None else resolve_sh_max_degree(sh_max_degree)
        # > 0: sync-free rasterizer forward with per-tile lists of at most this many entries — with the fused
        # inputs the whole decoder call then has no host sync and can be captured in a HIP graph
        # (check ``ggrt_official_amd.last_forward_status()`` for overflow when a sync is affordable)
        self.list_capacity = int(list_capacity)
        self.fused_depth = fused_depth  # False: two rasterizations per view, literally as the reference
        self.fused_inputs = fused_inputs  # False: the reference's torch pre-processing of the Gaussian tensors
        self.register_buffer("background_color", torch.zeros(3, dtype=torch.float32), persistent=False)

    @staticmethod
    def _per_view(t: Tensor, v: int) -> Tensor:
        """[b, ...] → [(b v), ...] (every view sees the same Gaussians; the rasterizer only reads them)."""
        return t[:, None].expand(-1, v, *t.shape[1:]).reshape(-1, *t.shape[1:])

    @classmethod
    def _opt_per_view(cls, t: Optional[Tensor], v: int) -> Optional[Tensor]:
        return None if t is None else cls._per_view(t, v)

    @classmethod
    def _ellipsoids(cls, gaussians: Gaussians, v: int) -> dict:
        if gaussians.covariances is not None:
            return {}
        return dict(gaussian_scales=cls._per_view(gaussians.scales, v),
                    gaussian_rotations=cls._per_view(gaussians.rotations, v))

    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape, depth_mode: Optional[DepthRenderingMode] = None, scissor=None) -> DecoderOutput:
        """``scissor=(x0, y0, x1, y1)`` (extension, fused path): render only that pixel window's tiles — the
        deferred-backprop cell of ``finetune_ggrt_stable.py:126-142``."""
        b, v = extrinsics.shape[:2]
        if scissor is not None and not (self.fused_inputs and self.fused_depth):
            raise ValueError("scissor needs the fused call site (fused_inputs and fused_depth)")
        bg = self.background_color.to(far.device)[None].expand(b * v, 3)
        if self.fused_inputs and self.fused_depth:
            # no per-view copies of the Gaussians, no torch op on a Gaussian-sized tensor (render_views_fused)
            color, depth = render_views_fused(
                extrinsics.flatten(0, 1), intrinsics.flatten(0, 1), near.flatten(), far.flatten(), image_shape, bg,
                gaussians, [n // v for n in range(b * v)], depth_mode, list_capacity=self.list_capacity,
                scissor=scissor, sh_max_degree=self.sh_max_degree)
            return DecoderOutput(color.reshape(b, v, *color.shape[1:]),
                                 None if depth is None else depth.reshape(b, v, *depth.shape[1:]))
        if depth_mode is not None and self.fused_depth:
            color, depth = render_color_and_depth(
                extrinsics.flatten(0, 1), intrinsics.flatten(0, 1), near.flatten(), far.flatten(), image_shape, bg,
                self._per_view(gaussians.means, v), self._opt_per_view(gaussians.covariances, v),
                self._per_view(gaussians.harmonics, v), self._per_view(gaussians.opacities, v), depth_mode,
                sh_max_degree=self.sh_max_degree, **self._ellipsoids(gaussians, v))
            return DecoderOutput(color.reshape(b, v, *color.shape[1:]), depth.reshape(b, v, *depth.shape[1:]))
        color = render_cuda(extrinsics.flatten(0, 1), intrinsics.flatten(0, 1), near.flatten(), far.flatten(),
                            image_shape, bg, self._per_view(gaussians.means, v),
                            self._opt_per_view(gaussians.covariances, v), self._per_view(gaussians.harmonics, v),
                            self._per_view(gaussians.opacities, v), sh_max_degree=self.sh_max_degree,
                            **self._ellipsoids(gaussians, v))
        color = color.reshape(b, v, *color.shape[1:])
        depth = None if depth_mode is None else self.render_depth(gaussians, extrinsics, intrinsics, near, far,
                                                                  image_shape, depth_mode)
        return DecoderOutput(color, depth)

    def render_depth(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                     image_shape, mode: DepthRenderingMode = "depth") -> Tensor:
        b, v = extrinsics.shape[:2]
        result = render_depth_cuda(extrinsics.flatten(0, 1), intrinsics.flatten(0, 1), near.flatten(), far.flatten(),
                                   image_shape, self._per_view(gaussians.means, v),
                                   self._per_view(gaussians.covariances, v), self._per_view(gaussians.opacities, v),
                                   mode=mode)
        return result.reshape(b, v, *result.shape[1:])
