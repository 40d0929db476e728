#!/usr/bin/env python3
"""Generates tests/golden/callsite_*.npz by running the REFERENCE's own call site
(/root/reference/ggrt/model/pixelsplat/decoder/cuda_splatting.py: render_cuda, render_depth_cuda) on
seeded CPU inputs and recording every argument that reaches the rasterizer boundary, plus the images
the reference call site returns when the boundary is served by the CPU oracle.

Runs ONLY in the build container (needs /root/reference); nothing of the reference travels: the
fixtures are plain arrays (inputs + recorded boundary arguments + expected images).

How the import works: `jaxtyping` and `diff_gaussian_rasterization` are
absent here, so two stub modules are injected into sys.modules; the leaf files are loaded by path to
avoid `encoder/__init__` (torchvision / e3nn / torch.hub).
"""
import importlib.util
import os
import sys
import types
from typing import NamedTuple

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("GGR_GOLDEN_OUT", HERE)   # (tests/test_golden_provenance.py regenerates into a temporary directory)
sys.path.insert(0, ROOT)

from oracle import torch_raster as tr  # noqa: E402

RECORD = []
SH_CAP = [3]  # highest SH band the stand-in rasterizer evaluates for the case being recorded


class _Sub:
    def __getitem__(self, item):
        return object


def install_stubs():
    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int64", "Bool", "Shaped", "Int", "UInt8"):
        setattr(jt, n, _Sub())
    sys.modules["jaxtyping"] = jt

    dgr = types.ModuleType("diff_gaussian_rasterization")

    class GaussianRasterizationSettings(NamedTuple):
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool = False

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.rs = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            rs = self.rs
            RECORD.append(dict(
                image_height=rs.image_height, image_width=rs.image_width, tanfovx=float(rs.tanfovx),
                tanfovy=float(rs.tanfovy), bg=rs.bg.clone(), scale_modifier=float(rs.scale_modifier),
                viewmatrix=rs.viewmatrix.clone(), projmatrix=rs.projmatrix.clone(), sh_degree=int(rs.sh_degree),
                campos=rs.campos.clone(), prefiltered=bool(rs.prefiltered), means3D=means3D.clone(),
                means2D_shape=tuple(means2D.shape), means2D_requires_grad=bool(means2D.requires_grad),
                opacities=opacities.clone(), shs=None if shs is None else shs.clone(),
                colors_precomp=None if colors_precomp is None else colors_precomp.clone(),
                cov3D_precomp=cov3D_precomp.clone()))
            color, radii, depth = tr.rasterize(
                means3D, opacities, rs.viewmatrix, rs.projmatrix, rs.campos, rs.bg, rs.image_width, rs.image_height,
                rs.tanfovx, rs.tanfovy, rs.sh_degree, shs=shs, colors_precomp=colors_precomp,
                cov3D_precomp=cov3D_precomp, sh_cap=SH_CAP[0])
            return color, radii, depth

    dgr.GaussianRasterizationSettings = GaussianRasterizationSettings
    dgr.GaussianRasterizer = GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = dgr


def load_reference():
    sys.path.insert(0, REF)
    import ggrt.geometry.projection  # noqa: F401  (imports cleanly with the jaxtyping stub)
    for name in ("ggrt.model", "ggrt.model.pixelsplat", "ggrt.model.pixelsplat.decoder",
                 "ggrt.model.pixelsplat.encoder", "ggrt.model.pixelsplat.encoder.epipolar"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    load("ggrt.model.pixelsplat.encoder.epipolar.conversions", "ggrt/model/pixelsplat/encoder/epipolar/conversions.py")
    return load("ggrt.model.pixelsplat.decoder.cuda_splatting", "ggrt/model/pixelsplat/decoder/cuda_splatting.py")


def random_pose(g, jitter=0.15):
    """camera-to-world: small random rotation + translation around the origin, looking down +z."""
    w = (torch.rand(3, generator=g) - 0.5) * 2 * jitter
    K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = torch.matrix_exp(K)
    T = torch.eye(4)
    T[:3, :3] = R
    T[:3, 3] = (torch.rand(3, generator=g) - 0.5) * 0.4
    return T


def make_inputs(seed, b, g_count, d_sh, h, w, near_vals, far_vals, cx=0.5, cy=0.5):
    g = torch.Generator().manual_seed(seed)
    extr = torch.stack([random_pose(g) for _ in range(b)])
    fx = 0.9 + 0.2 * torch.rand(1, generator=g).item()
    intr = torch.eye(3).repeat(b, 1, 1)
    intr[:, 0, 0] = fx
    intr[:, 1, 1] = fx * w / h
    intr[:, 0, 2] = cx
    intr[:, 1, 2] = cy
    near = torch.tensor(near_vals, dtype=torch.float32)
    far = torch.tensor(far_vals, dtype=torch.float32)
    # Gaussians in front of the cameras
    z = near.max() * (1.5 + 6 * torch.rand(g_count, generator=g))
    xy = (torch.rand(g_count, 2, generator=g) - 0.5) * z[:, None] * 1.0
    means = torch.cat([xy, z[:, None]], -1)[None].repeat(b, 1, 1)
    A = torch.randn(g_count, 3, 3, generator=g) * 0.04 * z[:, None, None]
    cov = (A @ A.transpose(1, 2) + 1e-5 * torch.eye(3))[None].repeat(b, 1, 1, 1)
    sh = torch.randn(g_count, 3, d_sh, generator=g) * 0.3
    sh[..., 0] = torch.randn(g_count, 3, generator=g)
    sh = sh[None].repeat(b, 1, 1, 1)
    op = (0.05 + 0.9 * torch.rand(g_count, generator=g))[None].repeat(b, 1)
    bgc = torch.rand(b, 3, generator=g)
    return dict(extrinsics=extr, intrinsics=intr, near=near, far=far, background_color=bgc, gaussian_means=means,
                gaussian_covariances=cov, gaussian_sh_coefficients=sh, gaussian_opacities=op)


def to_np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().numpy()
        elif v is None:
            continue
        else:
            out[k] = np.asarray(v)
    return out


def main():
    install_stubs()
    cs = load_reference()
    torch.manual_seed(0)
    cases = [
        # name, kwargs for make_inputs, image_shape, call
        # GGRt's own form (sh_degree = 4, 25 coefficients) under BOTH readings of what the replaced extension does
        # with band 4: cap 3 (the default) and cap 4
        ("color_d25_offcentre", dict(seed=1, b=2, g_count=300, d_sh=25, h=40, w=56, near_vals=[0.7, 1.3],
                                     far_vals=[60.0, 90.0], cx=0.46, cy=0.55), (40, 56), "color", {}),
        ("color_d25_offcentre_shcap4", dict(seed=1, b=2, g_count=300, d_sh=25, h=40, w=56, near_vals=[0.7, 1.3],
                                            far_vals=[60.0, 90.0], cx=0.46, cy=0.55), (40, 56), "color", {}),
        ("color_d16_noscale", dict(seed=2, b=1, g_count=250, d_sh=16, h=48, w=48, near_vals=[1.0],
                                   far_vals=[100.0]), (48, 48), "color", dict(scale_invariant=False)),
        ("color_d1", dict(seed=3, b=1, g_count=200, d_sh=1, h=32, w=48, near_vals=[2.0], far_vals=[50.0]),
         (32, 48), "color", {}),
        ("depth_depth", dict(seed=4, b=2, g_count=260, d_sh=1, h=36, w=44, near_vals=[0.8, 1.1],
                             far_vals=[40.0, 70.0]), (36, 44), "depth", dict(mode="depth")),
        ("depth_disparity", dict(seed=5, b=1, g_count=220, d_sh=1, h=32, w=32, near_vals=[1.0], far_vals=[80.0]),
         (32, 32), "depth", dict(mode="disparity")),
        ("depth_relative_disparity", dict(seed=6, b=1, g_count=220, d_sh=1, h=32, w=32, near_vals=[1.2],
                                          far_vals=[50.0]), (32, 32), "depth", dict(mode="relative_disparity")),
        # reference cuda_splatting.py:251-252: `minimum(near).maximum(far).log()` — as written this clamps every
        # depth to `far` (the reference's quirk); the golden pins exactly that
        ("depth_log", dict(seed=7, b=2, g_count=240, d_sh=1, h=32, w=40, near_vals=[0.9, 1.4],
                           far_vals=[30.0, 55.0]), (32, 40), "depth", dict(mode="log")),
    ]
    for name, ikw, shape, kind, extra in cases:
        RECORD.clear()
        SH_CAP[0] = 4 if name.endswith("_shcap4") else 3
        inp = make_inputs(**ikw)
        if kind == "color":
            out = cs.render_cuda(inp["extrinsics"], inp["intrinsics"], inp["near"], inp["far"], shape,
                                 inp["background_color"], inp["gaussian_means"], inp["gaussian_covariances"],
                                 inp["gaussian_sh_coefficients"], inp["gaussian_opacities"], **extra)
        else:
            out = cs.render_depth_cuda(inp["extrinsics"], inp["intrinsics"], inp["near"], inp["far"], shape,
                                       inp["gaussian_means"], inp["gaussian_covariances"], inp["gaussian_opacities"],
                                       **extra)
        blob = {f"in_{k}": v for k, v in to_np(inp).items()}
        blob["image_shape"] = np.asarray(shape)
        blob["kind"] = np.asarray(kind)
        blob["extra_keys"] = np.asarray(list(extra.keys()))
        blob["extra_vals"] = np.asarray([str(v) for v in extra.values()])
        blob["n_views"] = np.asarray(len(RECORD))
        blob["sh_cap"] = np.asarray(SH_CAP[0])
        for i, rec in enumerate(RECORD):
            for k, v in to_np(rec).items():
                blob[f"view{i}_{k}"] = v
        blob["out_image"] = out.detach().numpy()
        path = os.path.join(OUT, f"callsite_{name}.npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: {len(RECORD)} boundary calls, out {tuple(out.shape)}, {os.path.getsize(path) / 1024:.0f} KiB")


def deferred_backprop_golden(cs, sh_cap=3):
    """The call pattern of the reference's fine-tune loop (finetune_ggrt_stable.py:112-142, "deferred
    back-propagation"): render the whole frame without a graph, take dL/d(rgb) of the image loss, then for every
    cell (i, j) of a crop_size × crop_size grid render again WITH a graph, slice the cell out of the image and
    call `.backward(rgb_pred_grad[cell])` on the slice.  `random_crop` at :31-43 leaves the target camera alone, so
    at the rasterizer boundary each cell is a full-frame forward whose backward sees an upstream gradient that is
    zero outside the cell.  Recorded: the input gradients each cell produces when the boundary is served by the
    oracle (torch autograd)."""
    RECORD.clear()
    SH_CAP[0] = sh_cap
    h, w, crop = 48, 64, 2
    inp = make_inputs(seed=11, b=1, g_count=320, d_sh=25, h=h, w=w, near_vals=[0.8], far_vals=[70.0], cx=0.47, cy=0.54)
    g = torch.Generator().manual_seed(99)
    target = torch.rand(1, 3, h, w, generator=g)
    names = ("gaussian_means", "gaussian_covariances", "gaussian_sh_coefficients", "gaussian_opacities")

    def render(leaves):
        return cs.render_cuda(inp["extrinsics"], inp["intrinsics"], inp["near"], inp["far"], (h, w),
                              inp["background_color"], *leaves)

    with torch.no_grad():
        rgb = render([inp[n] for n in names])
    rgb.requires_grad_(True)
    ((rgb - target) ** 2).mean().backward()  # MaskedL2ImageLoss without a mask (reference ggrt/loss/criterion.py)
    rgb_pred_grad = rgb.grad
    oh, ow = h // crop, w // crop
    blob = {f"in_{k}": v for k, v in to_np(inp).items()}
    blob.update(image_shape=np.asarray((h, w)), crop_size=np.asarray(crop), sh_cap=np.asarray(sh_cap), target=target.numpy(),
                rgb=rgb.detach().numpy(), rgb_pred_grad=rgb_pred_grad.numpy())
    for i in range(crop):
        for j in range(crop):
            leaves = [inp[n].clone().requires_grad_(True) for n in names]
            patch = render(leaves)[:, :, oh * i: oh * (i + 1), ow * j: ow * (j + 1)]
            patch.backward(rgb_pred_grad[:, :, oh * i: oh * (i + 1), ow * j: ow * (j + 1)])
            for n, t in zip(names, leaves):
                blob[f"cell{i}{j}_grad_{n}"] = t.grad.numpy()
    name = "deferred_backprop_d25" + ("_shcap4" if sh_cap == 4 else "")
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name}: {crop * crop} cells, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
    install_stubs()
    cs = load_reference()
    deferred_backprop_golden(cs, sh_cap=3)
    deferred_backprop_golden(cs, sh_cap=4)
