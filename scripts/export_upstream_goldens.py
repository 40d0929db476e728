#!/usr/bin/env python3
"""Export golden vectors from the REAL `diff_gaussian_rasterization` extension (the one GGRt installs, reference
README.md:17-18, imported at ggrt/model/pixelsplat/decoder/cuda_splatting.py:6-9) into tests/golden/upstream_*.npz.

Why: that extension is the one thing this repository cannot hold — it is a third-party CUDA package, absent from the
reference tree and not buildable without CUDA — so the oracle's RASTERIZER ARITHMETIC is "parity unpinned" (DESIGN.md §3).
Any host where the extension imports (a CUDA box with GGRt's environment) closes the gap with one command:

    python scripts/export_upstream_goldens.py            # → tests/golden/upstream_*.npz  (≈ 10 MB in total)
    python -m pytest tests/test_upstream_goldens.py      # C oracle vs the files (CPU); add -m gpu for the HIP path

The scenes are the seeded ones of ggrt_official_amd/synthetic.py (inputs are stored in the files as well, so the
fixtures do not depend on the generator staying bit-stable): BASELINE config 1 (10 k Gaussians, 256², degree 0), a
degree-3 frame, the scale/rotation input path, precomputed colours, and GGRt's own form — `sh_degree = 4` with 25
coefficients per channel, which also settles what the extension does with band 4.

Only torch + numpy + the extension are needed; nothing here imports the HIP library or the oracle.  Both known call
signatures of the extension family are handled (settings with or without `debug`; 2- or 3-tuple return).
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name → make_scene arguments + which inputs the call uses
SCENES = {
    "c1_d0": dict(scene=dict(P=10000, W=256, H=256, sh_degree=0, profile="A", seed=0), geometry="cov", colour="sh"),
    "d3": dict(scene=dict(P=8000, W=208, H=144, sh_degree=3, profile="A", seed=1), geometry="cov", colour="sh"),
    "scale_rot_d1": dict(scene=dict(P=6000, W=160, H=112, sh_degree=1, profile="A", seed=2), geometry="scale_rot", colour="sh"),
    "precomp": dict(scene=dict(P=6000, W=160, H=112, sh_degree=0, profile="A", seed=3), geometry="cov", colour="precomp"),
    "ggrt_d4_m25": dict(scene=dict(P=8000, W=176, H=128, sh_degree=4, profile="B", seed=4), geometry="cov", colour="sh"),
}


def import_real_extension():
    """The installed CUDA extension — NOT this repository's import-name shim of the same name."""
    saved = list(sys.path)
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    sys.modules.pop("diff_gaussian_rasterization", None)
    try:
        mod = importlib.import_module("diff_gaussian_rasterization")
    finally:
        sys.path = saved
    where = os.path.abspath(getattr(mod, "__file__", "") or "")
    if where.startswith(ROOT + os.sep):
        raise ImportError("`diff_gaussian_rasterization` resolved to this repository's shim; the real extension is not installed")
    return mod


def make_settings(mod, sc, dev):
    fields = getattr(mod.GaussianRasterizationSettings, "_fields", ())
    kw = dict(image_height=sc.height, image_width=sc.width, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev),
              scale_modifier=1.0, viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev),
              sh_degree=sc.sh_degree, campos=sc.campos.to(dev), prefiltered=False)
    if "debug" in fields:
        kw["debug"] = False
    return mod.GaussianRasterizationSettings(**{k: v for k, v in kw.items() if not fields or k in fields})


def render_one(mod, name, spec, dev, upstream_gradient, make_scene):
    cfg = spec["scene"]
    sc = make_scene(cfg["P"], cfg["W"], cfg["H"], sh_degree=cfg["sh_degree"], profile=cfg["profile"], seed=cfg["seed"])
    dL = upstream_gradient(sc.width, sc.height, seed=100 + cfg["seed"])
    leaf = lambda t: t.detach().clone().to(dev).requires_grad_(True)
    means, op = leaf(sc.means3D), leaf(sc.opacities)
    means2D = torch.zeros_like(means, requires_grad=True)
    leaves = dict(means3D=means, opacities=op, means2D=means2D)
    call = dict(means3D=means, means2D=means2D, opacities=op)
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities)
    if spec["colour"] == "sh":
        leaves["shs"] = call["shs"] = leaf(sc.shs)
        inputs["shs"] = sc.shs
    else:
        g = torch.Generator().manual_seed(500 + cfg["seed"])
        colors = torch.rand(cfg["P"], 3, generator=g)
        leaves["colors_precomp"] = call["colors_precomp"] = leaf(colors)
        inputs["colors_precomp"] = colors
    if spec["geometry"] == "cov":
        leaves["cov3D_precomp"] = call["cov3D_precomp"] = leaf(sc.cov3D)
        inputs["cov3D_precomp"] = sc.cov3D
    else:
        leaves["scales"] = call["scales"] = leaf(sc.scales)
        leaves["rotations"] = call["rotations"] = leaf(sc.rotations)
        inputs["scales"], inputs["rotations"] = sc.scales, sc.rotations
    ret = mod.GaussianRasterizer(make_settings(mod, sc, dev))(**call)
    ret = ret if isinstance(ret, (tuple, list)) else (ret,)
    color, radii = ret[0], ret[1]
    depth = ret[2] if len(ret) > 2 and isinstance(ret[2], torch.Tensor) else None
    (color * dL.to(dev)).sum().backward()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    blob = {f"in_{k}": v.detach().cpu().numpy() for k, v in inputs.items()}
    blob.update(in_viewmatrix=sc.viewmatrix.numpy(), in_projmatrix=sc.projmatrix.numpy(), in_campos=sc.campos.numpy(),
                in_bg=sc.bg.numpy(), in_dL_dcolor=dL.numpy(),
                meta_size=np.asarray([sc.width, sc.height, sc.sh_degree], np.int64),
                meta_tanfov=np.asarray([sc.tanfovx, sc.tanfovy], np.float64),
                meta_scene=np.asarray([f"{k}={v}" for k, v in cfg.items()]),
                meta_return_len=np.asarray(len(ret)), meta_settings_fields=np.asarray(list(getattr(mod.GaussianRasterizationSettings, "_fields", ()))),
                meta_extension=np.asarray([str(getattr(mod, "__file__", "?")), str(getattr(mod, "__version__", "?")),
                                           torch.__version__, str(torch.version.cuda or torch.version.hip)]),
                out_color=color.detach().cpu().numpy(), out_radii=radii.detach().cpu().numpy().astype(np.int32))
    if depth is not None:
        blob["out_depth"] = depth.detach().cpu().numpy()
    for k, v in leaves.items():
        if v.grad is not None:
            blob[f"grad_{k}"] = v.grad.detach().cpu().numpy()
    return blob


def export(outdir: str, mod=None, device: str = None, names=None) -> list:
    sys.path.insert(0, ROOT)
    from ggrt_official_amd.synthetic import make_scene, upstream_gradient  # (pure torch; does not load the HIP library)
    mod = mod or import_real_extension()
    dev = torch.device(device or ("cuda:0" if torch.cuda.is_available() else "cpu"))
    os.makedirs(outdir, exist_ok=True)
    written = []
    for name, spec in SCENES.items():
        if names and name not in names:
            continue
        blob = render_one(mod, name, spec, dev, upstream_gradient, make_scene)
        path = os.path.join(outdir, f"upstream_{name}.npz")
        np.savez_compressed(path, **blob)
        written.append(path)
        print(f"wrote {path}: color {blob['out_color'].shape}, return tuple of {int(blob['meta_return_len'])}, "
              f"{sum(v.nbytes for v in blob.values()) / 1e6:.1f} MB raw")
    return written


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--device", default=None)
    ap.add_argument("--only", nargs="*", default=None, help=f"subset of: {', '.join(SCENES)}")
    a = ap.parse_args()
    export(a.out, device=a.device, names=a.only)
