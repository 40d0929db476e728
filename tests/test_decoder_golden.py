"""The build's `DecoderSplattingCUDA` (ggrt_official_amd/splatting.py) against golden vectors recorded from the
REFERENCE's own `DecoderSplattingCUDA.forward` / `render_depth`
(/root/reference/ggrt/model/pixelsplat/decoder/decoder_splatting_cuda.py:29-85, imported once in the build container
by tests/golden/make_decoder_golden.py): b = 2 DIFFERENT Gaussian sets × v = 3 DIFFERENT cameras, so a wrong `(b v)`
flattening or `repeat` order shows.

CPU: every boundary call, in the order the reference issued them (six colour calls, then six depth calls), and the
returned `color[b,v,3,h,w]` / `depth[b,v,h,w]` with the boundary served by the CPU oracle — for the build's
reference-shaped path and its fused colour + depth path.  `-m gpu`: the same images through the HIP rasterizer on all
three of the build's paths (reference-shaped, fused depth, launch set with fused inputs).
"""
import os

import numpy as np
import pytest
import torch

from ggrt_official_amd import splatting
from oracle import torch_raster as tr

PATH = os.path.join(os.path.dirname(__file__), "golden", "decoder_b2v3.npz")
MODES = [None, "depth", "log", "disparity"]


def _load():
    z = np.load(PATH, allow_pickle=False)
    inp = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, inp, tuple(int(v) for v in z["image_shape"])


CALLS = []


class _RecordingOracleRasterizer(torch.nn.Module):
    """TEST-ONLY stand-in for the HIP rasterizer (CPU oracle) that also records what reached the boundary."""

    def __init__(self, rs):
        super().__init__()
        self.rs = rs

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, aux_precomp=None):
        rs = self.rs
        CALLS.append(dict(image_height=rs.image_height, image_width=rs.image_width, tanfovx=float(rs.tanfovx),
                          tanfovy=float(rs.tanfovy), bg=rs.bg, scale_modifier=float(rs.scale_modifier),
                          viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix, sh_degree=int(rs.sh_degree),
                          campos=rs.campos, prefiltered=bool(rs.prefiltered), means3D=means3D, opacities=opacities,
                          shs=shs, cov3D_precomp=cov3D_precomp, means2D=means2D))
        return tr.rasterize(means3D, opacities, rs.viewmatrix, rs.projmatrix, rs.campos, rs.bg, rs.image_width,
                            rs.image_height, rs.tanfovx, rs.tanfovy, rs.sh_degree, shs=shs,
                            colors_precomp=colors_precomp, cov3D_precomp=cov3D_precomp, aux=aux_precomp,
                            sh_cap=int(getattr(rs, "sh_max_degree", 0) or 3))


def _gaussians(inp, dev="cpu"):
    return splatting.Gaussians(means=inp["means"].to(dev), covariances=inp["covariances"].to(dev),
                               harmonics=inp["harmonics"].to(dev), opacities=inp["opacities"].to(dev))


def _cameras(inp, dev="cpu"):
    return tuple(inp[k].to(dev) for k in ("extrinsics", "intrinsics", "near", "far"))


def test_golden_is_what_the_generator_describes():
    z, inp, shape = _load()
    b, v = inp["extrinsics"].shape[:2]
    assert (b, v) == (2, 3) and inp["harmonics"].shape[-1] == 25
    assert z["cap3_color"].shape == (b, v, 3, *shape) and z["depth_depth"].shape == (b, v, *shape)
    # the two Gaussian sets and all six cameras differ, band 4 matters, the three depth modes differ
    assert not np.allclose(inp["means"][0], inp["means"][1])
    assert len({z[f"colour_call{i}_viewmatrix"].tobytes() for i in range(b * v)}) == b * v
    assert np.abs(z["cap3_color"] - z["cap4_color"]).max() > 1e-2
    assert np.abs(z["depth_depth"] - z["disparity_depth"]).max() > 1e-2


def _check_call(got, z, prefix, fallback_prefix=None):
    def g(k):
        if f"{prefix}_{k}" in z.files:
            return z[f"{prefix}_{k}"]
        return z[f"{fallback_prefix}_{k}"]  # depth calls: same Gaussian tensors as the colour call of that view
    assert got["image_height"] == int(g("image_height")) and got["image_width"] == int(g("image_width"))
    assert got["sh_degree"] == int(g("sh_degree"))
    assert got["scale_modifier"] == float(g("scale_modifier")) and got["prefiltered"] == bool(g("prefiltered"))
    np.testing.assert_allclose(got["tanfovx"], float(g("tanfovx")), rtol=2e-6)
    np.testing.assert_allclose(got["tanfovy"], float(g("tanfovy")), rtol=2e-6)
    np.testing.assert_allclose(got["bg"].numpy(), g("bg"), atol=0)
    np.testing.assert_allclose(got["viewmatrix"].numpy(), g("viewmatrix"), atol=1e-6)
    np.testing.assert_allclose(got["projmatrix"].numpy(), g("projmatrix"), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got["campos"].numpy(), g("campos"), atol=0)
    np.testing.assert_allclose(got["means3D"].numpy(), g("means3D"), atol=0)
    np.testing.assert_allclose(got["opacities"].numpy(), g("opacities"), atol=0)
    assert got["opacities"].shape == g("opacities").shape
    np.testing.assert_allclose(got["cov3D_precomp"].numpy(), g("cov3D_precomp"), atol=0)
    np.testing.assert_allclose(got["shs"].detach().numpy(), g("shs"), rtol=1e-6, atol=1e-7)
    assert got["shs"].shape == g("shs").shape
    assert got["means2D"].shape == got["means3D"].shape and got["means2D"].requires_grad


@pytest.mark.parametrize("mode", MODES, ids=[m or "none" for m in MODES])
def test_boundary_calls_in_reference_order(monkeypatch, mode):
    """Reference-shaped path: the calls reaching the rasterizer, in order, equal what the reference's decoder issued:
    (b v)-ordered colour calls, then (b v)-ordered depth calls."""
    z, inp, shape = _load()
    monkeypatch.setattr(splatting, "GaussianRasterizer", _RecordingOracleRasterizer)
    monkeypatch.setattr(splatting, "SH_MAX_DEGREE", 3)
    CALLS.clear()
    dec = splatting.DecoderSplattingCUDA(fused_depth=False, fused_inputs=False)
    out = dec(_gaussians(inp), *_cameras(inp), shape, depth_mode=mode)
    n = 6
    assert len(CALLS) == (n if mode is None else 2 * n)
    for i in range(n):
        _check_call(CALLS[i], z, f"colour_call{i}")
    if mode is not None:
        for i in range(n):
            _check_call(CALLS[n + i], z, f"{mode}_call{i}", f"colour_call{i}")
    assert (out.depth is None) == (mode is None)


@pytest.mark.parametrize("cap", [3, 4])
@pytest.mark.parametrize("mode", MODES, ids=[m or "none" for m in MODES])
@pytest.mark.parametrize("fused_depth", [False, True], ids=["reference_shaped", "fused_depth"])
def test_images_match_reference_decoder_cpu(monkeypatch, cap, mode, fused_depth):
    z, inp, shape = _load()
    monkeypatch.setattr(splatting, "GaussianRasterizer", _RecordingOracleRasterizer)
    monkeypatch.setattr(splatting, "SH_MAX_DEGREE", cap)
    dec = splatting.DecoderSplattingCUDA(fused_depth=fused_depth, fused_inputs=False)
    out = dec(_gaussians(inp), *_cameras(inp), shape, depth_mode=mode)
    np.testing.assert_allclose(out.color.detach().numpy(), z[f"cap{cap}_color"], rtol=0, atol=2e-5)
    if mode is None:
        assert out.depth is None
    else:
        ref = z[f"{mode}_depth"]
        np.testing.assert_allclose(out.depth.detach().numpy(), ref, rtol=1e-5, atol=2e-5 * max(1.0, np.abs(ref).max()))


def test_render_depth_alone_cpu(monkeypatch):
    z, inp, shape = _load()
    monkeypatch.setattr(splatting, "GaussianRasterizer", _RecordingOracleRasterizer)
    dec = splatting.DecoderSplattingCUDA(fused_inputs=False)
    d = dec.render_depth(_gaussians(inp), *_cameras(inp), shape, mode="disparity")
    np.testing.assert_allclose(d.numpy(), z["disparity_depth"], rtol=1e-5, atol=2e-5)


PATHS = {"reference_shaped": dict(fused_depth=False, fused_inputs=False),
         "fused_depth": dict(fused_depth=True, fused_inputs=False),
         "launch_set": dict(fused_depth=True, fused_inputs=True)}


@pytest.mark.gpu
@pytest.mark.parametrize("cap", [3, 4])
@pytest.mark.parametrize("mode", MODES, ids=[m or "none" for m in MODES])
@pytest.mark.parametrize("path", list(PATHS))
def test_images_match_reference_decoder_hip(monkeypatch, cap, mode, path):
    z, inp, shape = _load()
    monkeypatch.setattr(splatting, "SH_MAX_DEGREE", cap)
    dev = "cuda:0"
    dec = splatting.DecoderSplattingCUDA(**PATHS[path]).to(dev)
    out = dec(_gaussians(inp, dev), *_cameras(inp, dev), shape, depth_mode=mode)
    got, ref = out.color.detach().cpu().numpy(), z[f"cap{cap}_color"]
    d = np.abs(got - ref)
    assert got.shape == ref.shape
    assert (d > 1e-4).mean() <= 2e-4 and d.max() <= 0.02 * max(1.0, np.abs(ref).max())
    if mode is None:
        assert out.depth is None
    else:
        got, ref = out.depth.detach().cpu().numpy(), z[f"{mode}_depth"]
        scale = max(1.0, np.abs(ref).max())
        d = np.abs(got - ref) / scale
        assert got.shape == ref.shape
        assert (d > 1e-4).mean() <= 2e-4 and d.max() <= 0.02
