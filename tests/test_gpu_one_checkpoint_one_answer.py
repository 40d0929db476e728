"""One GGRt checkpoint, one answer: a d_sh = 25 scene rendered through BOTH documented integration
paths gives the same images —

  (A) GGRt's own call site on the import-name shim: `from diff_gaussian_rasterization import …` and settings built exactly as
      reference ``cuda_splatting.py:101-113`` builds them (no `sh_max_degree`: the field does not exist upstream);
  (B) the decoder swap: ``ggrt_official_amd.splatting.render_cuda`` / ``DecoderSplattingCUDA``.

Until round 5 (A) evaluated bands 0..3 (with a warning) and (B) band 4: 3.5e-4 apart."""
import math

import pytest
import torch

from ggrt_official_amd import splatting
from ggrt_official_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
dev = "cuda:0"


def _scene():
    sc = make_scene(20000, 160, 128, sh_degree=4, profile="B", seed=3).to(dev)
    assert sc.shs.shape[1] == 25
    sc.shs[:, 16:] *= 20.0                                   # band 4 visibly matters
    c2w = torch.eye(4, device=dev)[None]
    fx = 0.5 / math.tan(math.radians(30.0))
    K = torch.tensor([[[fx, 0, 0.5], [0, fx * 160 / 128, 0.5], [0, 0, 1]]], device=dev)
    near, far = torch.tensor([1.0], device=dev), torch.tensor([100.0], device=dev)
    cov = torch.zeros(1, 20000, 3, 3, device=dev)
    for k, (i, j) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
        cov[0, :, i, j] = cov[0, :, j, i] = sc.cov3D[:, k]
    return sc, c2w, K, near, far, cov


def _reference_shaped_call(sc, c2w, K, near, far, cov):
    """What GGRt's render_cuda does with the package it imports (cuda_splatting.py:49-128), on the shim."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    calls = splatting.boundary_arguments(c2w, K, near, far, (128, 160), torch.zeros(1, 3, device=dev), sc.means3D[None], cov,
                                         sc.shs.permute(0, 2, 1)[None], sc.opacities[None, :, 0], sh_max_degree=0)
    st, kw = calls[0]
    # the reference's own field list: nothing beyond upstream's NamedTuple
    settings = GaussianRasterizationSettings(
        image_height=st.image_height, image_width=st.image_width, tanfovx=st.tanfovx, tanfovy=st.tanfovy, bg=st.bg,
        scale_modifier=st.scale_modifier, viewmatrix=st.viewmatrix, projmatrix=st.projmatrix, sh_degree=st.sh_degree,
        campos=st.campos, prefiltered=False)
    assert settings.sh_max_degree == 0 and settings.sh_degree == 4
    image, radii, depth = GaussianRasterizer(settings)(means2D=torch.zeros_like(kw["means3D"]), **kw)
    return image


def test_shim_and_decoder_swap_render_the_same_images():
    sc, c2w, K, near, far, cov = _scene()
    prev = splatting.set_sh_max_degree(4)
    try:
        for cap in (4, 3):
            splatting.set_sh_max_degree(cap)
            a = _reference_shaped_call(sc, c2w, K, near, far, cov)
            b = splatting.render_cuda(c2w, K, near, far, (128, 160), torch.zeros(1, 3, device=dev), sc.means3D[None], cov,
                                      sc.shs.permute(0, 2, 1)[None], sc.opacities[None, :, 0])[0]
            g = splatting.Gaussians(means=sc.means3D[None], covariances=cov, harmonics=sc.shs.permute(0, 2, 1)[None],
                                    opacities=sc.opacities[None, :, 0])
            dec = splatting.DecoderSplattingCUDA(fused_inputs=False, fused_depth=False).to(dev)
            c = dec(g, c2w[None], K[None], near[None], far[None], (128, 160)).color[0, 0]
            assert torch.equal(a, b) and torch.equal(a, c), cap
            if cap == 4:
                band4 = a
        assert (band4 - a).abs().max() > 1e-3          # the two caps do differ on this scene …
        # … and a decoder's own choice holds whatever the layer's default says
        dec3 = splatting.DecoderSplattingCUDA(fused_inputs=False, fused_depth=False, sh_max_degree=3).to(dev)
        splatting.set_sh_max_degree(4)
        assert torch.equal(dec3(g, c2w[None], K[None], near[None], far[None], (128, 160)).color[0, 0], a)
    finally:
        splatting.set_sh_max_degree(prev)
