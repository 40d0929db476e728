"""Every BASELINE.json configuration at its FULL size, HIP vs the C oracle (`-m gpu`).

The oracle's forward is OpenMP over pixel rows and its backward OpenMP over tiles (per-tile fp64 partial sums),
so a whole 1080p frame with 1 M Gaussians (C3: N = 10.76 M list entries) costs a few seconds of host time and the
benchmark's own workloads can be compared value for value — image, radii, num_rendered, the sorted lists, and all
five gradient tensors the reference's backward returns:

* C3   1 M Gaussians, 1920×1080, SH degree 3, profile A                  (BASELINE config 3, the headline)
* C5'  1 013 760 pixel-aligned Gaussians, 480×352, SH degree 4 / M = 25  (config 5's per-rank shape)
* C4'  1 146 880 pixel-aligned Gaussians, 448×320, SH degree 4 / M = 25  (config 4: GGRt's LLFF eval shape — the
       eval loop is forward-only, eval/eval_ggrt.py:317; gradients are compared anyway)
* C6'  4 915 200 pixel-aligned Gaussians, 960×640 (the Waymo eval shape, reference waymo.py:88-90): the scale check —
       1200 sort tiles (more than are resident at once), 2400 image tiles with depth segments
* C4' colour + depth through the call-site layer (`render_color_and_depth`-style: aux feature) is covered by
  tests/test_callsite_fused.py at small size and by the a4 goldens.

Bars: tests/helpers.py (PSNR ≥ 110 dB, ≤ 0.02 % threshold-flip pixels, gradients rel-L2 ≤ 1e-3 over all rows and ≤ 2e-5 once the 1e-5·P rows with the largest error (threshold flips; none below 100 k rows) are set aside).
"""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import CONFIGS, make_scene, upstream_gradient
from oracle import c_oracle
from tests.helpers import (FLIP_FRACTION, FWD_ATOL, check_grads, check_image, hip_forward_backward, oracle_forward,
                           record_metric)

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["C3", "C5p", "C4p", "C6p"])
def test_full_size_image_lists_and_gradients(name):
    sc = make_scene(seed=0, **CONFIGS[name])
    dL = upstream_gradient(sc.width, sc.height)
    st = oracle_forward(sc)
    ref = c_oracle.backward(st, dL.numpy())
    color, radii, depth, grads = hip_forward_backward(sc, dL)
    assert np.array_equal(radii, st.radii)
    check_image(color, st.color, tag=f"full:{name}")
    check_image(depth, st.out_depth, name="depth", tag=f"full:{name}:depth")
    check_grads(grads, ref, ["means3D", "means2D", "shs", "opacities", "cov3D_precomp"], tag=f"full:{name}")
    # the sorted per-tile lists and the per-pixel state of the same frame
    from ggrt_official_amd.rasterizer import debug_forward_state
    s = sc.to("cuda:0")
    out = debug_forward_state(s.means3D, s.opacities, s.settings(), shs=s.shs, cov3D_precomp=s.cov3D)
    assert out["num_rendered"] == st.num_rendered
    assert np.array_equal(out["point_list"].cpu().numpy().astype(np.uint32), st.point_list)
    assert np.array_equal(out["ranges"].cpu().numpy(), st.ranges)
    nc = (out["n_contrib"].cpu().numpy() != st.n_contrib).mean()
    ft = (np.abs(out["final_T"].cpu().numpy() - st.final_T) > FWD_ATOL).mean()
    record_metric(f"full:{name}:state", n_contrib_diff=nc, final_T_diff=ft)
    assert nc <= FLIP_FRACTION and ft <= FLIP_FRACTION


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["C3", "C5p"])
def test_full_size_tight_rects_against_the_reference_rects(name):
    """VERDICT r3 weak #1b: the test above compares the HIP default (tight tile rects) with the oracle in tight mode too —
    `tighten_rect` is the same formula on both sides, so a pair wrongly dropped at full size would be dropped by both.
    Here, at the headline size: (i) the HIP default is BIT-equal to the HIP build with the reference's rects in everything
    a caller sees (image, depth, radii, final_T); (ii) the HIP default matches the oracle run with the REFERENCE's rects —
    the restatement proper — in image, depth and all five gradient tensors; (iii) `reference_rects=True` builds the
    reference's lists, entry for entry, at this size."""
    from ggrt_official_amd.rasterizer import debug_forward_state
    sc = make_scene(seed=0, **CONFIGS[name])
    dL = upstream_gradient(sc.width, sc.height)
    st = oracle_forward(sc, tight=False)
    ref = c_oracle.backward(st, dL.numpy())
    color, radii, depth, grads = hip_forward_backward(sc, dL)
    color_r, radii_r, depth_r, grads_r = hip_forward_backward(sc, dL, reference_rects=True)
    assert np.array_equal(color, color_r) and np.array_equal(depth, depth_r) and np.array_equal(radii, radii_r)
    assert np.array_equal(radii, st.radii)
    check_image(color, st.color, tag=f"full-ref:{name}")
    check_image(depth, st.out_depth, name="depth", tag=f"full-ref:{name}:depth")
    keys = ["means3D", "means2D", "shs", "opacities", "cov3D_precomp"]
    check_grads(grads, ref, keys, tag=f"full-ref:{name}")
    check_grads(grads_r, ref, keys, tag=f"full-ref:{name}:refrects")
    s = sc.to("cuda:0")
    tight = debug_forward_state(s.means3D, s.opacities, s.settings(), shs=s.shs, cov3D_precomp=s.cov3D)
    full = debug_forward_state(s.means3D, s.opacities, s.settings()._replace(reference_rects=True), shs=s.shs,
                               cov3D_precomp=s.cov3D)
    assert torch.equal(tight["final_T"], full["final_T"]) and torch.equal(tight["color"], full["color"])
    assert full["num_rendered"] == st.num_rendered and tight["num_rendered"] < full["num_rendered"]
    assert np.array_equal(full["point_list"].cpu().numpy().astype(np.uint32), st.point_list)
    assert np.array_equal(full["ranges"].cpu().numpy(), st.ranges)
    assert np.array_equal(full["tiles_touched"].cpu().numpy(), st.tiles_touched)
    # every pixel's last contributor is the same Gaussian in both list forms (positions differ: the tight list is a sub-list)
    pl_t, pl_f = tight["point_list"].long(), full["point_list"].long()
    W, H = sc.width, sc.height
    gx = (W + 15) // 16
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda:0"), torch.arange(W, device="cuda:0"), indexing="ij")
    tile = (ys // 16) * gx + xs // 16
    for stt, pl in ((tight, pl_t), (full, pl_f)):
        nc = stt["n_contrib"].long()
        idx = (stt["ranges"][:, 0].long()[tile] + nc - 1).clamp(min=0, max=max(pl.numel() - 1, 0))
        stt["last_gaussian"] = torch.where(nc > 0, pl[idx], torch.full_like(nc, -1))
    assert torch.equal(tight["last_gaussian"], full["last_gaussian"])


@pytest.mark.timeout(900)
def test_c4p_forward_only_no_grad():
    """Config 4 as the eval loop runs it (eval/eval_ggrt.py:317: `torch.no_grad()`): forward only."""
    from ggrt_official_amd import GaussianRasterizer
    sc = make_scene(seed=1, **CONFIGS["C4p"])
    st = oracle_forward(sc)
    s = sc.to("cuda:0")
    with torch.no_grad():
        color, radii, depth = GaussianRasterizer(s.settings())(means3D=s.means3D, means2D=torch.zeros_like(s.means3D),
                                                               opacities=s.opacities, shs=s.shs, cov3D_precomp=s.cov3D)
    assert np.array_equal(radii.cpu().numpy(), st.radii)
    check_image(color.cpu().numpy(), st.color, tag="full:C4p:nograd")
