"""Largest shapes the kernels index (task brief: "maximum sizes"): a 4K frame of 4 M Gaussians (31 M list entries,
32 400 tiles), a frame that fills a tile-count wave's slots (768 tiles = 12 288 px) and wider ones (counted in column windows).
Checked by size-independent properties — the lists partition [0, N) and are (depth bits, id)-sorted, every Gaussian
is listed tiles_touched times, images and gradients are finite — as in test_gpu_full_size_properties.py; measured on
the MI355X box this round: 4K fwd+bwd 4.2 ms, 8K / 8 M Gaussians 17.9 ms, 12 288 × 2160 / 16 M Gaussians (124 M
entries, 25 GiB) 21.8 ms."""
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient

pytestmark = pytest.mark.gpu
dev = "cuda:0"


def _check_lists(sc, s):
    from ggrt_official_amd.rasterizer import debug_forward_state
    st = debug_forward_state(s.means3D, s.opacities, s.settings(), shs=s.shs, cov3D_precomp=s.cov3D)
    P = sc.means3D.shape[0]
    N = st["num_rendered"]
    ranges, tt = st["ranges"].long(), st["tiles_touched"].long()
    assert N == int(tt.sum()) and N > 0
    lens = ranges[:, 1] - ranges[:, 0]
    assert int(lens.sum()) == N and int(lens.min()) >= 0
    pl = st["point_list"].long()
    key = (st["depth"][pl].view(torch.int32).long() << 32) | pl
    tile_of = torch.repeat_interleave(torch.arange(ranges.shape[0], device=dev), lens)
    same = tile_of[1:] == tile_of[:-1]
    assert bool((key[1:][same] > key[:-1][same]).all())
    assert torch.equal(torch.bincount(pl, minlength=P), tt)
    assert bool(torch.isfinite(st["color"]).all())
    return N


def _fwd_bwd(s, W, H):
    from ggrt_official_amd import GaussianRasterizer
    leaves = [t.clone().requires_grad_() for t in (s.means3D, s.shs, s.opacities, s.cov3D)]
    m, sh, op, cov = leaves
    color, radii, _ = GaussianRasterizer(s.settings())(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sh,
                                                       cov3D_precomp=cov)
    color.backward(upstream_gradient(W, H, seed=1, device=dev))
    assert bool(torch.isfinite(color).all())
    for t in leaves:
        assert bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0
    return color, radii


def test_4k_frame_of_4m_gaussians():
    P, W, H = 4_000_000, 3840, 2160
    sc = make_scene(P, W, H, sh_degree=3, profile="A", seed=3)
    s = sc.to(dev)
    assert _check_lists(sc, s) > 4 * P
    c1, r1 = _fwd_bwd(s, W, H)
    c2, r2 = _fwd_bwd(s, W, H)
    assert torch.equal(c1, c2) and torch.equal(r1, r2)       # deterministic bit for bit


def test_frames_wider_than_a_count_waves_slots():
    """768 tiles per row (12 288 px) fill a tile-count wave's slots (ggr_common.h GGR_COUNT_SLOTS); wider rows are counted in
    column windows (tile_lists.hip bin_count_kernel, round 5 — until then GGR_E_LIMIT; the reference has no width limit).
    The widest one-window frame, and frames of two and three windows against the oracle's lists entry for entry."""
    P, W, H = 300_000, 768 * 16, 160
    sc = make_scene(P, W, H, sh_degree=1, profile="A", seed=5)
    s = sc.to(dev)
    _check_lists(sc, s)
    _fwd_bwd(s, W, H)
    import numpy as np
    from ggrt_official_amd.rasterizer import debug_forward_state
    from tests.helpers import check_grads, check_image, hip_forward_backward, oracle_forward
    from oracle import c_oracle
    for W2, H2, P2 in ((769 * 16, 64, 20_000), (1537 * 16 + 5, 48, 30_000), (2000 * 16, 40, 30_000)):
        sc2 = make_scene(P2, W2, H2, sh_degree=1, profile="A", seed=6)
        s2 = sc2.to(dev)
        st = oracle_forward(sc2)
        out = debug_forward_state(s2.means3D, s2.opacities, s2.settings(), shs=s2.shs, cov3D_precomp=s2.cov3D)
        assert out["num_rendered"] == st.num_rendered > 0
        assert np.array_equal(out["point_list"].cpu().numpy().astype(np.uint32), st.point_list)
        assert np.array_equal(out["ranges"].cpu().numpy(), st.ranges)
        dL = upstream_gradient(W2, H2, seed=2)
        color, radii, _, grads = hip_forward_backward(sc2, dL)
        assert np.array_equal(radii, st.radii)
        check_image(color, st.color)
        check_grads(grads, c_oracle.backward(st, dL.numpy()), ["means3D", "means2D", "shs", "opacities", "cov3D_precomp"])


def test_width_beyond_the_packed_rect_is_refused():
    sc2 = make_scene(1000, 65536 * 16, 16, sh_degree=0, profile="A", seed=5)
    s2 = sc2.to(dev)
    from ggrt_official_amd import GaussianRasterizer
    with pytest.raises(RuntimeError, match="image width .* exceeds"):
        GaussianRasterizer(s2.settings())(means3D=s2.means3D, means2D=torch.zeros_like(s2.means3D), opacities=s2.opacities,
                                          shs=s2.shs, cov3D_precomp=s2.cov3D)
