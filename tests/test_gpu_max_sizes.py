"""Largest shapes the kernels index (task brief: "maximum sizes"): a 4K frame of 4 M Gaussians (31 M list entries,
32 400 tiles), a frame as wide as the tile-list builder allows (768 tiles = 12 288 px), and the error beyond that.
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


def test_widest_frame_the_tile_lists_take_and_the_error_beyond():
    P, W, H = 300_000, 768 * 16, 160      # 768 tiles per row: a count wave's slots (ggr_common.h GGR_COUNT_SLOTS)
    sc = make_scene(P, W, H, sh_degree=1, profile="A", seed=5)
    s = sc.to(dev)
    _check_lists(sc, s)
    _fwd_bwd(s, W, H)
    sc2 = make_scene(1000, W + 16, 64, sh_degree=0, profile="A", seed=5)
    s2 = sc2.to(dev)
    from ggrt_official_amd import GaussianRasterizer
    with pytest.raises(RuntimeError, match="image width .* exceeds 12288 px"):
        GaussianRasterizer(s2.settings())(means3D=s2.means3D, means2D=torch.zeros_like(s2.means3D), opacities=s2.opacities,
                                          shs=s2.shs, cov3D_precomp=s2.cov3D)
