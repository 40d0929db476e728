"""The deferred back-propagation pattern at the rasterizer boundary (reference finetune_ggrt_stable.py:126-142: per
crop cell a full-frame render whose backward sees a gradient that is zero outside the cell), `-m gpu`:

* the backward's zero-gradient window skip (blend_bwd.hip) is exact: a windowed gradient gives the same input
  gradients as the same window with a 1e-30 "keep-alive" gradient everywhere else (which defeats the skip and is far
  below fp32 resolution of any sum it enters), to summation order, and as the C oracle;
* the forward's `scissor` extension renders the window's tiles bit-identically to the full frame, leaves the other
  tiles as background, and its backward equals the full-frame windowed backward.
"""
import numpy as np
import pytest
import torch

from ggrt_official_amd import GaussianRasterizer
from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from oracle import c_oracle
from tests.helpers import check_grads, oracle_forward, rel_l2

pytestmark = pytest.mark.gpu
KEYS = ("means3D", "opacities", "shs", "cov3D_precomp")


def _run(sc, dL, dLd=None, scissor=None):
    dev = "cuda:0"
    s = sc.to(dev)
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    leaves = dict(means3D=leaf(s.means3D), opacities=leaf(s.opacities), shs=leaf(s.shs), cov3D_precomp=leaf(s.cov3D))
    sink = torch.zeros_like(leaves["means3D"], requires_grad=True)
    rs = s.settings() if scissor is None else s.settings()._replace(scissor=scissor)
    color, radii, depth = GaussianRasterizer(rs)(means2D=sink, **leaves)
    loss = (color * dL.to(dev)).sum()
    if dLd is not None:
        loss = loss + (depth * dLd.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    g = {k: v.grad.detach().cpu().numpy() for k, v in leaves.items()}
    g["means2D"] = sink.grad.detach().cpu().numpy()
    return color.detach().cpu().numpy(), depth.detach().cpu().numpy(), radii.cpu().numpy(), g


@pytest.mark.parametrize("P,W,H,with_depth", [(30000, 200, 136, False),      # < 4096 tiles: the segmented backward (checkpoints)
                                              (30000, 200, 136, True),
                                              (150000, 1040, 1040, False)])   # 4225 tiles: the unsegmented backward
def test_zero_gradient_window_skip_is_exact(P, W, H, with_depth):
    sc = make_scene(P, W, H, sh_degree=2, profile="B" if P <= 30000 else "A", seed=21)  # (B fills the first P pixels only)
    dL = upstream_gradient(W, H, seed=5)
    dLd = upstream_gradient(W, H, seed=6)[0] if with_depth else None
    # window: not tile-aligned, cuts through tiles and quadrants
    x0, y0, x1, y1 = W // 3 + 3, H // 4 + 5, 2 * W // 3 - 2, 3 * H // 4 + 1
    mask = torch.zeros(H, W)
    mask[y0:y1, x0:x1] = 1.0
    dL_w = dL * mask
    dL_alive = dL_w + (1.0 - mask) * 1e-30     # defeats the skip; 1e-30 vanishes in every fp32 sum it enters
    dLd_w = None if dLd is None else dLd * mask
    dLd_alive = None if dLd is None else dLd_w + (1.0 - mask) * 1e-30
    _, _, _, g_skip = _run(sc, dL_w, dLd_w)
    _, _, _, g_ref = _run(sc, dL_alive, dLd_alive)
    _, _, _, g_ref2 = _run(sc, dL_alive, dLd_alive)
    assert np.abs(g_skip["means3D"]).max() > 1e-9          # the window does hold Gaussians
    for k in KEYS + ("means2D",):
        noise = rel_l2(g_ref2[k], g_ref[k])          # run-to-run: the order of the float atomics
        assert rel_l2(g_skip[k], g_ref[k]) <= max(3.0 * noise, 2e-7), k
    if not with_depth and P <= 30000:
        st = oracle_forward(sc)
        ref = c_oracle.backward(st, dL_w.numpy())
        check_grads(g_skip, ref, ["means3D", "means2D", "shs", "opacities", "cov3D_precomp"])


def test_all_zero_gradient_gives_all_zero():
    sc = make_scene(20000, 160, 112, sh_degree=1, profile="B", seed=3)
    _, _, _, g = _run(sc, torch.zeros(3, 112, 160), torch.zeros(112, 160))
    for k, v in g.items():
        assert not np.any(v), k


@pytest.mark.parametrize("W,H", [(200, 136), (1040, 1040)])
def test_scissored_forward_and_backward(W, H):
    P = 30000 if W < 1000 else 150000
    sc = make_scene(P, W, H, sh_degree=2, profile="B" if W < 1000 else "A", seed=22)
    dL = upstream_gradient(W, H, seed=7)
    dLd = upstream_gradient(W, H, seed=8)[0]
    x0, y0, x1, y1 = W // 2 + 5, H // 4 + 2, W - 7, 3 * H // 4 - 3
    mask = torch.zeros(H, W)
    mask[y0:y1, x0:x1] = 1.0
    full_c, full_d, full_r, g_full = _run(sc, dL * mask, dLd * mask)
    sc_c, sc_d, sc_r, g_sc = _run(sc, dL * mask, dLd * mask, scissor=(x0, y0, x1, y1))
    # the window's TILES are rendered exactly as in the full frame …
    tx0, ty0, tx1, ty1 = x0 // 16 * 16, y0 // 16 * 16, min(W, -(-x1 // 16) * 16), min(H, -(-y1 // 16) * 16)
    assert np.array_equal(sc_c[:, ty0:ty1, tx0:tx1], full_c[:, ty0:ty1, tx0:tx1])
    assert np.array_equal(sc_d[ty0:ty1, tx0:tx1], full_d[ty0:ty1, tx0:tx1])
    # … and every other tile is background, depth 0
    out = np.ones((H, W), bool)
    out[ty0:ty1, tx0:tx1] = False
    bg = sc.bg.numpy()
    assert np.array_equal(sc_c[:, out], np.broadcast_to(bg[:, None], (3, int(out.sum()))))
    assert not np.any(sc_d[out])
    # visibility refers to the window: a subset of the frame's visible Gaussians, same radii where visible
    vis = sc_r > 0
    assert vis.sum() < (full_r > 0).sum() and np.array_equal(sc_r[vis], full_r[vis])
    # same gradients as the full-frame render of the same windowed upstream gradient
    for k in KEYS + ("means2D",):
        assert rel_l2(g_sc[k], g_full[k]) < 1e-6, k


def test_scissor_validation():
    sc = make_scene(100, 64, 48, sh_degree=0, seed=1).to("cuda:0")
    args = dict(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs,
                cov3D_precomp=sc.cov3D)
    with pytest.raises(RuntimeError, match="scissor"):
        GaussianRasterizer(sc.settings()._replace(scissor=(10, 10, 10, 20)))(**args)
    # a window beyond the image clips to nothing: background everywhere, no error
    c, r, d = GaussianRasterizer(sc.settings()._replace(scissor=(640, 480, 700, 500)))(**args)
    assert torch.equal(c, sc.bg[:, None, None].expand_as(c)) and not r.any()


def test_colour_and_depth_gradients_with_different_windows():
    """A pixel is skipped only when ALL of its upstream gradients are zero: colour gradient in one window, depth gradient
    in another (overlapping) one."""
    W, H = 200, 136
    sc = make_scene(30000, W, H, sh_degree=1, profile="B", seed=23)
    dL = upstream_gradient(W, H, seed=9)
    dLd = upstream_gradient(W, H, seed=10)[0]
    mc, md = torch.zeros(H, W), torch.zeros(H, W)
    mc[20:70, 30:120] = 1.0
    md[50:110, 90:180] = 1.0
    _, _, _, g = _run(sc, dL * mc, dLd * md)
    alive = (1.0 - torch.maximum(mc, md)) * 1e-30
    _, _, _, ref = _run(sc, dL * mc + alive, dLd * md + alive)
    for k in KEYS + ("means2D",):
        assert rel_l2(g[k], ref[k]) < 1e-6, k
    assert np.abs(g["means3D"]).max() > 1e-9


def test_scissor_with_several_views_and_sets():
    """The scissor applies to every view of a launch set (views of one set, and sets)."""
    from ggrt_official_amd import rasterize_views
    from ggrt_official_amd.synthetic import camera_matrices
    W, H, P = 160, 112, 8000
    scs = [make_scene(P, W, H, sh_degree=2, seed=40 + b) for b in range(2)]
    cams = [camera_matrices(W, H) for _ in range(4)]
    dev = "cuda:0"
    view = torch.stack([c[0] for c in cams]).to(dev); proj = torch.stack([c[1] for c in cams]).to(dev)
    cam = torch.stack([c[2] for c in cams]).to(dev)
    tf = torch.tensor([[c[3], c[4]] for c in cams], dtype=torch.float32, device=dev)
    bg = torch.rand(4, 3).to(dev)
    stack = lambda f: torch.stack([f(s) for s in scs]).to(dev)
    args = (stack(lambda s: s.means3D), stack(lambda s: s.opacities), view, proj, cam, bg, tf)
    kw = dict(shs=stack(lambda s: s.shs), cov3D_precomp=stack(lambda s: s.cov3D))
    rs = scs[0].to(dev).settings()
    full, _, _ = rasterize_views(*args, rs, **kw)
    win = (37, 20, 101, 77)
    part, radii, _ = rasterize_views(*args, rs._replace(scissor=win), **kw)
    tx0, ty0, tx1, ty1 = win[0] // 16 * 16, win[1] // 16 * 16, -(-win[2] // 16) * 16, -(-win[3] // 16) * 16
    assert torch.equal(part[:, :, ty0:ty1, tx0:tx1], full[:, :, ty0:ty1, tx0:tx1])
    out = torch.ones(H, W, dtype=torch.bool, device=dev)
    out[ty0:ty1, tx0:tx1] = False
    assert torch.equal(part[:, :, out], bg[:, :, None].expand(4, 3, int(out.sum())))
