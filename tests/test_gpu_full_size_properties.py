"""BASELINE.json's full sizes (C3: 1 M Gaussians at 1920×1080; C5′: GGRt's own shape), where the CPU oracle
would take minutes: size-independent properties of the HIP path instead.

* the per-tile lists partition [0, N), N = Σ tiles_touched, every list is sorted by (depth bits, id);
* forward is deterministic bit for bit; 0 ≤ final_T ≤ 1, images finite, Σ n_contrib ≤ N·256;
* backward is linear in the upstream gradient (g(a·dL₁ + dL₂) = a·g(dL₁) + g(dL₂) up to atomic order);
* a strip of tiles rendered by the (already oracle-checked) small-image path agrees with the same strip of the
  full frame: cropping the image to its top-left corner must not change those pixels."""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import CONFIGS, make_scene, upstream_gradient
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
dev = "cuda:0"


def _state(s):
    from ggrt_official_amd.rasterizer import debug_forward_state
    return debug_forward_state(s.means3D, s.opacities, s.settings(), shs=s.shs, cov3D_precomp=s.cov3D)


@pytest.mark.parametrize("name", ["C3", "C5p", "C6p"])   # C6p: 4.9 M keys = 1200 sort tiles (beyond one resident wave of tiles)
def test_lists_partition_and_are_depth_sorted(name):
    sc = make_scene(**CONFIGS[name])
    s = sc.to(dev)
    st = _state(s)
    N = st["num_rendered"]
    ranges = st["ranges"].long()
    tiles_touched = st["tiles_touched"].long()
    assert N == int(tiles_touched.sum()) and N > 2 * sc.means3D.shape[0]   # (tight rects: C5′ 3.62 M → 2.80 M entries)
    lens = ranges[:, 1] - ranges[:, 0]
    nz = lens > 0
    starts = ranges[nz, 0]
    assert int(lens.sum()) == N
    assert bool((starts[1:] == starts[:-1] + lens[nz][:-1]).all()) and int(starts[0]) == 0
    pl = st["point_list"].long()
    assert pl.numel() == N and int(pl.min()) >= 0 and int(pl.max()) < sc.means3D.shape[0]
    # (depth bits, id) non-decreasing inside every tile's run
    key = (st["depth"][pl].view(torch.int32).long() << 32) | pl
    tile_of = torch.repeat_interleave(torch.arange(ranges.shape[0], device=dev), lens)
    same = tile_of[1:] == tile_of[:-1]
    assert bool((key[1:][same] > key[:-1][same]).all())
    # every Gaussian appears exactly tiles_touched times
    assert torch.equal(torch.bincount(pl, minlength=sc.means3D.shape[0]), tiles_touched)
    assert float(st["final_T"].min()) >= 0.0 and float(st["final_T"].max()) <= 1.0
    assert bool(torch.isfinite(st["color"]).all())


def _fwd_bwd(s, dL):
    from ggrt_official_amd import GaussianRasterizer
    leaves = [t.clone().requires_grad_() for t in (s.means3D, s.shs, s.opacities, s.cov3D)]
    m, sh, op, cov = leaves
    color, radii, depth = GaussianRasterizer(s.settings())(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sh,
                                                           cov3D_precomp=cov)
    color.backward(dL)
    return color.detach(), radii, [t.grad for t in leaves]


def test_c3_forward_is_deterministic_and_backward_is_linear():
    sc = make_scene(**CONFIGS["C3"])
    s = sc.to(dev)
    d1 = upstream_gradient(sc.width, sc.height, seed=1, device=dev)
    d2 = upstream_gradient(sc.width, sc.height, seed=2, device=dev)
    c1, r1, g1 = _fwd_bwd(s, d1)
    c2, r2, g2 = _fwd_bwd(s, d2)
    assert torch.equal(c1, c2) and torch.equal(r1, r2)
    _, _, g12 = _fwd_bwd(s, 3.0 * d1 + d2)
    for a, b, c in zip(g1, g2, g12):
        assert float(c.abs().max()) > 0
        assert rel_l2((3.0 * a + b).cpu().numpy(), c.cpu().numpy()) < 2e-5


def test_c3_centre_crop_matches_the_full_frame():
    """Rendering only a 256×256 window around the image centre (tile-aligned offset, same focal length in pixels,
    principal point kept where it is in the full frame) must reproduce those pixels of the 1080p frame — ties
    this size to the small sizes the oracle checks.  (A window far off-axis would not: the Jacobian's frustum
    clamp ±1.3·tan(fov/2) is relative to the rendered image's own field of view — SURVEY Appendix A.5.)"""
    from ggrt_official_amd import GaussianRasterizer
    from ggrt_official_amd.rasterizer import GaussianRasterizationSettings
    sc = make_scene(**CONFIGS["C3"])
    s = sc.to(dev)
    full = GaussianRasterizer(s.settings())(means3D=s.means3D, means2D=torch.zeros_like(s.means3D), opacities=s.opacities,
                                            shs=s.shs, cov3D_precomp=s.cov3D)[0]
    W, H, w, h, ox, oy = sc.width, sc.height, 256, 256, 832, 416          # offsets are multiples of the tile size
    fx = W / (2 * sc.tanfovx)
    fy = H / (2 * sc.tanfovy)
    tanx, tany = w / (2 * fx), h / (2 * fy)
    near, far = 1.0, 100.0
    P = torch.zeros(4, 4, dtype=torch.float64)
    P[0, 0] = 2 * fx / w; P[1, 1] = 2 * fy / h
    P[0, 2] = 2 * (W / 2 - ox) / w - 1; P[1, 2] = 2 * (H / 2 - oy) / h - 1   # principal point inside the window
    P[3, 2] = 1; P[2, 2] = far / (far - near); P[2, 3] = -(far * near) / (far - near)
    proj = (s.viewmatrix.double().cpu() @ P.T).float().to(dev)
    rs = GaussianRasterizationSettings(image_height=h, image_width=w, tanfovx=tanx, tanfovy=tany, bg=s.bg, scale_modifier=1.0,
                                       viewmatrix=s.viewmatrix, projmatrix=proj, sh_degree=sc.sh_degree, campos=s.campos,
                                       prefiltered=False)
    crop = GaussianRasterizer(rs)(means3D=s.means3D, means2D=torch.zeros_like(s.means3D), opacities=s.opacities, shs=s.shs,
                                  cov3D_precomp=s.cov3D)[0]
    ref = full[:, oy:oy + h, ox:ox + w]
    d = (crop - ref).abs()
    assert float(ref.abs().mean()) > 0.05
    assert float((d > 1e-4).float().mean()) < 2e-3      # a handful of threshold pixels may flip (projection rounding)
    assert float(d.mean()) < 1e-5


def test_window_cells_add_up_at_full_size():
    """The deferred back-propagation pattern at C3 (round 3): the gradients of the four cells of a 2 × 2 grid — each
    a windowed upstream gradient, three of them through a scissored forward as well — add up to the whole-frame
    backward (linearity + exactness of the zero-gradient skip and of the scissor at the benchmark's size)."""
    from ggrt_official_amd import GaussianRasterizer
    sc = make_scene(**CONFIGS["C3"])
    s = sc.to(dev)
    W, H = sc.width, sc.height
    dL = upstream_gradient(W, H, seed=17, device=dev)
    _, _, whole = _fwd_bwd(s, dL)
    total = [torch.zeros_like(g, dtype=torch.float64) for g in whole]
    full_color = None
    for i in range(2):
        for j in range(2):
            x0, y0, x1, y1 = j * W // 2, i * H // 2, (j + 1) * W // 2, (i + 1) * H // 2
            mask = torch.zeros(H, W, device=dev)
            mask[y0:y1, x0:x1] = 1.0
            leaves = [t.clone().requires_grad_() for t in (s.means3D, s.shs, s.opacities, s.cov3D)]
            m, sh, op, cov = leaves
            rs = s.settings() if (i, j) == (0, 0) else s.settings()._replace(scissor=(x0, y0, x1, y1))
            color, _, _ = GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sh, cov3D_precomp=cov)
            if full_color is None:
                full_color = color.detach()
            else:   # inside its window a scissored render IS the full render
                assert torch.equal(color.detach()[:, y0:y1, x0:x1], full_color[:, y0:y1, x0:x1])
            color.backward(dL * mask)
            for t, g in zip(total, leaves):
                t += g.grad.double()
    for t, w in zip(total, whole):
        assert rel_l2(t.cpu().numpy(), w.double().cpu().numpy()) < 2e-6


def test_two_gaussian_sets_at_ggrt_shape_equal_two_calls():
    """GgrViews.num_sets at GGRt's training shape (C5′ × 2 different scenes): bit-identical images and radii, same
    gradients per set as two separate calls."""
    from ggrt_official_amd import GaussianRasterizer, rasterize_views
    cfg = CONFIGS["C5p"]
    scs = [make_scene(seed=s, **cfg).to(dev) for s in (0, 1)]
    W, H = scs[0].width, scs[0].height
    dLs = torch.stack([upstream_gradient(W, H, seed=30 + k, device=dev) for k in range(2)])
    single = []
    for k, s in enumerate(scs):
        c, r, g = _fwd_bwd(s, dLs[k])
        single.append((c, r, g))
    stack = lambda f: torch.stack([f(s) for s in scs]).clone().requires_grad_()
    m, sh, op, cov = stack(lambda s: s.means3D), stack(lambda s: s.shs), stack(lambda s: s.opacities), stack(lambda s: s.cov3D)
    view = torch.stack([s.viewmatrix for s in scs]); proj = torch.stack([s.projmatrix for s in scs])
    cam = torch.stack([s.campos for s in scs]); bg = torch.stack([s.bg for s in scs])
    tf = torch.tensor([[s.tanfovx, s.tanfovy] for s in scs], dtype=torch.float32, device=dev)
    color, radii, _ = rasterize_views(m, op, view, proj, cam, bg, tf, scs[0].settings(), shs=sh, cov3D_precomp=cov)
    color.backward(dLs)
    for k in range(2):
        assert torch.equal(color[k].detach(), single[k][0]) and torch.equal(radii[k], single[k][1])
        for got, want in zip((m.grad[k], sh.grad[k], op.grad[k], cov.grad[k]), single[k][2]):
            assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) < 2e-6
