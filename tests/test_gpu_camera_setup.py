"""`camera_setup` (one library kernel for the call site's per-view camera quantities) against
the torch formulation that the golden vectors pin (`boundary_arguments`), and the whole decoder call —
camera kernel + fused inputs + sync-free forward — captured in one HIP graph."""
import math

import pytest
import torch

from ggrt_official_amd import splatting as sp
from tests.test_callsite_fused import _inputs

pytestmark = pytest.mark.gpu
dev = "cuda:0"


@pytest.mark.parametrize("scale_invariant", [True, False])
def test_camera_setup_matches_the_call_site_formulas(scale_invariant):
    from ggrt_official_amd.rasterizer import camera_setup
    b, v = 2, 2
    ext, intr, near, far, means, cov, harm, op = _inputs(b, v, 10, 4, seed=5)
    intr = intr.clone()
    intr[:, 0, 2] = torch.tensor([0.52, 0.4, 0.61, 0.5], device=dev)   # per-view principal points (fov uses them)
    view, full, campos, tanfov, scale = camera_setup(ext, intr, near, far, scale_invariant)
    calls = sp.boundary_arguments(ext, intr, near, far, (32, 48), torch.zeros(b * v, 3, device=dev), sp_rep(means, v),
                                  sp_rep(cov, v), sp_rep(harm, v), sp_rep(op, v), scale_invariant=scale_invariant)
    for i, (rs, kw) in enumerate(calls):
        assert torch.allclose(view[i], rs.viewmatrix, atol=2e-6)
        assert torch.allclose(full[i], rs.projmatrix, rtol=1e-5, atol=1e-5)
        assert torch.allclose(campos[i], rs.campos, atol=0 if scale_invariant else 0, rtol=0)
        assert math.isclose(float(tanfov[i, 0]), rs.tanfovx, rel_tol=2e-6)
        assert math.isclose(float(tanfov[i, 1]), rs.tanfovy, rel_tol=2e-6)
        want = 1.0 / float(near[i]) if scale_invariant else 1.0
        assert math.isclose(float(scale[i]), want, rel_tol=1e-7)


def sp_rep(x, v):
    return x[:, None].expand(-1, v, *x.shape[1:]).reshape(-1, *x.shape[1:])


def test_device_camera_renders_like_the_torch_camera():
    b, v, gc, h, w = 1, 2, 4000, 64, 96
    ext, intr, near, far, means, cov, harm, op = _inputs(b, v, gc, 9, seed=6)
    gs = sp.Gaussians(means=means, covariances=cov, harmonics=harm, opacities=op)
    bg = torch.zeros(b * v, 3, device=dev)
    a = sp.render_views_fused(ext, intr, near, far, (h, w), bg, gs, [0, 0], "depth", device_camera=True)
    c = sp.render_views_fused(ext, intr, near, far, (h, w), bg, gs, [0, 0], "depth", device_camera=False)
    # the two cameras differ in the last fp32 bit of a few matrix entries: a handful of threshold pixels
    # (alpha at 1/255, a radius at an integer) may flip, everything else agrees to rounding
    from tests.helpers import psnr
    for x, y in ((a[0], c[0]), (a[1], c[1])):
        assert psnr(x.detach().cpu().numpy(), y.detach().cpu().numpy()) > 75.0
        assert float(((x - y).abs() > 1e-4).float().mean()) < 1e-3


def test_whole_decoder_call_replays_from_a_hip_graph():
    """Camera kernel + fused inputs + sync-free forward: the decoder's forward AND backward run without a
    single host sync, so they can be captured once and replayed on new Gaussians / new poses."""
    from ggrt_official_amd import last_forward_status
    b, v, gc, h, w = 1, 1, 6000, 80, 112
    ext, intr, near, far, means, cov, harm, op = _inputs(b, v, gc, 25, seed=7)
    E, I, N, F = ext.reshape(b, v, 4, 4).clone(), intr.reshape(b, v, 3, 3), near.reshape(b, v), far.reshape(b, v)
    m, c, hm, o = [x.clone().requires_grad_() for x in (means, cov, harm, op)]
    dC = torch.randn(b, v, 3, h, w, device=dev)
    dD = torch.randn(b, v, h, w, device=dev) * 0.1
    dec = sp.DecoderSplattingCUDA(list_capacity=300_000).to(dev)

    def run():
        for t in (m, c, hm, o):
            t.grad = None
        out = dec(sp.Gaussians(means=m, covariances=c, harmonics=hm, opacities=o), E, I, N, F, (h, w), depth_mode="depth")
        torch.autograd.backward([out.color, out.depth], [dC, dD])
        return out

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_out = run()
    g_grads = [m.grad, c.grad, hm.grad, o.grad]
    with torch.no_grad():                      # new pose and new Gaussians, same storage
        E[0, 0, :3, 3] += torch.tensor([0.1, -0.05, 0.3], device=dev)
        m.add_(0.02)
        o.mul_(0.9)
    graph.replay()
    torch.cuda.synchronize()
    got = [g_out.color.clone(), g_out.depth.clone()] + [g.clone() for g in g_grads]
    assert not last_forward_status()[1]

    m2, c2, h2, o2 = [x.detach().clone().requires_grad_() for x in (m, c, hm, o)]
    ref = sp.DecoderSplattingCUDA().to(dev)(sp.Gaussians(means=m2, covariances=c2, harmonics=h2, opacities=o2), E, I, N, F,
                                            (h, w), depth_mode="depth")
    torch.autograd.backward([ref.color, ref.depth], [dC, dD])
    from tests.helpers import rel_l2
    assert torch.equal(got[0], ref.color.detach()) and torch.equal(got[1], ref.depth.detach())
    for x, y in zip(got[2:], [m2.grad, c2.grad, h2.grad, o2.grad]):
        assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 1e-5
