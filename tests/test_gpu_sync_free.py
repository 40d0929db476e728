"""Sync-free forward (`list_capacity > 0`, C ABI `GgrForwardOut.binning_capacity`): same results as the exact
mode, an overflow flag instead of out-of-bounds accesses when the buffer is too small, and forward + backward
captured in a HIP graph replay correctly on new inputs."""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


def _run(s, settings, dL):
    from ggrt_official_amd.rasterizer import GaussianRasterizer
    leaves = [t.clone().requires_grad_() for t in (s.means3D, s.shs, s.opacities, s.cov3D)]
    m, sh, op, cov = leaves
    color, radii, depth = GaussianRasterizer(settings)(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sh,
                                                       cov3D_precomp=cov)
    color.backward(dL)
    return color.detach(), radii, depth.detach(), [t.grad for t in leaves]


def test_capacity_mode_matches_exact_mode_and_reports_the_count():
    from ggrt_official_amd.rasterizer import debug_forward_state, last_forward_status
    sc = make_scene(12_000, 200, 150, sh_degree=2, seed=8)
    s = sc.to("cuda:0")
    dL = upstream_gradient(sc.width, sc.height, device="cuda:0")
    N = debug_forward_state(s.means3D, s.opacities, s.settings(), shs=s.shs, cov3D_precomp=s.cov3D)["num_rendered"]
    ca, ra, da, ga = _run(s, s.settings(), dL)
    for cap in (N, N + 1000, 4 * N):
        cb, rb, db, gb = _run(s, s.settings()._replace(list_capacity=cap), dL)
        assert last_forward_status() == (N, False)
        assert torch.equal(ca, cb) and torch.equal(ra, rb) and torch.equal(da, db)
        for x, y in zip(ga, gb):
            assert rel_l2(y.cpu().numpy(), x.cpu().numpy()) < 1e-5


def test_too_small_a_buffer_raises_the_flag_and_stays_in_bounds():
    from ggrt_official_amd.rasterizer import debug_forward_state, last_forward_status
    sc = make_scene(12_000, 200, 150, sh_degree=0, seed=9)
    s = sc.to("cuda:0")
    dL = upstream_gradient(sc.width, sc.height, device="cuda:0")
    N = debug_forward_state(s.means3D, s.opacities, s.settings(), shs=s.shs, cov3D_precomp=s.cov3D)["num_rendered"]
    guard = torch.full((1 << 20,), 7, dtype=torch.uint8, device="cuda:0")   # likely neighbours of the small buffer
    color, radii, depth, grads = _run(s, s.settings()._replace(list_capacity=N // 3), dL)
    n, overflow = last_forward_status()
    assert n == N and overflow
    assert torch.isfinite(color).all() and all(torch.isfinite(g).all() for g in grads)
    assert int((guard != 7).sum()) == 0
    # the tiles whose lists fit before the cut are rendered exactly
    ref = _run(s, s.settings(), dL)[0]
    same_rows = (color == ref).all(dim=0).all(dim=1)
    assert bool(same_rows[:16].all())            # tile row 0 comes first in the list buffer


def test_forward_backward_replay_from_a_hip_graph():
    from ggrt_official_amd.rasterizer import GaussianRasterizer, last_forward_status
    sc = make_scene(8_000, 160, 128, sh_degree=1, seed=10)
    s = sc.to("cuda:0")
    dL = upstream_gradient(sc.width, sc.height, device="cuda:0")
    rs = s.settings()._replace(list_capacity=400_000)
    means = s.means3D.clone().requires_grad_()
    shs = s.shs.clone().requires_grad_()
    op = s.opacities.clone().requires_grad_()
    cov = s.cov3D.clone().requires_grad_()
    m2d = torch.zeros_like(means, requires_grad=True)
    rast = GaussianRasterizer(rs)

    def fwd_bwd():
        for t in (means, shs, op, cov, m2d):
            t.grad = None
        color, radii, depth = rast(means3D=means, means2D=m2d, opacities=op, shs=shs, cov3D_precomp=cov)
        color.backward(dL)
        return color, radii

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_color, g_radii = fwd_bwd()
    g_grads = [means.grad, shs.grad, op.grad, cov.grad]

    # new inputs, same storage → replay
    with torch.no_grad():
        means.add_(torch.tensor([0.05, -0.03, 0.4], device="cuda:0"))
        op.mul_(0.8)
    graph.replay()
    torch.cuda.synchronize()
    got = [g_color.clone(), g_radii.clone()] + [g.clone() for g in g_grads]
    assert not last_forward_status()[1]

    eager_rs = s.settings()
    e_means, e_shs, e_op, e_cov = [t.detach().clone().requires_grad_() for t in (means, shs, op, cov)]
    color, radii, _ = GaussianRasterizer(eager_rs)(means3D=e_means, means2D=torch.zeros_like(e_means), opacities=e_op,
                                                   shs=e_shs, cov3D_precomp=e_cov)
    color.backward(dL)
    assert torch.equal(got[0], color.detach()) and torch.equal(got[1], radii)
    for a, b in zip(got[2:], [e_means.grad, e_shs.grad, e_op.grad, e_cov.grad]):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_sets_with_scissor_replay_from_a_hip_graph():
    """Round 3's launch-set extensions under stream capture: two Gaussian SETS, a scissor, sync-free lists — captured
    once, replayed on new inputs, equal to the eager exact-mode result."""
    from ggrt_official_amd import rasterize_views
    from ggrt_official_amd.rasterizer import last_forward_status
    dev = "cuda:0"
    W, H, P = 160, 128, 6000
    scs = [make_scene(P, W, H, sh_degree=2, seed=60 + b).to(dev) for b in range(2)]
    stack = lambda f: torch.stack([f(s) for s in scs])
    view, proj = stack(lambda s: s.viewmatrix), stack(lambda s: s.projmatrix)
    cam, bg = stack(lambda s: s.campos), stack(lambda s: s.bg)
    tf = torch.tensor([[s.tanfovx, s.tanfovy] for s in scs], dtype=torch.float32, device=dev)
    dL = torch.stack([upstream_gradient(W, H, seed=70 + b, device=dev) for b in range(2)])
    win = (40, 20, 130, 100)
    leaves = [stack(f).clone().requires_grad_() for f in (lambda s: s.means3D, lambda s: s.shs, lambda s: s.opacities,
                                                           lambda s: s.cov3D)]
    m, sh, op, cov = leaves

    def step(rs):
        for t in leaves:
            t.grad = None
        color, radii, _ = rasterize_views(m, op, view, proj, cam, bg, tf, rs, shs=sh, cov3D_precomp=cov)
        color.backward(dL)
        return color

    base = scs[0].settings()._replace(scissor=win)
    eager = step(base).detach().clone()
    eager_grads = [t.grad.clone() for t in leaves]
    rs = base._replace(list_capacity=400_000)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(rs)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    keep = {}
    with torch.cuda.graph(graph):
        keep["color"] = step(rs)
    graph.replay()
    torch.cuda.synchronize()
    n, overflow = last_forward_status()
    assert n > 0 and not overflow
    assert torch.equal(keep["color"].detach(), eager)
    for t, g in zip(leaves, eager_grads):
        assert rel_l2(t.grad.cpu().numpy(), g.cpu().numpy()) < 1e-5
    # new inputs in the same storage: the replay follows them
    with torch.no_grad():
        m.add_(0.01)
    graph.replay()
    torch.cuda.synchronize()
    moved = keep["color"].detach().clone()
    assert not torch.equal(moved, eager)
    assert torch.equal(step(base).detach(), moved)
