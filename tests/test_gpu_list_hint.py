"""The default mode's list-size guess (rasterizer._forward_with_guess, GgrForwardOut.capacity_is_hint): a forward whose
list buffer was sized from the previous call of the same shape returns what upstream's order returns — also when the
guess was too small and the call is repeated — and the same num_rendered."""
import pytest
import torch

import ggrt_official_amd
from ggrt_official_amd import rasterizer as R
from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
dev = "cuda:0"


def _run(sc, dL, scale_modifier=1.0):
    s = sc.to(dev)
    leaves = [t.clone().requires_grad_() for t in (s.means3D, s.shs, s.opacities, s.cov3D)]
    m, sh, op, cov = leaves
    rs = s.settings()._replace(scale_modifier=scale_modifier)
    color, radii, depth = ggrt_official_amd.GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m), opacities=op,
                                                                   shs=sh, cov3D_precomp=cov)
    n, _ = ggrt_official_amd.last_forward_status()
    color.backward(dL.to(dev))
    return color.detach().cpu(), radii.cpu(), depth.detach().cpu(), [t.grad.cpu() for t in leaves], n


def _same(a, b):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[4] == b[4]
    for x, y in zip(a[3], b[3]):
        assert rel_l2(x.numpy(), y.numpy()) < 2e-5


def test_guessed_buffer_gives_the_exact_modes_results():
    P, W, H = 60_000, 400, 304
    sc = make_scene(P, W, H, sh_degree=2, profile="A", seed=4)
    dL = upstream_gradient(W, H, seed=9)
    prev = ggrt_official_amd.set_list_hint(False)
    try:
        want = _run(sc, dL)                      # upstream's order
        ggrt_official_amd.set_list_hint(True)
        first = _run(sc, dL)                     # nothing known about this shape yet: upstream's order, notes N
        key = (0, P, W, H, 1, None)
        assert R._capacity_guess(key)[0] >= want[4]
        second = _run(sc, dL)                    # guessed buffer
        _same(want, first)
        _same(want, second)
    finally:
        ggrt_official_amd.set_list_hint(prev)


def test_a_guess_that_is_too_small_is_repaired_inside_the_call():
    P, W, H = 50_000, 320, 240
    small = make_scene(P, W, H, sh_degree=1, profile="B", seed=6)      # small splats: few list entries
    big = make_scene(P, W, H, sh_degree=1, profile="A", seed=7)        # same shape, several times the entries
    dL = upstream_gradient(W, H, seed=3)
    prev = ggrt_official_amd.set_list_hint(False)
    try:
        want_small, want_big = _run(small, dL), _run(big, dL)
        assert want_big[4] > 1.5 * want_small[4]
        ggrt_official_amd.set_list_hint(True)
        ggrt_official_amd.clear_list_hints()
        ggrt_official_amd.list_hint_stats(reset=True)
        _same(want_small, _run(small, dL))       # notes the small N
        assert ggrt_official_amd.list_hint_stats() == {"hinted": 0, "missed": 0, "exact": 1}
        got_big = _run(big, dL)                  # guess too small → repaired inside the call (exact buffer, scatter + blend again)
        _same(want_big, got_big)
        assert ggrt_official_amd.list_hint_stats() == {"hinted": 0, "missed": 1, "exact": 1}
        _same(want_big, _run(big, dL))           # now guessed from the big N
        _same(want_small, _run(small, dL))       # (an over-sized buffer is fine)
        assert ggrt_official_amd.list_hint_stats(reset=True) == {"hinted": 2, "missed": 1, "exact": 1}
        assert ggrt_official_amd.list_hint_stats() == {"hinted": 0, "missed": 0, "exact": 0}
    finally:
        ggrt_official_amd.set_list_hint(prev)


def test_streaming_copy_yardstick():
    """`ggr_debug_copy` (bench.py's HBM ceiling): copies, in both launch forms, and refuses misaligned arguments."""
    from ggrt_official_amd import _lib
    lib = _lib.load()
    a = torch.randn(1 << 20, device=dev)
    for blocks in (0, 512):
        b = torch.zeros_like(a)
        assert lib.ggr_debug_copy(a.data_ptr(), b.data_ptr(), a.numel() * 4, blocks, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(a, b)
    assert lib.ggr_debug_copy(a.data_ptr() + 4, b.data_ptr(), 1024, 0, None) == 1      # GGR_E_INVALID
    assert lib.ggr_debug_copy(a.data_ptr(), b.data_ptr(), 1000, 0, None) == 1


def test_a_missed_guess_is_repaired_in_a_launch_set_too():
    """The same repair on the launch-set entry (ggr_forward_views: three views of one Gaussian set): history holds the small
    scene's count, the large scene misses, and the call still returns what upstream's order returns."""
    from ggrt_official_amd.rasterizer import rasterize_views
    P, W, H, V = 40_000, 256, 192, 3
    small = make_scene(P, W, H, sh_degree=1, profile="B", seed=16).to(dev)
    big = make_scene(P, W, H, sh_degree=1, profile="A", seed=17).to(dev)
    dL = upstream_gradient(W, H, seed=5).to(dev)

    def cams(sc):
        vs = []
        for k in range(V):
            c2w = torch.eye(4, dtype=torch.float64)
            c2w[0, 3] = 0.04 * k
            v = make_scene(8, W, H, sh_degree=1, seed=1, c2w=c2w).to(dev)
            vs.append((v.viewmatrix, v.projmatrix, v.campos))
        tanfov = torch.tensor([[sc.tanfovx, sc.tanfovy]] * V, dtype=torch.float32, device=dev)
        return (torch.stack([v[0] for v in vs]), torch.stack([v[1] for v in vs]), torch.stack([v[2] for v in vs]),
                sc.bg[None].expand(V, 3).contiguous(), tanfov)

    def run(sc):
        leaves = [t.clone().requires_grad_() for t in (sc.means3D, sc.shs, sc.opacities, sc.cov3D)]
        m, sh, op, cov = leaves
        color, radii, depth = rasterize_views(m, op, *cams(sc), sc.settings(), shs=sh, cov3D_precomp=cov)
        (color * dL[None]).sum().backward()
        torch.cuda.synchronize()
        return color.detach().cpu(), radii.cpu(), depth.detach().cpu(), [t.grad.cpu() for t in leaves]

    def same(a, b):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for x, y in zip(a[3], b[3]):
            assert rel_l2(x.numpy(), y.numpy()) < 2e-5

    prev = ggrt_official_amd.set_list_hint(False)
    try:
        want_small, want_big = run(small), run(big)
        ggrt_official_amd.set_list_hint(True)
        ggrt_official_amd.clear_list_hints()
        ggrt_official_amd.list_hint_stats(reset=True)
        same(want_small, run(small))
        same(want_big, run(big))                 # the miss
        st = ggrt_official_amd.list_hint_stats(reset=True)
        assert st["missed"] == 1 and st["exact"] == 1, st
        same(want_big, run(big))
    finally:
        ggrt_official_amd.set_list_hint(prev)
