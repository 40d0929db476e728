"""Re-entrancy ("the library must be re-entrant and hold no global mutable state"): independent frames in
flight on different HIP streams — from one host thread and from two — give what the same calls give one after the other
(images bit for bit, gradients to the summation order of the float atomics)."""
import threading

import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
dev = "cuda:0"


class _Frame:
    def __init__(self, seed, P, W, H):
        sc = make_scene(P, W, H, sh_degree=3, profile="A", seed=seed).to(dev)
        self.sc = sc
        self.dL = upstream_gradient(W, H, seed=100 + seed, device=dev)
        self.leaves = [t.clone().requires_grad_() for t in (sc.means3D, sc.shs, sc.opacities, sc.cov3D)]

    def step(self):
        from ggrt_official_amd import GaussianRasterizer
        for t in self.leaves:
            t.grad = None
        m, sh, op, cov = self.leaves
        color, radii, _ = GaussianRasterizer(self.sc.settings())(means3D=m, means2D=torch.zeros_like(m), opacities=op,
                                                                 shs=sh, cov3D_precomp=cov)
        color.backward(self.dL)
        return color.detach().clone(), radii.clone(), [t.grad.clone() for t in self.leaves]


def _same(a, b):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 2e-5


def test_two_frames_in_flight_on_two_streams():
    frames = [_Frame(1, 120_000, 640, 360), _Frame(2, 90_000, 500, 410)]   # different sizes: different buffer shapes
    want = [f.step() for f in frames]
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for s in lanes:
        s.wait_stream(torch.cuda.current_stream())
    got = [None, None]
    for i in range(12):     # alternate, nothing waits in between: the frames overlap on the device
        with torch.cuda.stream(lanes[i & 1]):
            got[i & 1] = frames[i & 1].step()
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        _same(w, g)


def test_two_host_threads_each_with_its_own_stream():
    frames = [_Frame(3, 100_000, 512, 384), _Frame(4, 100_000, 512, 384)]
    want = [f.step() for f in frames]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def work(k):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.default_stream(torch.device(dev)))
            with torch.cuda.stream(s):
                for _ in range(8):
                    got[k] = frames[k].step()
            s.synchronize()
        except Exception as e:  # pragma: no cover - reported below
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for w, g in zip(want, got):
        _same(w, g)


@pytest.mark.parametrize("case", ["one_view_d3", "one_view_d4_m25", "scale_rot", "views3"])
def test_split_per_gaussian_stage_equals_one_kernel(case, monkeypatch):
    """Round 5: with SH colours the forward runs the per-Gaussian stage as a geometry kernel on the caller's stream and a
    colour kernel on a library-owned side stream (csrc/api.hip forward_impl); `GGR_SPLIT_COLOUR=0` runs it as one kernel.
    Same arithmetic in the same order: colour, radii, depth and every gradient are bit-identical (the gradients up to the
    blend's atomic order, as between any two runs)."""
    import numpy as np
    from ggrt_official_amd import GaussianRasterizer
    from ggrt_official_amd.rasterizer import rasterize_views
    from ggrt_official_amd.synthetic import make_scene, upstream_gradient
    dev = "cuda:0"
    deg = 4 if case == "one_view_d4_m25" else 3
    sc = make_scene(30000, 208, 160, sh_degree=deg, profile="B" if deg == 4 else "A", seed=41).to(dev)
    dL = upstream_gradient(sc.width, sc.height, seed=5, device=dev)

    def run():
        leaves = [t.clone().requires_grad_() for t in (sc.means3D, sc.shs, sc.opacities, sc.cov3D, sc.scales, sc.rotations)]
        m, sh, op, cov, scl, rot = leaves
        rs = sc.settings()._replace(sh_max_degree=4 if deg == 4 else 3)
        if case == "views3":
            views = []
            for k in range(3):
                c2w = torch.eye(4, dtype=torch.float64)
                c2w[0, 3] = 0.05 * k
                v = make_scene(8, sc.width, sc.height, sh_degree=deg, seed=1, c2w=c2w).to(dev)
                views.append((v.viewmatrix, v.projmatrix, v.campos))
            tanfov = torch.tensor([[sc.tanfovx, sc.tanfovy]] * 3, dtype=torch.float32, device=dev)
            color, radii, depth = rasterize_views(m, op, torch.stack([v[0] for v in views]), torch.stack([v[1] for v in views]),
                                                  torch.stack([v[2] for v in views]), sc.bg[None].expand(3, 3).contiguous(),
                                                  tanfov, rs, shs=sh, cov3D_precomp=cov)
            (color * dL[None]).sum().backward()
            used = (m, sh, op, cov)
        elif case == "scale_rot":
            color, radii, depth = GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sh,
                                                         scales=scl, rotations=rot)
            (color * dL).sum().backward()
            used = (m, sh, op, scl, rot)
        else:
            color, radii, depth = GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sh,
                                                         cov3D_precomp=cov)
            (color * dL).sum().backward()
            used = (m, sh, op, cov)
        torch.cuda.synchronize()
        return color.detach().cpu(), radii.cpu(), depth.detach().cpu(), [t.grad.cpu().numpy() for t in used]

    monkeypatch.setenv("GGR_SPLIT_COLOUR", "0")
    one = run()
    assert int((one[1] > 0).sum()) > 1000
    monkeypatch.setenv("GGR_SPLIT_COLOUR", "1")
    # the colour kernel as persistent blocks walking the chunks (default: 1 per CU; 3 per CU) and one block per chunk (0)
    for blocks in (None, "3", "0"):
        if blocks is None:
            monkeypatch.delenv("GGR_COLOUR_BLOCKS_PER_CU", raising=False)
        else:
            monkeypatch.setenv("GGR_COLOUR_BLOCKS_PER_CU", blocks)
        two = run()
        assert torch.equal(one[0], two[0]) and torch.equal(one[1], two[1]) and torch.equal(one[2], two[2]), blocks
        for a, b in zip(one[3], two[3]):
            assert np.abs(a).max() > 0
            assert np.linalg.norm(a - b) <= 2e-6 * np.linalg.norm(a), blocks
