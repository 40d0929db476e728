"""Non-finite inputs (include/ggr_raster.h "Non-finite inputs"): a handful of Gaussians with NaN / ±Inf in
their means, covariances, opacities or SH coefficients leave the frame — radius 0, zero gradient — and everything else
renders as the C oracle (which states the same rule, oracle/ggr_oracle.c) renders it; in single calls, with the per-Gaussian
stage split or not, with either form of the depth sort, and in launch sets.  The encoder that feeds this path can emit such
values early in training (reference encoder_epipolar.py:189-195 → gaussian_adapter.py:62-81)."""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from oracle import c_oracle
from tests.helpers import check_grads, check_image, hip_forward_backward, oracle_forward

pytestmark = pytest.mark.gpu
NAN, INF = float("nan"), float("inf")


def _poisoned(P=20000, W=320, H=240, deg=3, M=None, seed=11):
    sc = make_scene(P, W, H, sh_degree=deg, profile="A", seed=seed, sh_stride=M)
    geom, colour = [], []
    def hit(i, kind):
        (geom if kind == "g" else colour).append(i)
    sc.means3D[10, 0] = NAN; hit(10, "g")
    sc.means3D[11, 2] = INF; hit(11, "g")
    sc.means3D[12, 1] = -INF; hit(12, "g")
    sc.means3D[13] = NAN; hit(13, "g")
    sc.cov3D[20, 0] = NAN; hit(20, "g")
    sc.cov3D[21, 3] = INF; hit(21, "g")
    sc.cov3D[22, 5] = -INF; hit(22, "g")
    sc.cov3D[23] = 1e30; hit(23, "g")               # finite, but a radius beyond 2^30 px
    sc.opacities[30, 0] = NAN; hit(30, "g")
    sc.opacities[31, 0] = INF; hit(31, "g")
    sc.shs[40, 0, 1] = NAN; hit(40, "c")
    sc.shs[41, 2, 0] = INF; hit(41, "c")
    sc.shs[42, 5, 2] = -INF; hit(42, "c")
    untouched = []
    if sc.shs.shape[1] > (deg + 1) ** 2 or M:     # a coefficient of a band that is NOT evaluated is never read
        sc.shs[50, sc.shs.shape[1] - 1, 0] = NAN
        untouched.append(50)
    return sc, geom, colour, untouched


def _check(sc, geom, colour, untouched, dL, compare_lists=False, **kw):
    st = oracle_forward(sc)
    ref = c_oracle.backward(st, dL.numpy())
    color, radii, depth, grads = hip_forward_backward(sc, dL, **kw)
    bad = geom + colour
    assert np.isfinite(color).all() and np.isfinite(depth).all()
    assert np.array_equal(radii, st.radii) and (radii[bad] == 0).all()
    vis = st.radii > 0
    assert vis[untouched].all() if untouched else True
    check_image(color, st.color, tag="nonfinite")
    keys = ["means3D", "means2D", "shs", "opacities", "cov3D_precomp"]
    for k in keys:
        g = grads[k].reshape(len(radii), -1)
        assert np.isfinite(g).all(), k
        assert (g[bad] == 0).all(), k
    check_grads(grads, ref, keys, tag="nonfinite")
    return st


@pytest.mark.parametrize("split", ["1", "0"])
@pytest.mark.parametrize("mode", ["global", "per_tile"])
def test_poisoned_gaussians_leave_the_frame(monkeypatch, split, mode):
    monkeypatch.setenv("GGR_SPLIT_COLOUR", split)
    monkeypatch.setenv("GGR_DEPTH_SORT", mode)
    sc, geom, colour, untouched = _poisoned()
    dL = upstream_gradient(sc.width, sc.height, seed=5)
    _check(sc, geom, colour, untouched, dL)


def test_unevaluated_bands_are_never_read():
    """GGRt's shape: 25 coefficients, cap 3 → coefficient 24 may hold anything; with cap 4 it is evaluated and the Gaussian leaves."""
    sc, geom, colour, untouched = _poisoned(deg=4, M=25)
    dL = upstream_gradient(sc.width, sc.height, seed=6)
    assert untouched == [50]
    _check(sc, geom, colour, untouched, dL, sh_max_degree=3)
    st4 = oracle_forward(sc, sh_cap=4)
    color, radii, depth, grads = hip_forward_backward(sc, dL, sh_max_degree=4)
    assert radii[50] == 0 and st4.radii[50] == 0 and np.array_equal(radii, st4.radii)
    check_image(color, st4.color, tag="nonfinite:cap4")


def test_poisoned_gaussians_in_a_launch_set():
    from ggrt_official_amd.rasterizer import rasterize_views
    from ggrt_official_amd.synthetic import camera_matrices
    dev = torch.device("cuda:0")
    sc, geom, colour, untouched = _poisoned(P=12000, W=256, H=192)
    V = 3
    views, projs, cams = [], [], []
    for v in range(V):
        c2w = torch.eye(4, dtype=torch.float64)
        c2w[0, 3] = 0.04 * v
        view, full, campos, *_ = camera_matrices(sc.width, sc.height, c2w=c2w)
        views.append(view); projs.append(full); cams.append(campos)
    s = sc.to(dev)
    leaves = {k: getattr(s, k).clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "cov3D")}
    dL = torch.stack([upstream_gradient(sc.width, sc.height, seed=7 + v) for v in range(V)]).to(dev)
    color, radii, depth = rasterize_views(leaves["means3D"], leaves["opacities"], torch.stack(views).to(dev),
                                          torch.stack(projs).to(dev), torch.stack(cams).to(dev), torch.zeros(V, 3, device=dev),
                                          torch.tensor([[sc.tanfovx, sc.tanfovy]] * V, device=dev), s.settings(),
                                          shs=leaves["shs"], cov3D_precomp=leaves["cov3D"])
    (color * dL).sum().backward()
    bad = geom + colour
    assert torch.isfinite(color).all() and (radii[:, bad] == 0).all()
    for k, t in leaves.items():
        g = t.grad.reshape(t.shape[0], -1)
        assert torch.isfinite(g).all() and (g[bad] == 0).all(), k
    # per view: what a single call renders
    from ggrt_official_amd import GaussianRasterizer
    for v in range(V):
        rs = s.settings()._replace(viewmatrix=views[v].to(dev), projmatrix=projs[v].to(dev), campos=cams[v].to(dev))
        c1, r1, _ = GaussianRasterizer(rs)(means3D=s.means3D, means2D=torch.zeros_like(s.means3D), opacities=s.opacities,
                                          shs=s.shs, cov3D_precomp=s.cov3D)
        assert torch.equal(c1, color[v]) and torch.equal(r1, radii[v])
