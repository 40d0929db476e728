"""Extensions beyond the reference's gradient set, checked against fp64 autograd of the PyTorch oracle:

* gradient of the third output (out_depth = Σ z·α·T) w.r.t. every input (the "w-depth" fork family
  differentiates it; GGRt discards the output, so this is an extension), and
* camera gradients ∂L/∂viewmatrix, ∂L/∂projmatrix, ∂L/∂campos (SURVEY.md §8f-3 — named by
  BASELINE.json's north-star, absent from the reference whose extension receives the matrices inside a
  NamedTuple).
"""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from oracle import torch_raster as tr
from tests.helpers import hip_forward_backward, rel_l2

pytestmark = pytest.mark.gpu


def _pose(seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(3, generator=g) - 0.5) * 0.3
    K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    T = torch.eye(4)
    T[:3, :3] = torch.matrix_exp(K)
    T[:3, 3] = (torch.rand(3, generator=g) - 0.5) * 0.5
    return T


def _oracle_grads(sc, dL, dLd, pose):
    leaf = lambda t: t.double().clone().requires_grad_(True)
    m, op, sh, cov = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.shs), leaf(sc.cov3D)
    V, PM, cam = leaf(sc.viewmatrix), leaf(sc.projmatrix), leaf(sc.campos)
    color, radii, depth = tr.rasterize(m, op, V, PM, cam, sc.bg, sc.width, sc.height, sc.tanfovx, sc.tanfovy,
                                       sc.sh_degree, shs=sh, cov3D_precomp=cov, depth_grad=dLd is not None)
    loss = (color * dL.double()).sum()
    if dLd is not None:
        loss = loss + (depth * dLd.double()).sum()
    loss.backward()
    out = dict(means3D=m.grad, opacities=op.grad, shs=sh.grad, cov3D_precomp=cov.grad)
    if pose:
        out.update(viewmatrix=V.grad, projmatrix=PM.grad, campos=cam.grad)
    return {k: v.numpy() for k, v in out.items()}, color.detach().numpy(), depth.detach().numpy()


@pytest.mark.parametrize("seed", [0, 1])
def test_depth_output_gradient(seed):
    sc = make_scene(2500, 96, 64, sh_degree=2, profile="A", seed=seed, c2w=_pose(seed))
    dL = upstream_gradient(96, 64, seed=seed)
    dLd = upstream_gradient(96, 64, seed=seed + 50)[0] * 0.3
    ref, rc, rd = _oracle_grads(sc, dL, dLd, pose=False)
    color, radii, depth, grads = hip_forward_backward(sc, dL, dL_ddepth=dLd)
    assert np.abs(depth - rd).max() < 1e-3 * max(1.0, np.abs(rd).max())
    for k in ("means3D", "opacities", "shs", "cov3D_precomp"):
        assert rel_l2(grads[k], ref[k]) < 1e-3, (k, rel_l2(grads[k], ref[k]))


@pytest.mark.parametrize("seed,with_depth", [(0, False), (1, True), (2, False)])
def test_camera_gradients(seed, with_depth):
    sc = make_scene(3000, 112, 80, sh_degree=3, profile="A", seed=seed, c2w=_pose(seed + 10))
    dL = upstream_gradient(112, 80, seed=seed)
    dLd = upstream_gradient(112, 80, seed=seed + 50)[0] * 0.3 if with_depth else None
    ref, _, _ = _oracle_grads(sc, dL, dLd, pose=True)
    color, radii, depth, grads = hip_forward_backward(sc, dL, dL_ddepth=dLd, pose=True)
    for k in ("viewmatrix", "projmatrix", "campos", "means3D"):
        r = rel_l2(grads[k], ref[k])
        assert r < 2e-3, (k, r, grads[k], ref[k])


def test_camera_gradients_chain_to_extrinsics():
    """End-to-end: a loss on the rendered image differentiated w.r.t. a camera-to-world pose through
    viewmatrix = inv(c2w)^T, projmatrix = viewmatrix @ P^T, campos = c2w[:3,3] (all built in torch)."""
    from ggrt_official_amd import GaussianRasterizer
    from ggrt_official_amd.synthetic import camera_matrices
    dev = torch.device("cuda:0")
    base = _pose(3)
    sc = make_scene(2000, 80, 64, sh_degree=1, profile="A", seed=4, c2w=base)
    dL = upstream_gradient(80, 64, seed=9)

    def build(c2w, dtype):
        view = torch.linalg.inv(c2w).T
        _, full0, _, tfx, tfy, _, _ = camera_matrices(80, 64)
        Pm_T = full0.to(dtype).to(c2w.device)  # identity pose: full = I @ P^T = P^T
        return view, view @ Pm_T, c2w[:3, 3]

    # oracle (fp64 autograd)
    c2w64 = base.double().clone().requires_grad_(True)
    V, PM, cam = build(c2w64, torch.float64)
    color, _, _ = tr.rasterize(sc.means3D.double(), sc.opacities.double(), V, PM, cam, sc.bg, 80, 64, sc.tanfovx,
                               sc.tanfovy, 1, shs=sc.shs.double(), cov3D_precomp=sc.cov3D.double())
    (color * dL.double()).sum().backward()
    # HIP
    c2w = base.clone().to(dev).requires_grad_(True)
    V, PM, cam = build(c2w, torch.float32)
    s = sc.to(dev)
    rs = s.settings()._replace(viewmatrix=V, projmatrix=PM, campos=cam)
    color, _, _ = GaussianRasterizer(rs)(means3D=s.means3D, means2D=torch.zeros_like(s.means3D), opacities=s.opacities,
                                         shs=s.shs, cov3D_precomp=s.cov3D)
    (color * dL.to(dev)).sum().backward()
    assert rel_l2(c2w.grad.cpu().numpy()[:3], c2w64.grad.numpy()[:3]) < 2e-3
