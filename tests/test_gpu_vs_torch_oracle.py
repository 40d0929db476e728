"""The HIP path against the SECOND restatement directly (`-m gpu`; VERDICT r3 weak #1a / next #1b).

Every other GPU parity test compares the kernels with `oracle/ggr_oracle.c`, whose preprocess the HIP kernel follows
operation for operation (that is what makes radii / rects / lists bit-exact) and whose backward is upstream's hand-written
analytic one.  A misreading of the specification shared by the two would be green in both.  `oracle/torch_raster.py` is
the independent leg: vectorised PyTorch, gradients by AUTOGRAD of the forward (SURVEY Appendix A.5's straight-through
terms made explicit), no shared code or operation order with either C file.  Until round 4 it only met the C oracle on
CPU at a few thousand Gaussians (tests/test_oracle.py); here it meets the kernels at 25-40 k Gaussians.

Arithmetic of the torch leg: fp32, like the specification itself.  The rasterizer's discrete decisions (radius =
ceil(3·sqrt(λ)), the truncations of the tile-rect rule, α ≥ 1/255) are taken on fp32 values, and an fp64 evaluation
legitimately decides a few of them differently: measured on CPU, fp64 torch vs the C oracle lists ≈ 0.1 % of the
pixel-aligned Gaussians of a GGRt-like scene (profile B: means exactly on pixel centres) in a different tile set (4e-3
rel-L2 in the gradients), and at 40 k Gaussians of profile A one radius differs by 1 (found on the MI355X, round 4) —
while fp32 torch vs the C oracle has identical radii and lists and 4e-7 … 4e-5 rel-L2.  The independence that matters
here is of FORMULATION (vectorised forward + autograd vs hand-written analytic backward), not of word length.

Bars: images through tests/helpers.check_image; gradients rel-L2 ≤ 1e-3 over all rows (north-star) and ≤ 2e-5 once the
TWO rows with the largest error are set aside (an α-threshold flip between differently rounded evaluations adds or drops
one (pixel, Gaussian) term: measured C-oracle-vs-torch on CPU at these sizes, one such row = 4.5e-5 of the norm while the
rest agrees to 1e-5).
"""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from oracle import torch_raster as tr
from tests.helpers import GRAD_RTOL, GRAD_RTOL_ALL, check_image, hip_forward_backward, record_metric, rel_l2

pytestmark = pytest.mark.gpu

CASES = [
    # P, W, H, D, profile, sh cap, covariance input?, seed
    (40000, 256, 192, 3, "A", 3, True, 21),
    (30000, 208, 160, 4, "B", 3, True, 22),     # GGRt's form: sh_degree 4, 25 coefficients, bands 0..3
    (30000, 208, 160, 4, "B", 4, True, 23),     # … and with band 4 evaluated
    (25000, 192, 128, 1, "A", 3, False, 24),    # scales + rotations instead of covariances
]


def _grads_close(got, ref, key, tag):
    a = np.asarray(got, np.float64)
    b = np.asarray(ref, np.float64)
    rows = a.shape[0]
    a2, b2 = a.reshape(rows, -1), b.reshape(rows, -1)
    r_all = rel_l2(a2, b2)
    err = np.linalg.norm(a2 - b2, axis=1)
    keep = np.ones(rows, bool)
    keep[np.argpartition(-err, 1)[:2]] = False
    r = float(np.linalg.norm((a2 - b2)[keep]) / max(np.linalg.norm(b2[keep]), 1e-30))
    record_metric(f"{tag}:{key}", kind=1, rel_l2=r, rel_l2_all=r_all)
    assert r_all <= GRAD_RTOL_ALL, f"grad {key}: rel-L2 over all rows {r_all:.3e}"
    assert r <= GRAD_RTOL, f"grad {key}: rel-L2 {r:.3e} (2 rows set aside; all rows {r_all:.3e})"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("P,W,H,D,profile,cap,use_cov,seed", CASES)
def test_hip_matches_torch_autograd(P, W, H, D, profile, cap, use_cov, seed):
    sc = make_scene(P, W, H, sh_degree=D, profile=profile, seed=seed)
    dL = upstream_gradient(W, H, seed=seed)
    dt = torch.float32
    leaf = lambda t: t.to(dt).clone().requires_grad_(True)
    m, op, sh = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.shs)
    kw = dict(cov3D_precomp=leaf(sc.cov3D)) if use_cov else dict(scales=leaf(sc.scales), rotations=leaf(sc.rotations))
    color, radii, depth, state = tr.rasterize(m, op, sc.viewmatrix.to(dt), sc.projmatrix.to(dt), sc.campos.to(dt), sc.bg,
                                              W, H, sc.tanfovx, sc.tanfovy, D, shs=sh, return_state=True, sh_cap=cap, **kw)
    (color * dL.to(dt)).sum().backward()
    ref = dict(means3D=m.grad, opacities=op.grad, shs=sh.grad, **{k: v.grad for k, v in kw.items()})

    tag = f"torch:{profile}{D}cap{cap}{'' if use_cov else 'sr'}"
    h_color, h_radii, h_depth, grads = hip_forward_backward(sc, dL, use_cov=use_cov, sh_max_degree=cap)
    assert np.array_equal(h_radii, radii.numpy())
    check_image(h_color, color.detach().numpy(), tag=tag)
    check_image(h_depth, depth.detach().numpy(), name="depth", tag=tag + ":depth")
    for k, g in ref.items():
        _grads_close(grads[k], g.numpy(), k, tag)
    # the reference-rect build lists exactly what the torch restatement's stable 64-bit key sort lists
    from ggrt_official_amd.rasterizer import debug_forward_state
    s = sc.to("cuda:0")
    geo = dict(cov3D_precomp=s.cov3D) if use_cov else dict(scales=s.scales, rotations=s.rotations)
    out = debug_forward_state(s.means3D, s.opacities, s.settings()._replace(reference_rects=True, sh_max_degree=cap),
                              shs=s.shs, **geo)
    assert out["num_rendered"] == state["num_rendered"]
    assert np.array_equal(out["point_list"].cpu().numpy().astype(np.int64), state["point_list"].numpy().astype(np.int64))
    assert np.array_equal(out["ranges"].cpu().numpy(), state["ranges"].numpy())
