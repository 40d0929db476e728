"""Segmented blend backward (`-m gpu`): images below 4096 tiles carry per-pixel checkpoints from the forward and
replay each tile's list as up to 16 independent depth segments (ggr_common.h `ggr_ckpt_slots`, blend_bwd.hip).
Every small-image parity test runs through that path; these cases make the lists long on purpose — many
active segments per tile, checkpoint strides of one and of several staging batches, pixels that stop inside a
segment — and check all gradients against the oracles (colour: C oracle; colour + depth: fp64 autograd of the
PyTorch oracle)."""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from oracle import c_oracle
from oracle import torch_raster as tr
from tests.helpers import hip_forward_backward, oracle_forward, rel_l2
from tests.test_gpu_parity import check_grads, check_image

pytestmark = pytest.mark.gpu

BATCH, SLOTS = 256, 16


def _long_list_scene(P, W, H, D, seed, opacity_scale):
    sc = make_scene(P, W, H, sh_degree=D, profile="B", seed=seed)
    sc.opacities.mul_(opacity_scale)      # faint splats: pixels stay unsaturated deep into the lists
    sc.bg = torch.tensor([0.2, 0.5, 0.1])
    return sc


@pytest.mark.parametrize("P,W,H,opacity_scale,min_stride", [
    (30000, 80, 64, 0.5, BATCH),          # lists ≈ 3–4 k entries: stride 256, > 8 segments per tile
    (90000, 64, 48, 0.15, 2 * BATCH),     # lists > 4096 entries: stride ≥ 512
    (50000, 96, 48, 3.0, BATCH),          # opaque: every pixel stops inside the first segments
])
def test_long_lists_colour_gradients(P, W, H, opacity_scale, min_stride):
    sc = _long_list_scene(P, W, H, 1, seed=P % 97, opacity_scale=opacity_scale)
    dL = upstream_gradient(W, H, seed=5)
    st = oracle_forward(sc)
    lens = (st.ranges[:, 1].astype(np.int64) - st.ranges[:, 0]).max()
    stride = BATCH * max(1, -(-int(lens) // (SLOTS * BATCH)))
    assert stride >= min_stride, (lens, stride)
    if opacity_scale < 1:
        assert st.n_contrib.max() > 4 * stride, "the case is meant to keep several segments busy"
    ref = c_oracle.backward(st, dL.numpy())
    color, radii, depth, grads = hip_forward_backward(sc, dL)
    assert np.array_equal(radii, st.radii)
    check_image(color, st.color)
    check_grads(grads, ref, ["means3D", "means2D", "shs", "opacities", "cov3D_precomp"])


def test_long_lists_depth_gradient():
    W, H = 48, 32
    sc = _long_list_scene(9000, W, H, 2, seed=3, opacity_scale=0.3)
    dL = upstream_gradient(W, H, seed=1)
    dLd = upstream_gradient(W, H, seed=51)[0] * 0.3
    leaf = lambda t: t.double().clone().requires_grad_(True)
    m, op, sh, cov = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.shs), leaf(sc.cov3D)
    color, radii, depth = tr.rasterize(m, op, sc.viewmatrix.double(), sc.projmatrix.double(), sc.campos.double(), sc.bg,
                                       W, H, sc.tanfovx, sc.tanfovy, sc.sh_degree, shs=sh, cov3D_precomp=cov,
                                       depth_grad=True)
    ((color * dL.double()).sum() + (depth * dLd.double()).sum()).backward()
    ref = dict(means3D=m.grad, opacities=op.grad, shs=sh.grad, cov3D_precomp=cov.grad)
    hc, hr, hd, grads = hip_forward_backward(sc, dL, dL_ddepth=dLd)
    assert np.abs(hd - depth.detach().numpy()).max() < 1e-3 * max(1.0, float(depth.detach().abs().max()))
    for k, v in ref.items():
        assert rel_l2(grads[k], v.numpy()) < 1e-3, (k, rel_l2(grads[k], v.numpy()))


def test_inference_forward_skips_the_checkpoints():
    """torch.no_grad() forward (GGRt's eval loop, eval/eval_ggrt.py:317): no checkpoint area is allocated or written
    (GgrForwardOut.no_backward), the image is the same bit for bit."""
    from ggrt_official_amd import GaussianRasterizer, _lib
    sc = make_scene(30000, 160, 112, sh_degree=2, profile="B", seed=5).to("cuda:0")
    lib = _lib.load()
    assert lib.ggr_image_bytes_inference(160, 112, 1) < lib.ggr_image_bytes(160, 112) // 10
    args = dict(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs,
                cov3D_precomp=sc.cov3D)
    with torch.no_grad():
        a = GaussianRasterizer(sc.settings())(**args)
    m = sc.means3D.clone().requires_grad_(True)
    b = GaussianRasterizer(sc.settings())(**{**args, "means3D": m})
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    b[0].sum().backward()
    assert torch.isfinite(m.grad).all()


def test_no_grad_with_a_grad_requiring_sink_is_still_inference():
    """the call site's `means2D` sink always requires grad, and `ctx.needs_input_grad` mirrors
    `requires_grad` even under `torch.no_grad()` — the inference path must key on the caller's grad mode.  Seen
    through the allocator: the checkpoint area (320 B per pixel) must not be allocated."""
    from ggrt_official_amd import GaussianRasterizer, _lib
    W, H = 480, 352
    sc = make_scene(20000, W, H, sh_degree=1, profile="B", seed=6).to("cuda:0")
    lib = _lib.load()
    full, small = lib.ggr_image_bytes(W, H), lib.ggr_image_bytes_inference(W, H, 1)
    assert full > 40e6 > 10 * small
    sink = torch.zeros_like(sc.means3D).requires_grad_()
    args = dict(means3D=sc.means3D, means2D=sink, opacities=sc.opacities, shs=sc.shs, cov3D_precomp=sc.cov3D)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        out = GaussianRasterizer(sc.settings())(**args)
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() - base < full // 2
    assert not out[0].requires_grad
