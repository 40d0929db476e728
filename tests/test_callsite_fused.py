"""`render_views_fused` / the decoder's default path — no torch operation on a Gaussian-sized tensor at the call
site — against the reference-shaped call site (`render_color_and_depth`, itself pinned by the
golden vectors): same images, same gradients w.r.t. the call site's own inputs (means [b,g,3], covariances
[b,g,3,3], harmonics [b,g,3,d_sh], opacities [b,g])."""
import math

import pytest
import torch

from ggrt_official_amd import splatting as sp

pytestmark = pytest.mark.gpu
dev = "cuda:0"


def _inputs(b, v, g_count, d_sh, seed):
    g = torch.Generator().manual_seed(seed)
    n = b * v
    ext = torch.eye(4).repeat(n, 1, 1)
    for i in range(n):
        a = 0.15 * (i - 1)
        ext[i, :3, :3] = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        ext[i, :3, 3] = torch.tensor([0.2 * i, -0.1 * i, 0.05 * i])
    intr = torch.tensor([[0.8, 0, 0.52], [0, 1.1, 0.47], [0, 0, 1]]).repeat(n, 1, 1)
    near = torch.tensor([1.0, 2.5, 0.7, 4.0][:n])
    far = torch.tensor([100.0, 250.0, 70.0, 400.0][:n])
    cam = torch.cat([(torch.rand(b, g_count, 2, generator=g) - 0.5) * 1.6, torch.ones(b, g_count, 1)], -1)
    depth = (4.0 + 30.0 * torch.rand(b, g_count, 1, generator=g))
    means = cam * depth
    q = torch.nn.functional.normalize(torch.randn(b, g_count, 4, generator=g), dim=-1)
    s = (0.01 + 0.04 * torch.rand(b, g_count, 3, generator=g)) * depth
    cov = sp.adapter_covariances(s, q, torch.eye(3).expand(b, 1, 3, 3))
    harm = torch.randn(b, g_count, 3, d_sh, generator=g) * 0.3
    op = 0.1 + 0.8 * torch.rand(b, g_count, generator=g)
    t = lambda x: x.to(dev)
    return t(ext), t(intr), t(near), t(far), t(means), t(cov), t(harm), t(op)


@pytest.mark.parametrize("d_sh,depth_mode", [(9, "depth"), (25, "depth"), (16, "disparity"), (4, None)])
def test_fused_call_site_equals_reference_shaped_call_site(d_sh, depth_mode):
    from tests.helpers import psnr, rel_l2
    b, v, gc, h, w = 2, 2, 5000, 80, 112
    ext, intr, near, far, means, cov, harm, op = _inputs(b, v, gc, d_sh, seed=d_sh)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev).expand(b * v, 3)
    gen = torch.Generator().manual_seed(1)
    dC = torch.randn(b * v, 3, h, w, generator=gen).to(dev)
    dD = torch.randn(b * v, h, w, generator=gen).to(dev) * 0.1

    def leaves():
        return [x.clone().requires_grad_() for x in (means, cov, harm, op)]

    rep = lambda x: x[:, None].expand(-1, v, *x.shape[1:]).reshape(-1, *x.shape[1:])
    # reference-shaped: Gaussians repeated per view, torch renorm / transpose / gather
    m, c, hm, o = leaves()
    if depth_mode is None:
        col_a = sp.render_cuda(ext, intr, near, far, (h, w), bg, rep(m), rep(c), rep(hm), rep(o))
        dep_a = None
        torch.autograd.backward([col_a], [dC])
    else:
        col_a, dep_a = sp.render_color_and_depth(ext, intr, near, far, (h, w), bg, rep(m), rep(c), rep(hm), rep(o),
                                                 depth_mode)
        torch.autograd.backward([col_a, dep_a], [dC, dD])
    ga = [x.grad for x in (m, c, hm, o)]
    # fused: nothing repeated, nothing pre-processed
    m, c, hm, o = leaves()
    gs = sp.Gaussians(means=m, covariances=c, harmonics=hm, opacities=o)
    col_b, dep_b = sp.render_views_fused(ext, intr, near, far, (h, w), bg, gs, [i // v for i in range(b * v)], depth_mode)
    if depth_mode is None:
        assert dep_b is None
        torch.autograd.backward([col_b], [dC])
    else:
        torch.autograd.backward([col_b, dep_b], [dC, dD])
    gb = [x.grad for x in (m, c, hm, o)]

    assert psnr(col_a.detach().cpu().numpy(), col_b.detach().cpu().numpy()) > 90.0
    assert float(col_a.abs().mean()) > 0.05
    if depth_mode is not None:
        assert psnr(dep_a.detach().cpu().numpy(), dep_b.detach().cpu().numpy()) > 90.0
    for x, y, name in zip(ga, gb, ("means", "covariances", "harmonics", "opacities")):
        assert x.shape == y.shape and float(x.abs().max()) > 0
        assert rel_l2(y.cpu().numpy(), x.cpu().numpy()) < 2e-4, name
    # the covariance gradient has the reference's structure: nothing below the diagonal
    assert float(gb[1][..., 1, 0].abs().max()) == 0 and float(gb[1][..., 2, 0].abs().max()) == 0


def test_decoder_default_path_is_the_fused_one_and_matches():
    b, v, gc, h, w = 1, 3, 4000, 64, 96
    ext, intr, near, far, means, cov, harm, op = _inputs(b, v, gc, 25, seed=3)
    gs = sp.Gaussians(means=means, covariances=cov, harmonics=harm, opacities=op)
    E, I = ext.reshape(b, v, 4, 4), intr.reshape(b, v, 3, 3)
    N, F = near.reshape(b, v), far.reshape(b, v)
    fused = sp.DecoderSplattingCUDA().to(dev)(gs, E, I, N, F, (h, w), depth_mode="depth")
    plain = sp.DecoderSplattingCUDA(fused_inputs=False).to(dev)(gs, E, I, N, F, (h, w), depth_mode="depth")
    twice = sp.DecoderSplattingCUDA(fused_depth=False, fused_inputs=False).to(dev)(gs, E, I, N, F, (h, w), depth_mode="depth")
    for a, c in ((fused.color, plain.color), (fused.depth, plain.depth), (fused.color, twice.color), (fused.depth, twice.depth)):
        assert torch.allclose(a, c, atol=2e-5, rtol=1e-5)
    assert fused.color.shape == (b, v, 3, h, w) and fused.depth.shape == (b, v, h, w)
