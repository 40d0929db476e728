"""Views of one batch rendered on concurrent HIP streams must give exactly what the
serial loop of the reference call site gives — images bit-identical (forward is deterministic), gradients
within the atomics tolerance — with and without the fused depth pass."""
import time

import numpy as np
import pytest
import torch

from ggrt_official_amd import splatting
from ggrt_official_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def _batch(v, P, W, H, seed=0):
    """v views of the SAME Gaussians from slightly different cameras, in the call-site's layout."""
    dev = torch.device("cuda:0")
    sc = make_scene(P, W, H, sh_degree=4, profile="B", seed=seed)
    g = torch.Generator().manual_seed(seed)
    extr = torch.eye(4).repeat(v, 1, 1)
    extr[:, :3, 3] = (torch.rand(v, 3, generator=g) - 0.5) * 0.3
    fx = 0.5 / sc.tanfovx
    intr = torch.eye(3).repeat(v, 1, 1)
    intr[:, 0, 0], intr[:, 1, 1], intr[:, 0, 2], intr[:, 1, 2] = fx, fx * W / H, 0.5, 0.5
    near, far = torch.full((v,), 1.0), torch.full((v,), 100.0)
    # covariance matrices from the 6-vectors
    c = sc.cov3D
    cov = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], -1).reshape(-1, 3, 3)
    rep = lambda t: t[None].expand(v, *t.shape).contiguous().to(dev)
    return dict(extrinsics=extr.to(dev), intrinsics=intr.to(dev), near=near.to(dev), far=far.to(dev),
                image_shape=(H, W), background_color=torch.zeros(v, 3, device=dev), gaussian_means=rep(sc.means3D),
                gaussian_covariances=rep(cov), gaussian_sh_coefficients=rep(sc.shs.permute(0, 2, 1)),
                gaussian_opacities=rep(sc.opacities[:, 0]))


@pytest.mark.parametrize("fused", [False, True])
def test_concurrent_views_equal_serial(fused):
    b = _batch(4, 30000, 240, 176)
    g = torch.Generator().manual_seed(1)
    wc = torch.randn(4, 3, 176, 240, generator=g).cuda()
    wd = torch.randn(4, 176, 240, generator=g).cuda()
    res = []
    for conc in (False, True):
        kw = {k: (v.clone().requires_grad_(True) if k.startswith("gaussian") else v) for k, v in b.items()}
        if fused:
            color, depth = splatting.render_color_and_depth(**kw, concurrent_views=conc)
            loss = (color * wc).sum() + (depth * wd).sum()
        else:
            color = splatting.render_cuda(**kw, concurrent_views=conc)
            loss = (color * wc).sum()
        loss.backward()
        torch.cuda.synchronize()
        res.append((color.detach().cpu(), [kw[k].grad.cpu().numpy() for k in sorted(kw) if k.startswith("gaussian")]))
    assert torch.equal(res[0][0], res[1][0])
    for a, c in zip(res[0][1], res[1][1]):
        assert np.linalg.norm(a - c) <= 1e-3 * max(np.linalg.norm(a), 1e-30)


def test_concurrent_views_are_not_slower():
    """Not a strict perf gate (shared box), just evidence: 4 GGRt-sized views, forward only."""
    b = _batch(4, 400_000, 480, 352, seed=3)
    out = {}
    for conc in (False, True):
        for _ in range(2):
            splatting.render_cuda(**b, concurrent_views=conc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            splatting.render_cuda(**b, concurrent_views=conc)
        torch.cuda.synchronize()
        out[conc] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"4 views, 400k Gaussians, 480x352: serial {out[False]:.2f} ms, concurrent streams {out[True]:.2f} ms")
    assert out[True] < out[False] * 1.25
