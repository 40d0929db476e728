"""The HIP path against the INDEPENDENT restatement at FULL size (`-m gpu`).

At BASELINE's full sizes the kernels had only ever met `oracle/ggr_oracle.c`, whose preprocess they follow operation for
operation; the independent leg (`oracle/torch_raster.py`: vectorised PyTorch, autograd backward, no shared code or
operation order) stopped at 40 k Gaussians (tests/test_gpu_vs_torch_oracle.py) because a whole 1080p frame takes it minutes.
Here it meets the kernels at C3 (1 M Gaussians, 1920×1080) and C5′ (1.01 M pixel-aligned Gaussians, 480×352, GGRt's
sh_degree 4 / 25 coefficients, band 4 evaluated) on a SAMPLE of tiles — every `stride`-th tile of the frame, the sampler of
`bench.py`'s `cpu_baseline_torch` — for the image and, with the upstream gradient masked to the sampled tiles' pixels on
BOTH sides, for every gradient tensor (a gradient is a sum over pixels: restricted to the same pixels the two sums are
the same quantity, over all 1 M Gaussians).  The preprocess of the torch leg runs on the whole scene, so radii and the
number of list entries are compared for the full frame.

Costs ≈ 10-20 s of host time per case (the torch leg's whole-scene preprocess + 64-bit key sort take seconds; its blend
is what a sample bounds).  GGR_FULLSIZE_TORCH_STRIDE=<n> overrides the tile stride (1 = every tile: minutes).  The measured
figures land in gpurun_out/parity_metrics.jsonl and DESIGN.md §3.
"""
import os

import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import CONFIGS, make_scene, upstream_gradient
from oracle import torch_raster as tr
from tests.helpers import GRAD_RTOL_ALL, hip_forward_backward, record_metric, rel_l2

pytestmark = pytest.mark.gpu

# config, tile stride of the sample (C3: 628 of 8160 tiles, C5′: 220 of 660), highest SH band evaluated
CASES = [("C3", 13, 3), ("C5p", 3, 4)]


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("name,stride,cap", CASES, ids=[c[0] for c in CASES])
def test_full_size_sampled_tiles_match_torch_autograd(name, stride, cap):
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    stride = int(os.environ.get("GGR_FULLSIZE_TORCH_STRIDE", stride))
    sc = make_scene(seed=0, **CONFIGS[name])
    W, H = sc.width, sc.height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    sel = lambda tx, ty: (ty * gx + tx) % stride == 0
    tiles = [(t % gx, t // gx) for t in range(gx * gy) if t % stride == 0]
    mask = torch.zeros(H, W, dtype=torch.bool)
    for tx, ty in tiles:
        mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    dL = upstream_gradient(W, H, seed=77) * mask            # zero outside the sampled tiles, on both sides

    # ---- independent leg: whole-scene preprocess + key sort, blend of the sampled tiles, autograd
    leaf = lambda t: t.clone().requires_grad_(True)
    m, cov, op, sh = leaf(sc.means3D), leaf(sc.cov3D), leaf(sc.opacities), leaf(sc.shs)
    pre = tr.preprocess(m, op, sc.viewmatrix, sc.projmatrix, sc.campos, W, H, sc.tanfovx, sc.tanfovy, sc.sh_degree,
                        shs=sh, cov3D_precomp=cov, sh_cap=cap)
    point_list, ranges, keys, N = tr.bin_tiles(pre, W, H)
    color_t, *_ = tr.blend(pre, point_list, ranges, sc.bg, W, H, tile_filter=sel)
    (color_t * dL).sum().backward()
    ref = dict(means3D=m.grad.numpy(), cov3D_precomp=cov.grad.numpy(), opacities=op.grad.numpy(), shs=sh.grad.numpy())

    # ---- the product path: whole frame, the REFERENCE's tile rects (what the torch leg bins by)
    color_h, radii_h, _, grads = hip_forward_backward(sc, dL, sh_max_degree=cap, reference_rects=True)
    from ggrt_official_amd.rasterizer import debug_forward_state
    s = sc.to("cuda:0")
    st = debug_forward_state(s.means3D, s.opacities, s.settings()._replace(reference_rects=True, sh_max_degree=cap),
                             shs=s.shs, cov3D_precomp=s.cov3D)

    # discrete outputs of the whole frame
    assert np.array_equal(radii_h, pre["radii"].to(torch.int32).numpy())
    assert int(st["num_rendered"]) == int(N)
    # image on the sampled tiles
    mk = mask.numpy()
    d = np.abs(color_h - color_t.detach().numpy())[:, mk]
    peak = float(np.abs(color_t.detach().numpy()[:, mk]).max())
    frac_off = float((d > 1e-4 * max(peak, 1.0)).mean())
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = float("inf") if mse == 0 else -10.0 * np.log10(mse)
    record_metric(f"fullsize_torch:{name}:image", kind=0, max_abs=float(d.max()), frac_off=frac_off, psnr=min(psnr, 999.0),
                  sampled_tiles=len(tiles), sampled_pixels=int(mk.sum()))
    assert frac_off <= 5e-5 and psnr >= 110.0, (frac_off, psnr, float(d.max()))
    # gradients of all P Gaussians w.r.t. the sampled pixels: rel-L2 over all rows (north-star 1e-3), and with the
    # 1e-5·P rows of largest error set aside (α-threshold flips between differently rounded evaluations)
    P = sc.means3D.shape[0]
    for k in ("means3D", "cov3D_precomp", "opacities", "shs"):
        a = np.asarray(grads[k], np.float64).reshape(P, -1)
        b = np.asarray(ref[k], np.float64).reshape(P, -1)
        r_all = rel_l2(a, b)
        err = np.linalg.norm(a - b, axis=1)
        keep = np.ones(P, bool)
        keep[np.argpartition(-err, max(1, int(1e-5 * P)) - 1)[:max(1, int(1e-5 * P))]] = False
        r = float(np.linalg.norm((a - b)[keep]) / max(np.linalg.norm(b[keep]), 1e-30))
        record_metric(f"fullsize_torch:{name}:{k}", kind=1, rel_l2=r, rel_l2_all=r_all, rows_nonzero=int((np.abs(b).sum(1) > 0).sum()))
        assert r_all <= GRAD_RTOL_ALL, f"{name} grad {k}: rel-L2 over all rows {r_all:.3e}"
        assert r <= 1e-4, f"{name} grad {k}: rel-L2 {r:.3e} ({int(1e-5 * P)} rows set aside; all rows {r_all:.3e})"
