"""Call-site benchmark at GGRt's shape (C5': 1.01 M pixel-aligned Gaussians, 480x352, d_sh 25, colour + depth,
forward + backward through the call-site layer, Gaussian tensors in GGRt's own layouts):

  reference_literal  two rasterizations per view + torch pre-processing of the Gaussian tensors, as reference
                     decoder_splatting_cuda.py:29-61 / cuda_splatting.py:49-128,227-269 do it
  one_pass           colour + depth from one rasterization, torch pre-processing kept
  fused              `render_views_fused`: input forms + device camera (the decoder's default path)

Run on the GPU box:  python scripts/callsite_bench.py       (bench.py embeds the same numbers as `callsite_ggrt_shape`)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def measure(dev="cuda:0", steps=10, warmup=3, config="C5p"):
    from ggrt_official_amd import splatting as sp
    from ggrt_official_amd.synthetic import CONFIGS, make_scene
    cfg = CONFIGS[config]
    sc = make_scene(**cfg).to(dev)
    P, H, W = sc.means3D.shape[0], sc.height, sc.width
    c2w = torch.eye(4, device=dev)[None]
    fx, fy = 0.5 / sc.tanfovx, 0.5 / sc.tanfovy
    intr = torch.tensor([[fx, 0, 0.5], [0, fy, 0.5], [0, 0, 1]], device=dev)[None]
    near, far = torch.tensor([1.0], device=dev), torch.tensor([100.0], device=dev)
    cov = torch.zeros(P, 3, 3, device=dev)
    for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
        cov[:, i, j] = sc.cov3D[:, k]
        cov[:, j, i] = sc.cov3D[:, k]
    leaves = [t.clone().requires_grad_() for t in (sc.means3D[None], cov[None], sc.shs.permute(0, 2, 1).contiguous()[None],
                                                   sc.opacities[:, 0][None])]
    means, covs, harm, op = leaves
    bg = torch.zeros(1, 3, device=dev)
    g = torch.Generator().manual_seed(0)
    dL = (torch.randn(1, 3, H, W, generator=g) / (3 * H * W)).to(dev)
    dD = (torch.randn(1, H, W, generator=g) / (H * W)).to(dev)

    def reference_literal():
        c = sp.render_cuda(c2w, intr, near, far, (H, W), bg, means, covs, harm, op)
        d = sp.render_depth_cuda(c2w, intr, near, far, (H, W), means, covs, op, mode="depth")
        torch.autograd.backward([c, d], [dL, dD])

    def one_pass():
        c, d = sp.render_color_and_depth(c2w, intr, near, far, (H, W), bg, means, covs, harm, op, "depth")
        torch.autograd.backward([c, d], [dL, dD])

    gs = sp.Gaussians(means=means, covariances=covs, harmonics=harm, opacities=op)

    def fused():
        c, d = sp.render_views_fused(c2w, intr, near, far, (H, W), bg, gs, [0], "depth")
        torch.autograd.backward([c, d], [dL, dD])

    out = {"workload": f"{config}: {P} Gaussians, {W}x{H}, d_sh {sc.shs.shape[1]}, colour + depth, fwd+bwd, 1 view"}
    for name, fn in (("reference_literal", reference_literal), ("one_pass", one_pass), ("fused", fused)):
        for _ in range(warmup):
            for t in leaves:
                t.grad = None
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for t in leaves:
                t.grad = None
            fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(measure(), indent=1))
