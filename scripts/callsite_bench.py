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


def measure_views(dev="cuda:0", steps=10, warmup=3, config="C5p", views=4):
    """SURVEY.md §8f-2 at GGRt's shape: `views` target views of the SAME Gaussians (pixelSplat-style samples render
    several target views; reference decoder_splatting_cuda.py:40-60), colour + depth, forward + backward:

      per_view_loop   one rasterizer call per view (the reference's structure, with this build's fused inputs),
                      autograd adds the per-view gradient tensors
      batched         ONE launch set for all views (`rasterize_views`)
    and the one-view time of the same path for scale."""
    import math
    from ggrt_official_amd import splatting as sp
    from ggrt_official_amd.synthetic import CONFIGS, make_scene
    cfg = CONFIGS[config]
    sc = make_scene(**cfg).to(dev)
    P, H, W = sc.means3D.shape[0], sc.height, sc.width
    ext = []
    for k in range(views):  # small pose jitter around the scene's camera: every view sees (almost) all Gaussians
        a = 0.02 * math.sin(1.3 * k)
        T = torch.eye(4)
        T[0, 0], T[0, 2], T[2, 0], T[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        T[:3, 3] = torch.tensor([0.03 * k, -0.02 * k, 0.0])
        ext.append(T)
    ext = torch.stack(ext).to(dev)
    fx, fy = 0.5 / sc.tanfovx, 0.5 / sc.tanfovy
    intr = torch.tensor([[fx, 0, 0.5], [0, fy, 0.5], [0, 0, 1]], device=dev)[None].expand(views, 3, 3).contiguous()
    near, far = torch.full((views,), 1.0, device=dev), torch.full((views,), 100.0, device=dev)
    cov = torch.zeros(P, 3, 3, device=dev)
    for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
        cov[:, i, j] = sc.cov3D[:, k]
        cov[:, j, i] = sc.cov3D[:, k]
    leaves = [t.clone().requires_grad_() for t in (sc.means3D[None], cov[None], sc.shs.permute(0, 2, 1).contiguous()[None],
                                                   sc.opacities[:, 0][None])]
    gs = sp.Gaussians(means=leaves[0], covariances=leaves[1], harmonics=leaves[2], opacities=leaves[3])
    bg = torch.zeros(views, 3, device=dev)
    g = torch.Generator().manual_seed(0)
    dL = (torch.randn(views, 3, H, W, generator=g) / (3 * H * W)).to(dev)
    dD = (torch.randn(views, H, W, generator=g) / (H * W)).to(dev)

    def run(nv, batched):
        c, d = sp.render_views_fused(ext[:nv], intr[:nv], near[:nv], far[:nv], (H, W), bg[:nv], gs, [0] * nv, "depth",
                                     batched=batched)
        torch.autograd.backward([c, d], [dL[:nv], dD[:nv]])

    out = {"workload": f"{config}: {P} Gaussians, {W}x{H}, d_sh {sc.shs.shape[1]}, colour + depth, fwd+bwd, {views} views "
                       f"of the same Gaussians"}
    for name, fn in (("one_view_ms", lambda: run(1, False)), ("per_view_loop_ms", lambda: run(views, False)),
                     ("batched_ms", lambda: run(views, True))):
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
        out[name] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    out["batched_over_one_view"] = round(out["batched_ms"] / out["one_view_ms"], 2)
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(measure(), indent=1))
    print(json.dumps(measure_views(), indent=1))
