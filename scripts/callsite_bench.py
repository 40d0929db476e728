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


def measure_views(dev="cuda:0", steps=10, warmup=3, config="C5p", views=4, only=None):
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
        if only and name != only:   # (kernel traces of one leg alone)
            continue
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
    if not only:
        out["batched_over_one_view"] = round(out["batched_ms"] / out["one_view_ms"], 2)
    return out


def measure_views_batched_only(steps=10, warmup=3):
    return measure_views(steps=steps, warmup=warmup, only="batched_ms")


def measure_sets(dev="cuda:0", steps=10, warmup=3, config="C5p", sets=4):
    """The reference's `(b v)` flattening with per-batch-element Gaussians (decoder_splatting_cuda.py:40-60) at GGRt's
    shape: `sets` independent (Gaussians, camera) problems, one view each, colour + depth, forward + backward —
      per_set_loop   one rasterizer call per problem (round 2's `render_views_fused`: a Python loop over batch elements)
      one_launch_set ONE launch set over all of them (GgrViews.num_sets)
    and one problem alone for scale."""
    import math
    from ggrt_official_amd import splatting as sp
    from ggrt_official_amd.synthetic import CONFIGS, make_scene
    cfg = CONFIGS[config]
    scs = [make_scene(seed=s, **cfg).to(dev) for s in range(sets)]
    P, H, W = scs[0].means3D.shape[0], scs[0].height, scs[0].width
    ext = torch.eye(4, device=dev)[None].repeat(sets, 1, 1)
    fx, fy = 0.5 / scs[0].tanfovx, 0.5 / scs[0].tanfovy
    intr = torch.tensor([[fx, 0, 0.5], [0, fy, 0.5], [0, 0, 1]], device=dev)[None].expand(sets, 3, 3).contiguous()
    near, far = torch.full((sets,), 1.0, device=dev), torch.full((sets,), 100.0, device=dev)

    def cov33(sc):
        cov = torch.zeros(P, 3, 3, device=dev)
        for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
            cov[:, i, j] = sc.cov3D[:, k]
            cov[:, j, i] = sc.cov3D[:, k]
        return cov
    def make_leaves(nb):   # own leaf tensors per problem count: a slice of a bigger leaf would add its backward (a
        return [t.clone().requires_grad_() for t in (   # zero-filled full-size tensor per slice) to the measurement
            torch.stack([s.means3D for s in scs[:nb]]), torch.stack([cov33(s) for s in scs[:nb]]),
            torch.stack([s.shs.permute(0, 2, 1).contiguous() for s in scs[:nb]]),
            torch.stack([s.opacities[:, 0] for s in scs[:nb]]))]
    leaf_sets = {1: make_leaves(1), sets: make_leaves(sets)}
    leaves = [t for ls in leaf_sets.values() for t in ls]
    bg = torch.zeros(sets, 3, device=dev)
    g = torch.Generator().manual_seed(0)
    dL = (torch.randn(sets, 3, H, W, generator=g) / (3 * H * W)).to(dev)
    dD = (torch.randn(sets, H, W, generator=g) / (H * W)).to(dev)

    def run(nb, batched):
        lv = leaf_sets[nb]
        gs = sp.Gaussians(means=lv[0], covariances=lv[1], harmonics=lv[2], opacities=lv[3])
        c, d = sp.render_views_fused(ext[:nb], intr[:nb], near[:nb], far[:nb], (H, W), bg[:nb], gs, list(range(nb)), "depth",
                                     batched=batched)
        torch.autograd.backward([c, d], [dL[:nb], dD[:nb]])

    out = {"workload": f"{config}: {sets} independent problems of {P} Gaussians, {W}x{H}, d_sh {scs[0].shs.shape[1]}, "
                       f"colour + depth, fwd+bwd, one view each"}
    for name, fn in (("one_problem_ms", lambda: run(1, False)), ("per_set_loop_ms", lambda: run(sets, False)),
                     ("one_launch_set_ms", lambda: run(sets, True))):
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
    out["launch_set_over_one_problem"] = round(out["one_launch_set_ms"] / out["one_problem_ms"], 2)
    return out


def measure_eval_sets(dev="cuda:0", steps=10, warmup=3, config="C4p", sets=4):
    """GGRt's evaluation loop (eval/eval_ggrt.py:317: forward only under `torch.no_grad()`) at its LLFF shape, `sets`
    independent frames per launch set against one frame per call: frames per second."""
    from ggrt_official_amd import splatting as sp
    from ggrt_official_amd.synthetic import CONFIGS, make_scene
    cfg = CONFIGS[config]
    scs = [make_scene(seed=s, **cfg).to(dev) for s in range(sets)]
    P, H, W = scs[0].means3D.shape[0], scs[0].height, scs[0].width
    ext = torch.eye(4, device=dev)[None].repeat(sets, 1, 1)
    fx, fy = 0.5 / scs[0].tanfovx, 0.5 / scs[0].tanfovy
    intr = torch.tensor([[fx, 0, 0.5], [0, fy, 0.5], [0, 0, 1]], device=dev)[None].expand(sets, 3, 3).contiguous()
    near, far = torch.full((sets,), 1.0, device=dev), torch.full((sets,), 100.0, device=dev)

    def cov33(sc):
        cov = torch.zeros(P, 3, 3, device=dev)
        for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
            cov[:, i, j] = sc.cov3D[:, k]
            cov[:, j, i] = sc.cov3D[:, k]
        return cov
    gs_all = sp.Gaussians(means=torch.stack([s.means3D for s in scs]), covariances=torch.stack([cov33(s) for s in scs]),
                          harmonics=torch.stack([s.shs.permute(0, 2, 1).contiguous() for s in scs]),
                          opacities=torch.stack([s.opacities[:, 0] for s in scs]))
    gs_one = sp.Gaussians(means=gs_all.means[:1].clone(), covariances=gs_all.covariances[:1].clone(),
                          harmonics=gs_all.harmonics[:1].clone(), opacities=gs_all.opacities[:1].clone())
    bg = torch.zeros(sets, 3, device=dev)

    def run(gs, nb):
        with torch.no_grad():
            sp.render_views_fused(ext[:nb], intr[:nb], near[:nb], far[:nb], (H, W), bg[:nb], gs, list(range(nb)), None)

    out = {"workload": f"{config}: {P} Gaussians per frame, {W}x{H}, d_sh {scs[0].shs.shape[1]}, colour, forward only (no_grad)"}
    for name, fn, frames in (("one_frame_per_call", lambda: run(gs_one, 1), 1), (f"{sets}_frames_per_launch_set", lambda: run(gs_all, sets), sets)):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[name] = {"ms_per_call": round(ms, 3), "frames_per_s": round(frames * 1e3 / ms, 1)}
    return out


def measure_window(dev="cuda:0", steps=10, warmup=3, config="C5p", crop=2):
    """The reference's deferred back-propagation cell (finetune_ggrt_stable.py:126-142) at GGRt's shape: a full-frame
    render whose backward sees a gradient that is zero outside ONE cell of a crop × crop grid.  Per-stage times of the
    rasterizer (HIP events): the dense-gradient step, the windowed gradient (the backward's zero-gradient skip), and
    the windowed gradient with the forward scissored to the cell as well."""
    from ggrt_official_amd import GaussianRasterizer
    from ggrt_official_amd.rasterizer import profile_stages
    from ggrt_official_amd.synthetic import CONFIGS, make_scene, upstream_gradient
    cfg = CONFIGS[config]
    sc = make_scene(**cfg).to(dev)
    H, W = sc.height, sc.width
    leaf = lambda t: t.clone().requires_grad_(True)
    means, cov, op, shs = leaf(sc.means3D), leaf(sc.cov3D), leaf(sc.opacities), leaf(sc.shs)
    sink = torch.zeros_like(means, requires_grad=True)
    dL = upstream_gradient(W, H, seed=3, device=dev)
    oh, ow = H // crop, W // crop
    win = (ow * (crop - 1), oh * (crop - 1), ow * crop, oh * crop)   # the last cell
    mask = torch.zeros(H, W, device=dev)
    mask[win[1]:win[3], win[0]:win[2]] = 1.0
    out = {"workload": f"{config}: one cell of a {crop}x{crop} grid ({ow}x{oh} px of {W}x{H}), fwd+bwd"}
    for name, grad, scissor in (("dense", dL, None), ("windowed_gradient", dL * mask, None),
                                ("windowed_gradient_scissored_forward", dL * mask, win)):
        rast = GaussianRasterizer(sc.settings()._replace(scissor=scissor))

        def step():
            for t in (means, cov, op, shs, sink):
                t.grad = None
            c, _, _ = rast(means3D=means, means2D=sink, opacities=op, shs=shs, cov3D_precomp=cov)
            c.backward(grad)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) < 0.04:   # (the set-up above left the device idle: power-state ramp, NOTES r4)
            step()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))[steps // 2]   # median: one host pause in 10 steps is 0.5 ms of mean
        with profile_stages() as prof:
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        st = prof.as_dict()
        out[name] = {"ms_per_step": round(ms, 3), "fwd_ms": round(sum(v for k, v in st.items() if k.startswith("fwd_")), 4),
                     "bwd_blend_ms": round(st["bwd_blend_ms"], 4), "bwd_preprocess_ms": round(st["bwd_preprocess_ms"], 4)}
    out["bwd_blend_windowed_over_dense"] = round(out["windowed_gradient"]["bwd_blend_ms"] / out["dense"]["bwd_blend_ms"], 3)
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(measure_eval_sets(), indent=1))
    print(json.dumps(measure_sets(), indent=1))
    print(json.dumps(measure_window(), indent=1))
    print(json.dumps(measure_window(config="C3"), indent=1))
    sys.exit(0)
    print(json.dumps(measure(), indent=1))
    print(json.dumps(measure_views(), indent=1))
