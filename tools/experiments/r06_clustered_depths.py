"""dev: the two forms of the depth sort on frames whose depths CLUSTER (the per-tile sort's LSD route): C3-sized scenes with the
Gaussians on k depth shells of relative thickness t instead of log-uniform depths."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggrt_official_amd import GaussianRasterizer
from ggrt_official_amd.synthetic import make_scene, upstream_gradient
dev = "cuda:0"
P, W, H = 1_000_000, 1920, 1080
for shells, thick in ((0, 0.0), (2, 1e-2), (2, 1e-4), (1, 1e-3), (4, 1e-3), (3, 0.0)):
    sc = make_scene(P, W, H, sh_degree=3, profile="A", seed=0)
    if shells:
        g = torch.Generator().manual_seed(1)
        base = torch.tensor([2.0, 9.0, 25.0, 45.0])[torch.randint(0, shells, (P,), generator=g)]
        z = base * (1.0 + thick * torch.rand(P, generator=g))
        s = (z / sc.means3D[:, 2])
        sc.means3D *= s[:, None]
        sc.cov3D *= (s * s)[:, None]
    sc = sc.to(dev)
    dL = upstream_gradient(W, H, device=dev)
    leaves = [t.clone().requires_grad_() for t in (sc.means3D, sc.shs, sc.opacities, sc.cov3D)]
    out = {"shells": shells, "thickness": thick}
    for mode in ("global", "per_tile", "auto"):
        rast = GaussianRasterizer(sc.settings()._replace(depth_sort=mode))
        def step():
            for t in leaves: t.grad = None
            c, _, _ = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), opacities=leaves[2], shs=leaves[1], cov3D_precomp=leaves[3])
            c.backward(dL)
            return c
        for _ in range(20): step()
        torch.cuda.synchronize()
        evs = []
        for _ in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for t in leaves: t.grad = None
            e0.record()
            c, _, _ = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), opacities=leaves[2], shs=leaves[1], cov3D_precomp=leaves[3])
            e1.record(); c.backward(dL)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        out[mode + "_fwd_ms"] = round(sorted(a.elapsed_time(b) for a, b in evs)[20], 4)
        out[mode + "_img"] = c.detach().clone()
    out["auto_img_same"] = bool(torch.equal(out["global_img"], out.pop("auto_img")))
    from ggrt_official_amd.rasterizer import last_forward_binning, sort_watch_stats, clear_list_hints
    out["auto_ended_on"] = last_forward_binning()[0]
    out["watch"] = [tuple(round(x, 3) if isinstance(x, float) else x for x in v) for v in sort_watch_stats().values()]
    clear_list_hints()
    out["same_image"] = bool(torch.equal(out.pop("global_img"), out.pop("per_tile_img")))
    print(json.dumps(out), flush=True)
