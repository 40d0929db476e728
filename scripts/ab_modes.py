#!/usr/bin/env python3
"""dev (GPU box): fwd+bwd step and forward time of several shapes under both forms of the depth sort.
usage: python scripts/ab_modes.py [name:P:W:H[:layout] ...]"""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggrt_official_amd import GaussianRasterizer
from ggrt_official_amd.rasterizer import last_forward_binning
from ggrt_official_amd.synthetic import make_scene, upstream_gradient

shapes = sys.argv[1:] or ["C3:1000000:1920:1080", "C3low:1000000:1920:1080:lower_half", "2M:2000000:1920:1080",
                          "4K:1000000:3840:2160", "sparse:200000:1920:1080", "720p:500000:1280:720"]
dev = "cuda:0"
for spec in shapes:
    f = spec.split(":")
    if len(f) == 1:   # a named configuration of ggrt_official_amd.synthetic.CONFIGS
        from ggrt_official_amd.synthetic import CONFIGS
        sc = make_scene(seed=0, **CONFIGS[spec]).to(dev)
        W, H = sc.width, sc.height
    else:
        name, P, W, H = f[0], int(f[1]), int(f[2]), int(f[3])
        layout = f[4] if len(f) > 4 else "uniform"
        sc = make_scene(P, W, H, sh_degree=3, profile="A", seed=0, layout=layout).to(dev)
    dL = upstream_gradient(W, H, device=dev)
    leaves = [t.clone().requires_grad_() for t in (sc.means3D, sc.shs, sc.opacities, sc.cov3D)]
    out = {"shape": spec}
    for mode in ("global", "per_tile"):
        rast = GaussianRasterizer(sc.settings()._replace(depth_sort=mode))
        def step():
            for t in leaves: t.grad = None
            c, _, _ = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), opacities=leaves[2], shs=leaves[1], cov3D_precomp=leaves[3])
            c.backward(dL)
        for _ in range(30): step()
        torch.cuda.synchronize()
        evs = []
        for _ in range(60):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            for t in leaves: t.grad = None
            e0.record()
            c, _, _ = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), opacities=leaves[2], shs=leaves[1], cov3D_precomp=leaves[3])
            e1.record(); c.backward(dL); e2.record()
            evs.append((e0, e1, e2))
        torch.cuda.synchronize()
        fw = sorted(a.elapsed_time(b) for a, b, _ in evs)[len(evs) // 2]
        st = sorted(a.elapsed_time(c_) for a, _, c_ in evs)[len(evs) // 2]
        out[mode] = {"fwd_ms": round(fw, 4), "step_ms": round(st, 4), "used": last_forward_binning()}
    out["per_tile_over_global_step"] = round(out["per_tile"]["step_ms"] / out["global"]["step_ms"], 4)
    out["per_tile_over_global_fwd"] = round(out["per_tile"]["fwd_ms"] / out["global"]["fwd_ms"], 4)
    print(json.dumps(out), flush=True)
    del sc, leaves
    torch.cuda.empty_cache()
