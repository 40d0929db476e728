#!/usr/bin/env python3
"""dev (GPU box, library built with GGR_EXTRA_HIPCC_FLAGS=-DGGR_DEV_COUNTERS): what the two blend kernels walk on one frame —
survivors of the forward's quadrant culls against the (quadrant, entry) slots of the backward's.
usage: GGR_SKIP_SOURCE_HASH=1 python tools/blend_slot_counts.py [config ...]"""
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ggrt_official_amd import GaussianRasterizer, _lib  # noqa: E402
from ggrt_official_amd.synthetic import CONFIGS, make_scene, upstream_gradient  # noqa: E402

lib = _lib.load()
for name in (sys.argv[1:] or ["C3"]):
    sc = make_scene(seed=0, **CONFIGS[name]).to("cuda:0")
    dL = upstream_gradient(sc.width, sc.height).to("cuda:0")
    leaves = [t.clone().requires_grad_() for t in (sc.means3D, sc.shs, sc.opacities, sc.cov3D)]
    out = (C.c_uint64 * 8)()
    lib.ggr_debug_counters(out, 1)
    color, radii, depth = GaussianRasterizer(sc.settings())(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]),
                                                            opacities=leaves[2], shs=leaves[1], cov3D_precomp=leaves[3])
    rc = lib.ggr_debug_counters(out, 0)
    fwd = list(out)[:4]
    (color * dL).sum().backward()
    rc = lib.ggr_debug_counters(out, 1)
    bwd = list(out)[4:]
    rec = dict(config=name, rc=rc, fwd_survivors_listed=fwd[0], fwd_survivors_walked=fwd[1], fwd_pairs_composited=fwd[2],
               fwd_wave_batches=fwd[3], bwd_slots=bwd[0], bwd_slots_without_a_valid_lane=bwd[1], bwd_valid_pairs=bwd[2],
               bwd_wave_batches=bwd[3])
    if bwd[0]:
        rec.update(bwd_over_fwd_walked=round(bwd[0] / max(fwd[1], 1), 4), bwd_dead_fraction=round(bwd[1] / bwd[0], 4),
                   bwd_live_lanes_per_slot=round(bwd[2] / bwd[0], 2), fwd_live_lanes_per_survivor=round(fwd[2] / max(fwd[1], 1), 2))
    print(json.dumps(rec))
