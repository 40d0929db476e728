#!/usr/bin/env python3
"""bench.py — Gaussian raster fwd+bwd Mpix/s (BASELINE.json metric) on MI355X.

A "step" = one forward + one backward of the rasterizer over one synthetic frame whose inputs are
already resident in HBM (BASELINE.json config 3: 1 M Gaussians, 1920×1080, SH degree 3, profile A —
``ggrt_official_amd/synthetic.py``).  The backward produces the reference's gradient set (means3D, cov3D,
SH, opacity, means2D) plus the camera gradient (viewmatrix / projmatrix / campos).  With N > 1 ranks
every rank renders its OWN frame (frames shard one-per-GPU, SURVEY.md §8e: the per-frame Gaussians are
never exchanged) and the step ends with the path's one exchange: a mean all-reduce of the camera
gradient over RCCL.  ``--grad-buffer-floats 65000000`` additionally all-reduces a stand-in for GGRt's
encoder + pose-network gradients (≈260 MB); that is off by default because those modules
are outside the measured path.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying two extra objects:
  "roofline":     the dominant kernel's algorithmic bytes / its live HIP-event duration vs 8 TB/s
  "cpu_baseline": the PyTorch-CPU restatement (oracle/torch_raster.py) timed on this host on a bounded
                  sample of the same workload (rank 0, N = 1 only)

Launch (N > 1):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
                 --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (≈6.3 TB/s achievable)


def algorithmic_bytes(P: int, N: int, W: int, H: int, K: int, M: int) -> dict:
    """SURVEY.md §8(d) per-unit figures × the units one launch processes."""
    return {
        # per-stage split of B_fwd = P(12+24+4+12K) + N(12+12) + N·40 + W·H·20
        "fwd_preprocess": P * (12 + 24 + 4 + 12 * K),
        "fwd_binning": N * 24,  # SURVEY's figure (12-B pair written + read once); this build writes 4 B/entry
        "fwd_blend": N * 40 + W * H * 20,
        # B_bwd = W·H·20 + N·40 + P(12+24+4+12K) + P(12+12+24+4+12M)
        "bwd_blend": W * H * 20 + N * 40,
        "bwd_preprocess": P * (12 + 24 + 4 + 12 * K) + P * (12 + 12 + 24 + 4 + 12 * M),
    }


def cpu_baseline(cfg: dict, seed: int, budget_s: float = 20.0) -> dict:
    """Times the PyTorch-CPU restatement (fwd + autograd bwd) on a bounded sample of the SAME scene:
    the full preprocess + key sort, and the blend fwd+bwd on an evenly strided subset of tiles sized
    from a 4-tile probe so the whole leg stays within ~budget_s.  The per-frame time is
    t_pre+sort + t_blend(sample)·tiles/sample and is reported as such (``sample``)."""
    from ggrt_official_amd.synthetic import make_scene, upstream_gradient
    from oracle import torch_raster as tr

    # per-tile tensors are small ([list, 256]); beyond a few dozen threads torch's intra-op
    # parallelism only adds synchronisation cost, so cap the thread count and report what was used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sc = make_scene(seed=seed, **cfg)
    W, H = sc.width, sc.height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ntiles = gx * gy
    leaf = lambda t: t.clone().requires_grad_(True)
    m, cov, op, sh = leaf(sc.means3D), leaf(sc.cov3D), leaf(sc.opacities), leaf(sc.shs)
    t0 = time.perf_counter()
    pre = tr.preprocess(m, op, sc.viewmatrix, sc.projmatrix, sc.campos, W, H, sc.tanfovx, sc.tanfovy,
                        sc.sh_degree, shs=sh, cov3D_precomp=cov)
    point_list, ranges, keys, N = tr.bin_tiles(pre, W, H)
    t_prebin = time.perf_counter() - t0
    log(f"cpu_baseline: threads={cores} preprocess+sort {t_prebin:.2f}s N={N}")
    dL = upstream_gradient(W, H)

    def run(stride, offset):
        sel = lambda tx, ty: (ty * gx + tx) % stride == offset
        n = len([t for t in range(ntiles) if t % stride == offset])
        t0 = time.perf_counter()
        color, *_ = tr.blend(pre, point_list, ranges, sc.bg, W, H, tile_filter=sel)
        t_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        (color * dL).sum().backward(retain_graph=True)
        return n, t_f, time.perf_counter() - t0

    # probe: ~4 tiles spread over the image (the backward includes the per-Gaussian autograd pass)
    n_p, tf_p, tb_p = run(max(1, ntiles // 4), (ntiles // 8) % max(1, ntiles // 4))
    per_tile = (tf_p + max(tb_p - 0.0, 0.0)) / max(n_p, 1)
    log(f"cpu_baseline: probe {n_p} tiles fwd {tf_p:.2f}s bwd {tb_p:.2f}s")
    n_target = int(max(4, min(ntiles, budget_s / max(per_tile, 1e-4))))
    stride = max(1, ntiles // n_target)
    for t in (m, cov, op, sh):
        t.grad = None
    n_s, t_f, t_b = run(stride, 0)
    frac = n_s / ntiles
    t_frame = t_prebin + (t_f + t_b) / frac
    log(f"cpu_baseline: sample {n_s}/{ntiles} tiles fwd {t_f:.2f}s bwd {t_b:.2f}s -> frame {t_frame:.1f}s")
    return {
        "value": round(W * H / t_frame / 1e6, 6), "unit": "Mpix/s", "cores": cores, "kind": "port",
        "sample": (f"oracle/torch_raster.py (PyTorch CPU fp32, {cores} threads, autograd bwd) on the same scene: "
                   f"full preprocess+sort of P={cfg['num_points']} / N={N} ({t_prebin:.2f}s) + blend fwd+bwd on "
                   f"{n_s}/{ntiles} tiles (every {stride}th; {t_f:.2f}s + {t_b:.2f}s, bwd includes the per-Gaussian "
                   f"autograd), scaled by {1 / frac:.1f} to one frame"),
        "frame_s_estimated": round(t_frame, 3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", help="key of ggrt_official_amd.synthetic.CONFIGS")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the informational HIP-graph replay leg")
    ap.add_argument("--no-callsite", action="store_true", help="skip the informational GGRt-shape call-site leg")
    ap.add_argument("--grad-buffer-floats", type=int, default=0,
                    help="N>1 only: ALSO all-reduce a flat fp32 buffer of this size per step, a stand-in for GGRt's "
                         "encoder + pose-network gradients (≈65_000_000, SURVEY.md §5); off by default because those "
                         "modules are outside the measured path and nothing in this benchmark could overlap it")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra untimed steps with per-stage HIP events")
    ap.add_argument("--dist-backend", default=None,
                    help="debug: process-group backend (default: nccl = RCCL).  `gloo` together with `--device 0` "
                         "lets several ranks share ONE GPU to dry-run the N>1 control flow on a 1-GPU box")
    ap.add_argument("--device", type=int, default=None, help="debug: CUDA device index instead of LOCAL_RANK")
    args = ap.parse_args()

    from ggrt_official_amd import GaussianRasterizer
    from ggrt_official_amd.rasterizer import profile_stages
    from ggrt_official_amd.synthetic import CONFIGS, make_scene, upstream_gradient
    from ggrt_official_amd import parallel

    rank, world, local = parallel.init_from_env(args.gpus, backend=args.dist_backend)
    if args.device is not None:
        local = args.device
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    cfg = CONFIGS[args.config]
    sc = make_scene(seed=rank, **cfg).to(dev)       # one frame per rank, inputs resident in HBM
    W, H = sc.width, sc.height
    dL = upstream_gradient(W, H, seed=1234 + rank, device=dev)
    # camera tensors are leaves too: the step produces dL/d(viewmatrix, projmatrix, campos) — the one
    # gradient of this path that data-parallel ranks share (the per-frame Gaussians are not shared)
    view = sc.viewmatrix.clone().requires_grad_(True)
    proj = sc.projmatrix.clone().requires_grad_(True)
    campos = sc.campos.clone().requires_grad_(True)
    rs = sc.settings()._replace(viewmatrix=view, projmatrix=proj, campos=campos)
    rast = GaussianRasterizer(rs)
    means = sc.means3D.clone().requires_grad_(True)
    cov = sc.cov3D.clone().requires_grad_(True)
    op = sc.opacities.clone().requires_grad_(True)
    shs = sc.shs.clone().requires_grad_(True)
    means2D = torch.zeros_like(means, requires_grad=True)
    leaves = (means, cov, op, shs, means2D, view, proj, campos)
    pose_buf = torch.zeros(35, device=dev)
    grad_buf = torch.zeros(args.grad_buffer_floats, device=dev) if (world > 1 and args.grad_buffer_floats) else None

    def step():
        for t in leaves:
            t.grad = None
        color, radii, depth = rast(means3D=means, means2D=means2D, opacities=op, shs=shs, cov3D_precomp=cov)
        color.backward(dL)  # the upstream gradient dL/dcolor goes straight into the rasterizer's backward
        if world > 1:
            # the path's exchange step: mean all-reduce of the camera gradient over RCCL/xGMI
            torch.cat([view.grad.reshape(-1), proj.grad.reshape(-1), campos.grad.reshape(-1)], out=pose_buf)
            parallel.allreduce_mean_(pose_buf)
            if grad_buf is not None:  # optional stand-in for the encoder + pose-network gradients
                parallel.allreduce_mean_(grad_buf)
        return color

    log(f"scene {args.config} resident on {dev}; warmup {args.warmup}")
    for i in range(args.warmup):
        t_w = time.perf_counter()
        step()
        torch.cuda.synchronize(dev)
        log(f"warmup step {i}: {(time.perf_counter() - t_w) * 1e3:.2f} ms")
    parallel.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, dev)
    log(f"timed {args.steps} steps: {elapsed / args.steps * 1e3:.3f} ms/step")

    # per-stage HIP-event timing on the launch stream (untimed extra steps)
    with profile_stages() as prof:
        for _ in range(max(args.profile_steps, 1)):
            step()
    torch.cuda.synchronize(dev)
    stages = prof.as_dict()
    log("stages: " + ", ".join(f"{k}={v:.3f}" for k, v in stages.items()))
    # num_rendered of this rank's frame
    from ggrt_official_amd.rasterizer import debug_forward_state
    N = debug_forward_state(sc.means3D, sc.opacities, rs, shs=sc.shs, cov3D_precomp=sc.cov3D)["num_rendered"]

    # informational: the same step with the sync-free forward (list buffer sized 1.25·N up front) captured in ONE
    # HIP graph and replayed — no host sync, no per-kernel launch cost.  Not the headline value: the default,
    # drop-in mode above is.
    graph_rec = None
    if world == 1 and not args.no_graph:
        try:
            from ggrt_official_amd.rasterizer import last_forward_status
            cap = int(N * 1.25) + 4096
            rast_g = GaussianRasterizer(rs._replace(list_capacity=cap))

            def step_g():
                for t in leaves:
                    t.grad = None
                color, _, _ = rast_g(means3D=means, means2D=means2D, opacities=op, shs=shs, cov3D_precomp=cov)
                color.backward(dL)

            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    step_g()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step_g()
            for _ in range(args.warmup):
                graph.replay()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize(dev)
            g_ms = (time.perf_counter() - t0) / args.steps * 1e3
            n_g, overflow = last_forward_status()
            graph_rec = {"ms_per_step": round(g_ms, 4), "mpix_s": round(W * H / g_ms / 1e3, 1), "list_capacity": cap,
                         "num_rendered": n_g, "overflow": overflow,
                         "note": "sync-free forward + backward captured in one HIP graph, replayed; informational"}
            log(f"hip graph replay: {g_ms:.3f} ms/step")
        except Exception as e:  # never let the informational leg break the contract line
            log(f"hip graph leg skipped: {type(e).__name__}: {e}")

    if rank == 0:
        P = cfg["num_points"]
        D = cfg["sh_degree"]
        K = (min(D, 3) + 1) ** 2
        M = sc.shs.shape[1]
        ab = algorithmic_bytes(P, N, W, H, K, M)
        kernel_ms = {"fwd_preprocess": stages["fwd_preprocess_ms"], "fwd_blend": stages["fwd_blend_ms"],
                     "bwd_blend": stages["bwd_blend_ms"], "bwd_preprocess": stages["bwd_preprocess_ms"]}
        dom = max(kernel_ms, key=kernel_ms.get)
        achieved = ab[dom] / (kernel_ms[dom] * 1e-3) / 1e9
        t_fwd = sum(v for k, v in stages.items() if k.startswith("fwd_"))
        t_bwd = sum(v for k, v in stages.items() if k.startswith("bwd_"))
        b_fwd = ab["fwd_preprocess"] + ab["fwd_binning"] + ab["fwd_blend"]
        b_bwd = ab["bwd_blend"] + ab["bwd_preprocess"]
        ms_per_step = elapsed / args.steps * 1e3
        # HBM bytes of the dominant kernel from the rocprofv3 PMC passes of this same command (FETCH_SIZE and
        # WRITE_SIZE in separate passes, FETCH ×2 per the gfx950 correction — scripts/profile_bench.sh,
        # scripts/pmc_summary.py); only valid for the configuration it was collected on (C3)
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_v16_pmc_traffic.json")
        if args.config == "C3" and os.path.exists(pmc_path):
            kname = {"bwd_blend": "blend_bwd_kernel", "fwd_blend": "blend_fwd_kernel",
                     "fwd_preprocess": "preprocess_fwd_kernel", "bwd_preprocess": "preprocess_bwd_kernel"}[dom]
            for k, v in json.load(open(pmc_path)).items():
                if kname in k:
                    traffic = int(v["hbm_bytes_per_launch"])
        # measured VALU issue occupancy of the two blend kernels (SQ_ACTIVE_INST_VALU ÷ SIMD quad-cycles of the
        # launch, rocprofv3 PMC passes of this command at C3 — profiles/r01_v16_pmc_sq.json): the bound that matters
        valu_busy = {}
        sq_path = os.path.join(ROOT, "profiles", "r01_v16_pmc_sq.json")
        if args.config == "C3" and os.path.exists(sq_path):
            for k, v in json.load(open(sq_path)).items():
                for short, kn in (("fwd", "blend_fwd_kernel"), ("bwd", "blend_bwd_kernel")):
                    if kn in k:
                        valu_busy[short] = v["valu_busy_frac_at_2p4GHz"]
        rec = {
            "metric": "Gaussian raster fwd+bwd Mpix/s @1M Gaussians 1080p",
            "value": round(world * W * H * args.steps / elapsed / 1e6, 3),
            "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, {W}x{H}, SH deg {D}, profile {cfg['profile']}, "
                                   f"fwd+bwd incl. camera gradient, 1 frame per GPU" +
                                   (", RCCL all-reduce of the camera gradient" if world > 1 else "") +
                                   (f" + of {args.grad_buffer_floats} stand-in fp32 grads"
                                    if world > 1 and args.grad_buffer_floats else ""),
                       "num_rendered": N, "parallelism": f"frames x{world}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes": ab[dom], "kernel_ms": round(kernel_ms[dom], 4),
                         "note": "blend kernels are fp32-VALU/exp-bound (≈160 flop per list-entry byte), not "
                                 "HBM-bound; see DESIGN.md §4"},
            "render_forward": {"ms": round(t_fwd, 4), "algorithmic_bytes": b_fwd,
                               "hbm_frac": round(b_fwd / (t_fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "render_backward": {"ms": round(t_bwd, 4), "algorithmic_bytes": b_bwd,
                                "hbm_frac": round(b_bwd / (t_bwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            # informational: the blend kernels against the fp32 VECTOR roofline (they are VALU-, not HBM-bound).
            # Algorithmic flops are SURVEY.md §8(d)'s: 20 flop per (list entry, pixel of its tile) forward, 2.5x
            # that backward.  The exact quadrant cull skips most of those evaluations, so the "algorithmic"
            # rate may exceed the 157.3 TFLOP/s peak — that excess is the cull, not a faster ALU.
            "blend_valu_roofline": {
                "peak_tflops": 157.3, "valu_busy_frac_pmc": valu_busy or None,
                "fwd": {"algorithmic_flops": 20 * N * 256, "tflops": round(20 * N * 256 / (kernel_ms["fwd_blend"] * 1e-3) / 1e12, 1)},
                "bwd": {"algorithmic_flops": 50 * N * 256, "tflops": round(50 * N * 256 / (kernel_ms["bwd_blend"] * 1e-3) / 1e12, 1)},
            },
        }
        if graph_rec is not None:
            rec["hipgraph_replay"] = graph_rec
        if world == 1 and not args.no_callsite:
            # informational: GGRt's own shape through the call-site layer (scripts/callsite_bench.py)
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import callsite_bench
                rec["callsite_ggrt_shape"] = callsite_bench.measure(str(dev), steps=10, warmup=3)
                log(f"call site at GGRt's shape: {rec['callsite_ggrt_shape']}")
            except Exception as e:
                log(f"call-site leg skipped: {type(e).__name__}: {e}")
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(cfg, seed=0)
        print(json.dumps(rec), flush=True)
    parallel.shutdown()


if __name__ == "__main__":
    main()
