"""The library keeps no state between calls and may be called from any thread (autograd runs
backward on a worker thread; a trainer may render on several streams): concurrent forward + backward from
several Python threads, each on its own HIP stream, give the single-threaded results."""
import threading

import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
dev = "cuda:0"


def _fwd_bwd(s, dL):
    from ggrt_official_amd import GaussianRasterizer
    leaves = [t.clone().requires_grad_() for t in (s.means3D, s.shs, s.opacities, s.cov3D)]
    m, sh, op, cov = leaves
    color, radii, depth = GaussianRasterizer(s.settings())(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sh,
                                                           cov3D_precomp=cov)
    (color * dL).sum().backward()
    return color.detach(), radii, [t.grad for t in leaves]


def test_concurrent_calls_from_threads_on_separate_streams():
    scenes = [make_scene(6000 + 500 * i, 160 + 16 * i, 120, sh_degree=i % 4, seed=20 + i).to(dev) for i in range(4)]
    grads_in = [upstream_gradient(s.width, s.height, seed=30 + i, device=dev) for i, s in enumerate(scenes)]
    ref = [_fwd_bwd(s, g) for s, g in zip(scenes, grads_in)]
    torch.cuda.synchronize()
    out, errors = [None] * 4, []

    def work(i):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(6):                       # several rounds to give the threads time to interleave
                    out[i] = _fwd_bwd(scenes[i], grads_in[i])
            stream.synchronize()
        except Exception as e:                           # pragma: no cover
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (c0, r0, g0), (c1, r1, g1) in zip(ref, out):
        assert torch.equal(c0, c1) and torch.equal(r0, r1)
        for a, b in zip(g0, g1):
            assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5


def test_two_backwards_over_one_forward():
    """The forward clears the backward's scratch on the side; a SECOND backward over the same forward (retain_graph)
    must clear a scratch of its own and give the same gradients."""
    import numpy as np
    from ggrt_official_amd import GaussianRasterizer
    from ggrt_official_amd.synthetic import make_scene, upstream_gradient
    sc = make_scene(4000, 96, 64, sh_degree=2, seed=12).to("cuda:0")
    m = sc.means3D.clone().requires_grad_(True)
    op = sc.opacities.clone().requires_grad_(True)
    color, _, _ = GaussianRasterizer(sc.settings())(means3D=m, means2D=torch.zeros_like(m), opacities=op, shs=sc.shs,
                                                    cov3D_precomp=sc.cov3D)
    dL = upstream_gradient(96, 64, device="cuda:0")
    g1 = torch.autograd.grad(color, (m, op), dL, retain_graph=True)
    g2 = torch.autograd.grad(color, (m, op), dL, retain_graph=True)
    g3 = torch.autograd.grad(color, (m, op), 2.0 * dL)
    for a, b, c in zip(g1, g2, g3):
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
        assert rel_l2(c.cpu().numpy(), 2.0 * a.cpu().numpy()) < 1e-5


def test_short_lived_threads_reuse_the_host_slots():
    """A host thread owns a read-back slot (pinned line + events) and, with the global depth sort, a side-stream slot per
    device; when it ends they go back to a process-wide pool (api.hip SlotPool): twelve threads that render one
    after the other must not allocate twelve of each — and what they render is what the main thread renders."""
    import ctypes as C
    from ggrt_official_amd import _lib
    lib = _lib.load()

    def slots():
        a, b = C.c_int32(0), C.c_int32(0)
        assert lib.ggr_debug_host_slots(C.byref(a), C.byref(b)) == 0
        return a.value, b.value

    sc = make_scene(8000, 160, 120, sh_degree=3, seed=5).to(dev)
    dL = upstream_gradient(sc.width, sc.height, seed=6, device=dev)
    import os
    old = os.environ.get("GGR_DEPTH_SORT")
    os.environ["GGR_DEPTH_SORT"] = "global"     # the form that uses the side stream
    try:
        ref = _fwd_bwd(sc, dL)
        torch.cuda.synchronize()
        before = slots()
        out, errors = [], []

        def work():
            try:
                out.append(_fwd_bwd(sc, dL))
                torch.cuda.synchronize()
            except Exception as e:                       # pragma: no cover
                errors.append(repr(e))

        for _ in range(12):
            t = threading.Thread(target=work)
            t.start()
            t.join()
        assert not errors, errors
        after = slots()
    finally:
        if old is None:
            os.environ.pop("GGR_DEPTH_SORT", None)
        else:
            os.environ["GGR_DEPTH_SORT"] = old
    # (the forward runs on the thread itself, the backward on autograd's worker thread, which lives on: at most a slot or
    #  two beyond the main thread's, never one per thread)
    assert after[0] - before[0] <= 2 and after[1] - before[1] <= 2, (before, after)
    for c1, r1, g1 in out:
        assert torch.equal(ref[0], c1) and torch.equal(ref[1], r1)
        for a, b in zip(ref[2], g1):
            assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
