"""V views of the same Gaussians in ONE launch set (``rasterize_views`` → ``ggr_forward_views`` /
``ggr_backward_views``) against the reference's way of doing it — one rasterizer call per view (reference
``cuda_splatting.py:93-127``), gradients added by autograd — and against the C oracle (`-m gpu`).

Per view the batched path must build the SAME tile lists, hence bit-identical images and radii; the gradients w.r.t.
the Gaussians are sums over the views formed inside the backward kernel instead of by autograd, so they agree to
summation order."""
import math

import numpy as np
import pytest
import torch

from ggrt_official_amd import GaussianRasterizer, rasterize_views
from ggrt_official_amd.synthetic import camera_matrices, make_scene, upstream_gradient
from oracle import c_oracle
from tests.helpers import check_grads, check_image, rel_l2

pytestmark = pytest.mark.gpu
dev = "cuda:0"


def _pose(k):
    a, b = 0.12 * math.sin(1.7 * k + 0.3), 0.1 * math.cos(2.3 * k)
    Ry = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=torch.float64)
    Rx = torch.tensor([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]], dtype=torch.float64)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, :3] = Ry @ Rx
    c2w[:3, 3] = torch.tensor([0.2 * math.sin(k), -0.15 * math.cos(2 * k), 0.1 * k])
    return c2w


def _cameras(W, H, V):
    cams = [camera_matrices(W, H, c2w=_pose(k)) for k in range(V)]
    view = torch.stack([c[0] for c in cams])
    full = torch.stack([c[1] for c in cams])
    campos = torch.stack([c[2] for c in cams])
    tanfov = torch.tensor([[c[3], c[4]] for c in cams], dtype=torch.float32)
    return view, full, campos, tanfov


def _run(sc, cams, dLs, dDs, mode, use_sh, use_cov, bgs, scales, aux, batched, pose=False, cap=3):
    view, full, campos, tanfov = [t.to(dev) for t in cams]
    V = view.shape[0]
    leaf = lambda t: t.detach().clone().to(dev).requires_grad_(True)
    means, op = leaf(sc.means3D), leaf(sc.opacities)
    leaves = dict(means3D=means, opacities=op)
    kw = {}
    if use_sh:
        leaves["shs"] = kw["shs"] = leaf(sc.shs)
    else:
        leaves["colors_precomp"] = kw["colors_precomp"] = leaf(sc.shs[:, 0].abs())
    if use_cov:
        leaves["cov3D_precomp"] = kw["cov3D_precomp"] = leaf(sc.cov3D)
    else:
        leaves["scales"] = kw["scales"] = leaf(sc.scales)
        leaves["rotations"] = kw["rotations"] = leaf(sc.rotations)
    if pose:
        view, full, campos = leaf(view), leaf(full), leaf(campos)
        leaves.update(viewmatrix=view, projmatrix=full, campos=campos)
    aux_l = None if aux is None else leaf(aux)
    if aux_l is not None:
        leaves["aux"] = aux_l
    rs = sc.to(dev).settings()._replace(aux_affine=mode, sh_max_degree=cap)
    sink = torch.zeros(V, sc.means3D.shape[0], 3, device=dev, requires_grad=True)
    leaves["means2D"] = sink
    if batched:
        color, radii, depth = rasterize_views(means, op, view, full, campos, bgs.to(dev), tanfov, rs, aux_precomp=aux_l,
                                              input_scale=None if scales is None else scales.to(dev), means2D=sink, **kw)
    else:
        cs, rr, ds = [], [], []
        for v in range(V):
            r = rs._replace(viewmatrix=view[v], projmatrix=full[v], campos=campos[v], bg=bgs[v].to(dev),
                            tanfovx=float(tanfov[v, 0]), tanfovy=float(tanfov[v, 1]),
                            input_scale=None if scales is None else scales[v:v + 1].to(dev))
            c, r_, d = GaussianRasterizer(r)(means3D=means, means2D=sink[v], opacities=op,
                                             aux_precomp=None if aux_l is None else aux_l[v], **kw)
            cs.append(c); rr.append(r_); ds.append(d)
        color, radii, depth = torch.stack(cs), torch.stack(rr), torch.stack(ds)
    loss = (color * dLs.to(dev)).sum()
    if dDs is not None:
        loss = loss + (depth * dDs.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: (None if t.grad is None else t.grad.detach().cpu().numpy()) for k, t in leaves.items()}
    return color.detach().cpu().numpy(), radii.cpu().numpy(), depth.detach().cpu().numpy(), grads


@pytest.mark.parametrize("case", [
    dict(P=6000, W=112, H=80, D=3, V=3, use_sh=True, use_cov=True),
    dict(P=9000, W=130, H=70, D=4, V=4, use_sh=True, use_cov=False, scaled=True, depth=True),   # ragged size, M = 25
    dict(P=3000, W=64, H=48, D=0, V=2, use_sh=False, use_cov=True, aux=True, depth=True),
    dict(P=20000, W=160, H=112, D=4, V=5, use_sh=True, use_cov=True, profile="B", affine=True, depth=True, pose=True),
    dict(P=7000, W=112, H=80, D=4, V=3, use_sh=True, use_cov=True, pose=True, cap=4),   # band 4 evaluated: K = 25 in the views' SH kernel
])
def test_views_equal_per_view_calls(case):
    P, W, H, V = case["P"], case["W"], case["H"], case["V"]
    sc = make_scene(P, W, H, sh_degree=case["D"], profile=case.get("profile", "A"), seed=P % 17)
    cams = _cameras(W, H, V)
    g = torch.Generator().manual_seed(5)
    dLs = torch.stack([upstream_gradient(W, H, seed=10 + v) for v in range(V)])
    dDs = torch.stack([upstream_gradient(W, H, seed=40 + v)[0] * 0.2 for v in range(V)]) if case.get("depth") else None
    bgs = torch.rand(V, 3, generator=g)
    scales = (0.5 + torch.rand(V, generator=g)) if case.get("scaled") else None
    aux = torch.rand(V, P, generator=g) if case.get("aux") else None
    mode = (0.5, 0.28209479177387814) if case.get("affine") else None
    args = (sc, cams, dLs, dDs, mode, case["use_sh"], case["use_cov"], bgs, scales, aux)
    ca, ra, da, ga = _run(*args, batched=False, pose=case.get("pose", False), cap=case.get("cap", 3))
    cb, rb, db, gb = _run(*args, batched=True, pose=case.get("pose", False), cap=case.get("cap", 3))
    assert np.array_equal(ra, rb)
    assert np.array_equal(ca, cb), float(np.abs(ca - cb).max())      # same lists, same blend: bit-identical
    assert np.array_equal(da, db)
    keys = [k for k in ga if ga[k] is not None]
    assert set(keys) == set(k for k in gb if gb[k] is not None)
    check_grads(gb, ga, keys, tag="views")


def test_views_against_the_oracle():
    P, W, H, V = 8000, 128, 96, 3
    sc = make_scene(P, W, H, sh_degree=3, seed=3)
    cams = _cameras(W, H, V)
    dLs = torch.stack([upstream_gradient(W, H, seed=20 + v) for v in range(V)])
    bgs = torch.tensor([[0.1, 0.2, 0.3], [0.0, 0.0, 0.0], [0.9, 0.5, 0.1]])
    color, radii, depth, grads = _run(sc, cams, dLs, None, None, True, True, bgs, None, None, batched=True)
    n = lambda t: t.detach().cpu().numpy()
    total = None
    for v in range(V):
        st = c_oracle.forward(n(sc.means3D), n(sc.opacities), n(cams[0][v]), n(cams[1][v]), n(cams[2][v]), n(bgs[v]),
                              W, H, float(cams[3][v, 0]), float(cams[3][v, 1]), sh_degree=3, shs=n(sc.shs),
                              cov3D_precomp=n(sc.cov3D))
        assert np.array_equal(radii[v], st.radii)
        check_image(color[v], st.color, tag=f"views_oracle:{v}")
        ref = c_oracle.backward(st, n(dLs[v]))
        assert rel_l2(grads["means2D"][v], ref["means2D"]) < 2e-5
        total = ref if total is None else {k: (None if ref[k] is None else total[k] + ref[k]) for k in total}
    check_grads(grads, total, ["means3D", "shs", "opacities", "cov3D_precomp"], tag="views_oracle")


def test_decoder_batched_equals_per_view_loop():
    from ggrt_official_amd import splatting as sp
    b, v, gc, h, w = 2, 3, 5000, 80, 112
    g = torch.Generator().manual_seed(1)
    sc = [make_scene(gc, w, h, sh_degree=4, seed=30 + i) for i in range(b)]
    cov = torch.zeros(b, gc, 3, 3)
    for i in range(b):
        for k, (r, c) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
            cov[i, :, r, c] = sc[i].cov3D[:, k]
            cov[i, :, c, r] = sc[i].cov3D[:, k]
    ext = torch.stack([torch.stack([_pose(3 * i + k).float() for k in range(v)]) for i in range(b)])
    fx = 0.5 / sc[0].tanfovx
    intr = torch.tensor([[fx, 0, 0.5], [0, fx * w / h, 0.5], [0, 0, 1]]).expand(b, v, 3, 3).contiguous()
    near, far = torch.full((b, v), 0.8), torch.full((b, v), 90.0)
    wc = torch.randn(b, v, 3, h, w, generator=g)
    wd = torch.randn(b, v, h, w, generator=g) * 0.1
    outs = []
    for batched in (False, True):
        leaves = [t.clone().to(dev).requires_grad_(True) for t in (
            torch.stack([s.means3D for s in sc]), cov, torch.stack([s.shs.permute(0, 2, 1).contiguous() for s in sc]),
            torch.stack([s.opacities[:, 0] for s in sc]))]
        gs = sp.Gaussians(*leaves)
        bg = torch.zeros(b * v, 3, device=dev)
        color, depth = sp.render_views_fused(ext.flatten(0, 1).to(dev), intr.flatten(0, 1).to(dev), near.flatten().to(dev),
                                             far.flatten().to(dev), (h, w), bg, gs, [n // v for n in range(b * v)],
                                             "depth", batched=batched)
        ((color.reshape(b, v, 3, h, w) * wc.to(dev)).sum() + (depth.reshape(b, v, h, w) * wd.to(dev)).sum()).backward()
        outs.append((color.detach().cpu().numpy(), depth.detach().cpu().numpy(), [t.grad.cpu().numpy() for t in leaves]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    for a, bb in zip(outs[0][2], outs[1][2]):
        assert rel_l2(bb, a) < 2e-5


def test_views_sync_free_matches_exact_mode():
    """`list_capacity` (no read-back, graph-capturable) covers the lists of ALL views of the launch set."""
    from ggrt_official_amd import last_forward_status
    P, W, H, V = 7000, 112, 80, 3
    sc = make_scene(P, W, H, sh_degree=2, seed=8).to(dev)
    view, full, campos, tanfov = [t.to(dev) for t in _cameras(W, H, V)]
    bgs = torch.zeros(V, 3, device=dev)
    outs = []
    for cap in (0, 400_000, 2_000):   # exact, roomy, overflowing
        rs = sc.settings()._replace(list_capacity=cap)
        m = sc.means3D.clone().requires_grad_(True)
        color, radii, depth = rasterize_views(m, sc.opacities, view, full, campos, bgs, tanfov, rs, shs=sc.shs,
                                              cov3D_precomp=sc.cov3D)
        color.sum().backward()
        n, overflow = last_forward_status()
        outs.append((color.detach().clone(), m.grad.clone(), n, overflow))
    assert outs[0][2] == outs[1][2] > 0 and not outs[1][3]
    assert torch.equal(outs[0][0], outs[1][0])
    assert rel_l2(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy()) < 1e-5
    assert outs[2][3] and outs[2][2] == outs[0][2]          # overflow raised, true count still reported
    assert torch.isfinite(outs[2][0]).all()


def test_more_views_than_sort_segments():
    """Beyond 64 views the depth sort falls back to ONE segment over all (view, Gaussian) keys: same results."""
    P, W, H, V = 300, 48, 32, 70
    sc = make_scene(P, W, H, sh_degree=1, seed=2).to(dev)
    view, full, campos, tanfov = [t.to(dev) for t in _cameras(W, H, V)]
    bgs = torch.rand(V, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        color, radii, depth = rasterize_views(sc.means3D, sc.opacities, view, full, campos, bgs, tanfov, sc.settings(),
                                              shs=sc.shs, cov3D_precomp=sc.cov3D)
        for v in (0, 1, 33, 69):
            r = sc.settings()._replace(viewmatrix=view[v], projmatrix=full[v], campos=campos[v], bg=bgs[v],
                                       tanfovx=float(tanfov[v, 0]), tanfovy=float(tanfov[v, 1]))
            c1, r1, d1 = GaussianRasterizer(r)(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D),
                                               opacities=sc.opacities, shs=sc.shs, cov3D_precomp=sc.cov3D)
            assert torch.equal(color[v], c1) and torch.equal(radii[v], r1) and torch.equal(depth[v], d1)


@pytest.mark.parametrize("case", [
    dict(B=3, vps=2, P=5000, W=112, H=80, D=3, use_sh=True, use_cov=True),                         # MULTI kernels per set
    dict(B=4, vps=1, P=4097, W=96, H=64, D=4, use_sh=True, use_cov=True, cm=True, cap=4),          # one view per set: the
    dict(B=2, vps=1, P=6000, W=130, H=70, D=4, use_sh=True, use_cov=False, depth=True, pose=True),  # one-view kernels; ragged P
    dict(B=2, vps=3, P=3000, W=64, H=48, D=0, use_sh=False, use_cov=True, aux=True, depth=True, scaled=True),
])
def test_gaussian_sets_equal_per_set_calls(case):
    """GgrViews.num_sets (VERDICT r2 missing #3): B DIFFERENT Gaussian sets, V/B views each, in ONE launch set — the
    reference's `(b v)` flattening (decoder_splatting_cuda.py:40-60) — against one `rasterize_views` call per set:
    same tile lists per view, hence bit-identical images / radii; gradients come back per set, to summation order."""
    B, vps, P, W, H = case["B"], case["vps"], case["P"], case["W"], case["H"]
    V = B * vps
    scs = [make_scene(P, W, H, sh_degree=case["D"], profile="AB"[b % 2], seed=50 + b) for b in range(B)]
    view, full, campos, tanfov = [t.to(dev) for t in _cameras(W, H, V)]
    g = torch.Generator().manual_seed(9)
    dLs = torch.stack([upstream_gradient(W, H, seed=60 + v) for v in range(V)]).to(dev)
    dDs = torch.stack([upstream_gradient(W, H, seed=80 + v)[0] * 0.2 for v in range(V)]).to(dev) if case.get("depth") else None
    bgs = torch.rand(V, 3, generator=g).to(dev)
    scales_in = (0.5 + torch.rand(V, generator=g)).to(dev) if case.get("scaled") else None
    aux = torch.rand(V, P, generator=g).to(dev) if case.get("aux") else None
    cm = case.get("cm", False)
    stack = lambda f: torch.stack([f(s) for s in scs]).to(dev)
    inputs = dict(means3D=stack(lambda s: s.means3D), opacities=stack(lambda s: s.opacities))
    if case["use_sh"]:
        inputs["shs"] = stack(lambda s: s.shs.permute(0, 2, 1).contiguous() if cm else s.shs)
    else:
        inputs["colors_precomp"] = stack(lambda s: s.shs[:, 0].abs())
    if case["use_cov"]:
        inputs["cov3D_precomp"] = stack(lambda s: s.cov3D)
    else:
        inputs["scales"], inputs["rotations"] = stack(lambda s: s.scales), stack(lambda s: s.rotations)
    rs = scs[0].to(dev).settings()._replace(sh_channel_major=cm, sh_max_degree=case.get("cap", 3))

    def run(as_sets):
        leaf = lambda t: t.detach().clone().requires_grad_(True)
        lv = {k: leaf(t) for k, t in inputs.items()}
        cams = [leaf(view), leaf(full), leaf(campos)] if case.get("pose") else [view, full, campos]
        aux_l = None if aux is None else leaf(aux)
        sink = torch.zeros(V, P, 3, device=dev, requires_grad=True)
        m, o = lv.pop("means3D"), lv.pop("opacities")
        if as_sets:
            color, radii, depth = rasterize_views(m, o, *cams, bgs, tanfov.to(dev), rs, aux_precomp=aux_l,
                                                  input_scale=scales_in, means2D=sink, **lv)
        else:
            outs = []
            for b in range(B):
                sl = slice(b * vps, (b + 1) * vps)
                outs.append(rasterize_views(m[b], o[b], cams[0][sl], cams[1][sl], cams[2][sl], bgs[sl], tanfov.to(dev)[sl],
                                            rs, aux_precomp=None if aux_l is None else aux_l[sl],
                                            input_scale=None if scales_in is None else scales_in[sl], means2D=sink[sl],
                                            **{k: t[b] for k, t in lv.items()}))
            color, radii, depth = (torch.cat([o_[i] for o_ in outs]) for i in range(3))
        loss = (color * dLs).sum() + (0 if dDs is None else (depth * dDs).sum())
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: t.grad.detach().cpu().numpy() for k, t in lv.items()}
        grads.update(means3D=m.grad.cpu().numpy(), opacities=o.grad.cpu().numpy(), means2D=sink.grad.cpu().numpy())
        if case.get("pose"):
            grads.update(viewmatrix=cams[0].grad.cpu().numpy(), projmatrix=cams[1].grad.cpu().numpy(), campos=cams[2].grad.cpu().numpy())
        if aux_l is not None:
            grads["aux"] = aux_l.grad.cpu().numpy()
        return color.detach().cpu().numpy(), radii.cpu().numpy(), depth.detach().cpu().numpy(), grads

    ca, ra, da, ga = run(False)
    cb, rb, db, gb = run(True)
    assert np.array_equal(ra, rb)
    assert np.array_equal(ca, cb), float(np.abs(ca - cb).max())
    assert np.array_equal(da, db)
    assert set(ga) == set(gb)
    for k in ga:
        assert ga[k].shape == gb[k].shape, k
        flat = lambda t: t.reshape(-1, *t.shape[2:]) if k not in ("viewmatrix", "projmatrix", "campos") else t
        assert rel_l2(flat(gb[k]), flat(ga[k])) < 2e-6, (k, rel_l2(gb[k], ga[k]))
    # set 0 alone against the C oracle (view 0)
    n = lambda t: t.detach().cpu().numpy()
    if case["use_sh"] and case["use_cov"] and not cm and case.get("cap", 3) == 3:
        st = c_oracle.forward(n(scs[0].means3D), n(scs[0].opacities), n(view[0]), n(full[0]), n(campos[0]), n(bgs[0]),
                              W, H, float(tanfov[0, 0]), float(tanfov[0, 1]), sh_degree=case["D"], shs=n(scs[0].shs),
                              cov3D_precomp=n(scs[0].cov3D))
        assert np.array_equal(rb[0], st.radii)
        check_image(cb[0], st.color, tag="sets_oracle")


def test_decoder_all_batch_elements_in_one_launch_set():
    """`render_views_fused` with b > 1 batch elements takes ONE `rasterize_views` call over all of them (no Python loop
    over batch elements): same images and gradients as the per-element path."""
    from ggrt_official_amd import splatting as sp
    b, v, gc, h, w = 3, 2, 4000, 80, 112
    sc = [make_scene(gc, w, h, sh_degree=4, seed=70 + i) for i in range(b)]
    ext = torch.stack([torch.stack([_pose(3 * i + k).float() for k in range(v)]) for i in range(b)]).to(dev)
    intr = torch.eye(3).repeat(b, v, 1, 1)
    intr[..., 0, 0], intr[..., 1, 1], intr[..., 0, 2], intr[..., 1, 2] = 1.1, 1.1 * w / h, 0.5, 0.5
    near, far = torch.full((b, v), 0.9), torch.full((b, v), 80.0)

    def run(batched):
        leaf = lambda t: t.detach().clone().to(dev).requires_grad_(True)
        gs = sp.Gaussians(means=leaf(torch.stack([s.means3D for s in sc])),
                          covariances=leaf(torch.stack([torch.stack([s.cov3D[:, [0, 1, 2, 1, 3, 4, 2, 4, 5]].reshape(-1, 3, 3)]).squeeze(0) for s in sc])),
                          harmonics=leaf(torch.stack([s.shs.permute(0, 2, 1) for s in sc])),
                          opacities=leaf(torch.stack([s.opacities for s in sc])))
        col, dep = sp.render_views_fused(ext.flatten(0, 1), intr.flatten(0, 1).to(dev), near.flatten().to(dev),
                                         far.flatten().to(dev), (h, w), torch.zeros(b * v, 3, device=dev), gs,
                                         [n // v for n in range(b * v)], "depth", batched=batched)
        (col.sum() + 0.3 * dep.sum()).backward()
        torch.cuda.synchronize()
        return col.detach(), dep.detach(), [t.grad for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities)]

    c1, d1, g1 = run(True)
    c0, d0, g0 = run(False)
    assert torch.equal(c1, c0) and torch.equal(d1, d0)
    for a, r in zip(g1, g0):
        assert rel_l2(a.cpu().numpy(), r.cpu().numpy()) < 5e-6
