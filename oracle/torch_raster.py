"""PyTorch-CPU restatement of the tile rasterizer — TEST INFRASTRUCTURE ONLY.

Second, independently formulated oracle (vectorised per tile, gradients from
``torch.autograd``) for the rasterizer GGRt calls at
``ggrt/model/pixelsplat/decoder/cuda_splatting.py:101-125`` of the reference.  It is also the
"PyTorch sort+blend CPU fallback" BASELINE.json asks to be timed next to the GPU path (the
reference itself has no CPU fallback: ``decoder/__init__.py:4-6`` registers only
``splatting_cuda``).

PARITY UNPINNED: the rasterizer arithmetic is an un-vendored third-party CUDA extension
(reference ``README.md:17-18``); this file follows SURVEY.md Appendix A.  Where upstream's
hand-written backward deviates from naive autograd (Appendix A.5) the deviation is emulated
with ``detach`` so that autograd reproduces upstream's gradient:
  * ``min(0.99, α)`` is straight-through,
  * a frustum-clamped ``t.x``/``t.y`` is a constant,
  * depth / radius / tile rect / ``+0.3`` dilation carry no gradient.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this; the product package never does.
"""
from __future__ import annotations

import math

import torch

TILE = 16
NEAR_CULL = 0.2
DILATION = 0.3
FRUSTUM_CLAMP = 1.3
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
T_MIN = 1e-4

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)
SH_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
         -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761)
# Highest SH band evaluated.  graphdeco's rasterizer and its w-depth forks (the family GGRt's live call site's
# signature belongs to, see oracle/ggr_oracle.c) stop at 3; band 4 on request (sh_cap=4).
SH_CAP = 3


def sh_eff_degree(D: int, M: int, cap: int = SH_CAP) -> int:
    deg = max(min(D, cap), 0)
    while deg > 0 and (deg + 1) ** 2 > M:
        deg -= 1
    return deg


def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    """[P,3] unit directions -> [P,K] real SH basis in the rasterizer's sign convention."""
    x, y, z = d.unbind(-1)
    B = [torch.full_like(x, SH_C0)]
    if deg > 0:
        B += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        B += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        B += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
              SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
              SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    if deg > 3:
        B += [SH_C4[0] * xy * (xx - yy), SH_C4[1] * yz * (3 * xx - yy), SH_C4[2] * xy * (7 * zz - 1),
              SH_C4[3] * yz * (7 * zz - 3), SH_C4[4] * (zz * (35 * zz - 30) + 3), SH_C4[5] * xz * (7 * zz - 3),
              SH_C4[6] * (xx - yy) * (7 * zz - 1), SH_C4[7] * xz * (xx - 3 * yy),
              SH_C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(B, -1)


def cov3d_from_scale_rot(scales, rotations, mod=1.0):
    r, x, y, z = rotations.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Mx = R * (mod * scales)[:, None, :]
    S = Mx @ Mx.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


def preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree=0,
               shs=None, colors_precomp=None, cov3D_precomp=None, scales=None, rotations=None,
               scale_modifier=1.0, depth_grad=False, sh_cap=None):
    """``depth_grad``: keep view-space z differentiable as the blended depth FEATURE (the "w-depth"
    forks' out_depth gradient); its use as a sort key never carries gradient."""
    dt = means3D.dtype
    P = means3D.shape[0]
    V, PM = viewmatrix.to(dt), projmatrix.to(dt)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    cov6 = cov3D_precomp if cov3D_precomp is not None else cov3d_from_scale_rot(scales, rotations, scale_modifier)
    hom = torch.cat([means3D, torch.ones_like(means3D[:, :1])], -1)
    t = (hom @ V)[:, :3]
    ph = hom @ PM
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ppx, ppy = ph[:, 0] * pw, ph[:, 1] * pw
    tz = t[:, 2]
    in_front = tz > NEAR_CULL
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = FRUSTUM_CLAMP * tanfovx, FRUSTUM_CLAMP * tanfovy
    txtz, tytz = t[:, 0] / tz_safe, t[:, 1] / tz_safe
    xcl = (txtz < -limx) | (txtz > limx)
    ycl = (tytz < -limy) | (tytz > limy)
    tx = torch.where(xcl, (txtz.clamp(-limx, limx) * tz_safe).detach(), txtz * tz_safe)
    ty = torch.where(ycl, (tytz.clamp(-limy, limy) * tz_safe).detach(), tytz * tz_safe)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], -1).reshape(P, 2, 3)
    R = V[:3, :3].T  # world->view rotation
    A = J @ R
    S = torch.stack([cov6[:, 0], cov6[:, 1], cov6[:, 2], cov6[:, 1], cov6[:, 3], cov6[:, 4],
                     cov6[:, 2], cov6[:, 4], cov6[:, 5]], -1).reshape(P, 3, 3)
    c2 = A @ S @ A.transpose(1, 2)
    a, b, c = c2[:, 0, 0] + DILATION, c2[:, 0, 1], c2[:, 1, 1] + DILATION
    det = a * c - b * b
    det_ok = det != 0
    det_s = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], -1)
    with torch.no_grad():
        mid = 0.5 * (a + c)
        sq = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        lam = torch.maximum(mid + sq, mid - sq)
        radius = torch.ceil(3.0 * torch.sqrt(lam)).to(torch.int64)
    px = ((ppx + 1.0) * W - 1.0) * 0.5
    py = ((ppy + 1.0) * H - 1.0) * 0.5
    with torch.no_grad():
        rf = radius.to(dt)
        rminx = torch.trunc((px - rf) / TILE).to(torch.int64).clamp(0, gx)
        rminy = torch.trunc((py - rf) / TILE).to(torch.int64).clamp(0, gy)
        rmaxx = torch.trunc((px + rf + (TILE - 1)) / TILE).to(torch.int64).clamp(0, gx)
        rmaxy = torch.trunc((py + rf + (TILE - 1)) / TILE).to(torch.int64).clamp(0, gy)
        area = (rmaxx - rminx) * (rmaxy - rminy)
        visible = in_front & det_ok & (area > 0)
    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        deg = sh_eff_degree(sh_degree, shs.shape[1], SH_CAP if sh_cap is None else sh_cap)
        K = (deg + 1) ** 2
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=-1, keepdim=True)
        Bm = sh_basis(deg, d)
        raw = (Bm[:, :, None] * shs[:, :K, :]).sum(1) + 0.5
        clamped = raw < 0
        rgb = raw.clamp(min=0.0)
    return dict(xy=torch.stack([px, py], -1), conic=conic, opacity=opacities.reshape(-1), rgb=rgb,
                depth=tz if depth_grad else tz.detach(), radii=torch.where(visible, radius, torch.zeros_like(radius)),
                rect=(rminx, rminy, rmaxx, rmaxy), visible=visible,
                tiles_touched=torch.where(visible, area, torch.zeros_like(area)), clamped=clamped, cov3D=cov6)


def bin_tiles(pre, W, H):
    """Emit (tile<<32 | depth_bits) keys, stable sort, tile ranges (Appendix A.2)."""
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    vis = pre["visible"].nonzero().squeeze(-1)
    rminx, rminy, rmaxx, rmaxy = [r[vis] for r in pre["rect"]]
    wdt = rmaxx - rminx
    cnt = wdt * (rmaxy - rminy)
    N = int(cnt.sum())
    owner = torch.repeat_interleave(torch.arange(vis.numel()), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(N) - start[owner]
    ty = rminy[owner] + local // wdt[owner]
    tx = rminx[owner] + local % wdt[owner]
    tile = ty * gx + tx
    dbits = pre["depth"][vis].detach().to(torch.float32).view(torch.int32).to(torch.int64)[owner]
    keys = (tile << 32) | dbits
    order = torch.sort(keys, stable=True).indices
    keys = keys[order]
    point_list = vis[owner][order]
    tiles_sorted = keys >> 32
    counts = torch.bincount(tiles_sorted, minlength=gx * gy)
    ends = torch.cumsum(counts, 0)
    ranges = torch.stack([ends - counts, ends], -1)
    ranges[counts == 0] = 0
    return point_list, ranges, keys, N


def blend(pre, point_list, ranges, bg, W, H, want_depth=True, tile_filter=None, aux=None):
    """Per-tile front-to-back compositing with the exact skip/stop rules (Appendix A.3),
    vectorised over the tile's pixels × its list."""
    dt = pre["xy"].dtype
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    color = torch.zeros(3, H, W, dtype=dt)
    depth_img = torch.zeros(H, W, dtype=dt)
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    bg = bg.to(dt)
    color = color + bg[:, None, None]  # tiles without entries: T=1 -> bg
    pieces = []
    for tyi in range(gy):
        y0, y1 = tyi * TILE, min(tyi * TILE + TILE, H)
        for txi in range(gx):
            r0, r1 = int(ranges[tyi * gx + txi, 0]), int(ranges[tyi * gx + txi, 1])
            if r1 <= r0 or (tile_filter is not None and not tile_filter(txi, tyi)):
                continue
            x0, x1 = txi * TILE, min(txi * TILE + TILE, W)
            ids = point_list[r0:r1]
            ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
            pixx, pixy = xs.reshape(-1), ys.reshape(-1)
            xy = pre["xy"][ids]
            con = pre["conic"][ids]
            op = pre["opacity"][ids]
            dx = xy[:, 0:1] - pixx[None]
            dy = xy[:, 1:2] - pixy[None]
            power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
            araw = op[:, None] * torch.exp(power)
            alpha = araw + (araw.clamp(max=ALPHA_MAX) - araw).detach()
            valid = (power <= 0) & (alpha >= ALPHA_MIN)
            aeff = torch.where(valid, alpha, torch.zeros_like(alpha))
            one_m = 1.0 - aeff
            Tafter = torch.cumprod(one_m, 0)
            Tbefore = torch.cat([torch.ones_like(Tafter[:1]), Tafter[:-1]], 0)
            with torch.no_grad():
                stop = (Tafter < T_MIN) & valid
                done = torch.cumsum(stop.to(torch.int32), 0) > 0
                live = valid & ~done
                idx = torch.arange(1, ids.numel() + 1)[:, None].expand_as(live)
                last = torch.where(live, idx, torch.zeros_like(idx)).max(0).values
            w = torch.where(live, aeff * Tbefore, torch.zeros_like(aeff))
            Tfin = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), 0)
            c = (w[:, :, None] * pre["rgb"][ids][:, None, :]).sum(0)  # [pix,3]
            c = c + Tfin[:, None] * bg[None]
            pieces.append((y0, y1, x0, x1, c, Tfin, last,
                           (w * (pre["depth"] if aux is None else aux)[ids].to(dt)[:, None]).sum(0) if want_depth else None))
    # assemble without in-place ops on a graph tensor
    if pieces:
        canvas = [[None] * gx for _ in range(gy)]
        dcanvas = [[None] * gx for _ in range(gy)]
        for (y0, y1, x0, x1, c, Tfin, last, dz) in pieces:
            h, w_ = y1 - y0, x1 - x0
            canvas[y0 // TILE][x0 // TILE] = c.T.reshape(3, h, w_)
            final_T[y0:y1, x0:x1] = Tfin.detach().reshape(h, w_)
            n_contrib[y0:y1, x0:x1] = last.reshape(h, w_)
            if dz is not None:
                dcanvas[y0 // TILE][x0 // TILE] = dz.reshape(h, w_)
        rows = []
        for tyi in range(gy):
            y0, y1 = tyi * TILE, min(tyi * TILE + TILE, H)
            row = []
            for txi in range(gx):
                x0, x1 = txi * TILE, min(txi * TILE + TILE, W)
                blk = canvas[tyi][txi]
                if blk is None:
                    blk = bg[:, None, None].expand(3, y1 - y0, x1 - x0)
                row.append(blk)
            rows.append(torch.cat(row, 2))
        color = torch.cat(rows, 1)
        if want_depth:  # differentiable assembly of the 4th (depth / aux) channel
            drows = []
            for tyi in range(gy):
                y0, y1 = tyi * TILE, min(tyi * TILE + TILE, H)
                drow = []
                for txi in range(gx):
                    x0, x1 = txi * TILE, min(txi * TILE + TILE, W)
                    blk = dcanvas[tyi][txi]
                    drow.append(blk if blk is not None else torch.zeros(y1 - y0, x1 - x0, dtype=dt))
                drows.append(torch.cat(drow, 1))
            depth_img = torch.cat(drows, 0)
    return color, final_T, n_contrib, depth_img


def rasterize(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0,
              shs=None, colors_precomp=None, cov3D_precomp=None, scales=None, rotations=None,
              scale_modifier=1.0, return_state=False, tile_filter=None, aux=None, depth_grad=False, sh_cap=None):
    """Full forward.  Returns (color[3,H,W], radii[P], depth[H,W]) like the boundary's 3-tuple.
    ``tile_filter(tx, ty) -> bool`` restricts the blend to a subset of tiles (bounded CPU-baseline
    samples only; the other tiles are left at the background colour)."""
    pre = preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree,
                     shs, colors_precomp, cov3D_precomp, scales, rotations, scale_modifier, depth_grad=depth_grad,
                     sh_cap=sh_cap)
    point_list, ranges, keys, N = bin_tiles(pre, W, H)
    color, final_T, n_contrib, depth_img = blend(pre, point_list, ranges, bg, W, H, tile_filter=tile_filter, aux=aux)
    if return_state:
        return color, pre["radii"].to(torch.int32), depth_img, dict(
            pre=pre, point_list=point_list, ranges=ranges, keys=keys, num_rendered=N, final_T=final_T,
            n_contrib=n_contrib)
    return color, pre["radii"].to(torch.int32), depth_img


def rasterize_global_sort(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy,
                          sh_degree=0, shs=None, colors_precomp=None, cov3D_precomp=None, respect_rect=True):
    """Deliberately different formulation used as a cross-check: no key emission,
    no per-tile lists, no ranges — every pixel walks ALL visible Gaussians in global (depth, index)
    order.  With ``respect_rect`` a Gaussian only touches pixels of the tiles in its rect (upstream's
    `getRect` is not a strict superset of the 3σ disc: its upper bound truncates, so a pixel up to
    one pixel inside the radius can lie in an excluded tile); without it the result differs from
    ``rasterize`` exactly on those pixels."""
    pre = preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree,
                     shs, colors_precomp, cov3D_precomp)
    dt = means3D.dtype
    vis = pre["visible"].nonzero().squeeze(-1)
    dbits = pre["depth"][vis].detach().to(torch.float32).view(torch.int32).to(torch.int64)
    order = torch.sort(dbits, stable=True).indices
    ids = vis[order]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pixx, pixy = xs.reshape(-1), ys.reshape(-1)
    T = torch.ones(H * W, dtype=dt)
    C = torch.zeros(H * W, 3, dtype=dt)
    done = torch.zeros(H * W, dtype=torch.bool)
    for g in ids.tolist():
        dx = pre["xy"][g, 0] - pixx
        dy = pre["xy"][g, 1] - pixy
        con = pre["conic"][g]
        power = -0.5 * (con[0] * dx * dx + con[2] * dy * dy) - con[1] * dx * dy
        alpha = (pre["opacity"][g] * torch.exp(power)).clamp(max=ALPHA_MAX)
        ok = (power <= 0) & (alpha >= ALPHA_MIN) & ~done
        if respect_rect:
            rminx, rminy, rmaxx, rmaxy = [int(r[g]) for r in pre["rect"]]
            tx_, ty_ = (pixx / TILE).floor(), (pixy / TILE).floor()
            ok = ok & (tx_ >= rminx) & (tx_ < rmaxx) & (ty_ >= rminy) & (ty_ < rmaxy)
        testT = T * (1 - alpha)
        stop = ok & (testT < T_MIN)
        done = done | stop
        ok = ok & ~stop
        C = C + torch.where(ok, alpha * T, torch.zeros_like(T))[:, None] * pre["rgb"][g][None]
        T = torch.where(ok, testT, T)
    out = C + T[:, None] * bg.to(dt)[None]
    return out.T.reshape(3, H, W)


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """PSNR for [0,1] images (formula of reference utils_loc.py img2psnr: -10·log10(mse))."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)
