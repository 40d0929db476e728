"""The per-tile depth sort (`depth_sort="per_tile"`, csrc/tile_sort.hip) builds the SAME lists as the global depth sort
and as the oracle's stable 64-bit sort — entry for entry (`-m gpu`).

The reference sorts all (tile << 32 | depth bits, id) pairs once (ties keep ascending id).  The
build has two forms of that step with identical results: a global depth sort of the Gaussians in front of the tile-list build,
and — for frames of many tiles with short lists — lists built in id order and sorted tile by tile in LDS.  Covered here:
random scenes, exact depth ties, lists longer than the small / the large launch class (2048 / 8192 entries: the second falls
back to the global sort inside the call), the guessed-buffer mode, the sync-free mode's overflow flag, launch sets of several
views and sets, and what `auto` means.
"""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from tests.helpers import oracle_forward

pytestmark = pytest.mark.gpu


def _state(sc, mode, **kw):
    from ggrt_official_amd.rasterizer import debug_forward_state, last_forward_binning
    s = sc.to("cuda:0")
    out = debug_forward_state(s.means3D, s.opacities, s.settings()._replace(depth_sort=mode, **kw), shs=s.shs,
                              cov3D_precomp=s.cov3D)
    return out, last_forward_binning()


def _same_lists(a, b):
    assert a["num_rendered"] == b["num_rendered"]
    assert torch.equal(a["ranges"], b["ranges"])
    assert torch.equal(a["point_list"], b["point_list"])
    for k in ("color", "out_depth", "radii", "final_T", "n_contrib"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("P,W,H,seed", [(20000, 256, 256, 0), (200000, 504, 378, 1), (150000, 1280, 720, 2), (1, 64, 64, 3),
                                        (777, 48, 33, 4)])
def test_per_tile_lists_equal_global_and_oracle(P, W, H, seed):
    from ggrt_official_amd.rasterizer import clear_list_hints
    clear_list_hints()
    sc = make_scene(P, W, H, sh_degree=1, profile="A", seed=seed)
    glob, how_g = _state(sc, "global")
    tile, how_t = _state(sc, "per_tile")
    assert how_g[0] == "global" and how_t[0] == "per_tile"
    _same_lists(tile, glob)
    st = oracle_forward(sc)
    assert np.array_equal(tile["point_list"].cpu().numpy().astype(np.uint32), st.point_list)
    assert np.array_equal(tile["ranges"].cpu().numpy(), st.ranges)
    lens = (st.ranges[:, 1] - st.ranges[:, 0]).astype(np.int64)
    assert how_t[1] == int(lens.max()) and how_g[1] == int(lens.max())


def test_exact_depth_ties_keep_ascending_id_order_per_tile():
    """All Gaussians on a few depth planes (and one plane only): ties must keep ascending Gaussian id."""
    for planes in (1, 3):
        sc = make_scene(30000, 320, 240, sh_degree=0, profile="A", seed=5)
        z = 4.0 + (torch.arange(30000) % planes).float()
        sc.means3D[:, :2] *= (z / sc.means3D[:, 2])[:, None]
        sc.means3D[:, 2] = z
        glob, _ = _state(sc, "global")
        tile, how = _state(sc, "per_tile")
        assert how[0] == "per_tile"
        _same_lists(tile, glob)
        if planes == 1:
            pl, rg = tile["point_list"].cpu().numpy().astype(np.int64), tile["ranges"].cpu().numpy()
            for r0, r1 in rg:
                assert np.all(np.diff(pl[r0:r1]) > 0)


def _pile(P, W=1024, H=1024, frac=0.9, seed=7):
    """`frac` of the Gaussians squeezed onto the centre of the frame: one tile's list holds nearly all of them"""
    import math
    sc = make_scene(P, W, H, sh_degree=0, profile="A", seed=seed)
    k = int(frac * P)
    fpx = 0.5 / math.tan(math.radians(30.0)) * W          # focal length in pixels (synthetic.camera_matrices)
    off = 8.0 * sc.means3D[:k, 2] / fpx                    # 8 px right / below the frame centre: the middle of a tile
    sc.means3D[:k, 0] = sc.means3D[:k, 0] * 0.002 + off
    sc.means3D[:k, 1] = sc.means3D[:k, 1] * 0.002 + off
    return sc


@pytest.mark.parametrize("P,expect", [(3000, "per_tile"), (7000, "per_tile"), (12000, "global")])
def test_long_lists_large_class_and_fallback(P, expect):
    """Longest list ≈ 0.9·P: 2 700 (small launch class is enough: ≤ 2048? no — large class), 6 300 (large class),
    10 800 (> 8192: the call rebuilds the lists through the global sort and says so)."""
    from ggrt_official_amd.rasterizer import clear_list_hints, list_hint_stats
    sc = _pile(P)
    glob, _ = _state(sc, "global")
    assert glob["ranges"][:, 1].sub(glob["ranges"][:, 0]).max().item() > 2048
    for attempt in ("exact", "hinted", "hinted-again"):   # first call of a shape: upstream's order; then with guesses
        if attempt == "exact":
            clear_list_hints()
        list_hint_stats(reset=True)
        tile, how = _state(sc, "per_tile")
        assert how[0] == expect, (attempt, how)
        _same_lists(tile, glob)
        stats = list_hint_stats()
        assert stats["exact" if attempt == "exact" else "hinted" if expect == "per_tile" else "missed"] == 1, (attempt, stats)


@pytest.mark.parametrize("K", [2, 63, 64, 65, 255, 256, 257, 1023, 1025, 2047, 2048, 2049, 3071, 3072, 3073, 4095, 4096, 4097,
                               8191, 8192, 8193])
def test_list_lengths_at_the_class_boundaries(K):
    """ONE tile's list of exactly K entries (K tiny Gaussians in the middle of a tile, a few hundred elsewhere): the wave /
    round / class boundaries of the per-tile sort — 64, 256, 2048, 3072, 4096, 8192 — and one beyond the last (global rebuild);
    depths drawn from few values so that equal keys (ties by id) and clustered buckets (the LSD route) both occur."""
    import math
    from ggrt_official_amd.rasterizer import clear_list_hints
    W = H = 256
    P = K + 300
    sc = make_scene(P, W, H, sh_degree=0, profile="A", seed=K)
    g = torch.Generator().manual_seed(K)
    fpx = 0.5 / math.tan(math.radians(30.0)) * W
    z = sc.means3D[:K, 2].clone()
    if K % 3 == 0:
        z = 2.0 + torch.randint(0, 7, (K,), generator=g).float() * 0.5            # seven depth planes: ties, big buckets
    elif K % 3 == 1:
        z = 3.0 + torch.rand(K, generator=g) * 1e-4                                # one tight cluster
    u = 136.0 + torch.rand(K, generator=g) * 4.0                                   # inside tile (8, 8): pixels 128 … 143
    v = 136.0 + torch.rand(K, generator=g) * 4.0
    sc.means3D[:K, 0] = (u - 0.5 * W) / fpx * z
    sc.means3D[:K, 1] = (v - 0.5 * H) / fpx * z
    sc.means3D[:K, 2] = z
    sc.cov3D[:K] = 0.0
    s2 = (0.3 * z / fpx) ** 2                                                      # σ = 0.3 px: radius 3 → stays inside the tile
    sc.cov3D[:K, 0] = s2; sc.cov3D[:K, 3] = s2; sc.cov3D[:K, 5] = s2
    clear_list_hints()
    glob, how_g = _state(sc, "global")
    lens = glob["ranges"][:, 1] - glob["ranges"][:, 0]
    assert int(lens.max()) >= K and how_g[1] == int(lens.max())
    for attempt in range(2):      # exact (nothing known), then with guesses
        tile, how = _state(sc, "per_tile")
        assert how[0] == ("per_tile" if int(lens.max()) <= 8192 else "global"), how
        _same_lists(tile, glob)


def test_length_guess_too_small_is_repaired():
    """Two frames of one shape: the first with short lists, the second with a list of the large class — the second runs
    with the first's guesses (no large launch enqueued) and must repair itself."""
    from ggrt_official_amd.rasterizer import clear_list_hints, list_hint_stats
    clear_list_hints()
    a = make_scene(6000, 1024, 1024, sh_degree=0, profile="A", seed=8)
    b = _pile(6000, frac=0.6, seed=8)
    _state(a, "per_tile")                      # exact: notes (N, longest) of the short-list frame
    list_hint_stats(reset=True)
    # the pile touches few tiles: its N is below the first frame's, only the LENGTH guess fails
    tile, how = _state(b, "per_tile")
    assert how[0] == "per_tile" and how[1] > 2048
    assert list_hint_stats()["missed"] == 1
    clear_list_hints()
    glob, _ = _state(b, "global")
    _same_lists(tile, glob)


def test_sync_free_per_tile_overflow_flag():
    from ggrt_official_amd.rasterizer import last_forward_status
    short, long_ = make_scene(5000, 512, 512, sh_degree=0, profile="A", seed=9), _pile(12000, 512, 512)
    for sc, want_overflow in ((short, False), (long_, True)):
        tile, how = _state(sc, "per_tile", list_capacity=400000)
        n, ov = last_forward_status()
        assert how[0] == "per_tile" and ov == want_overflow
        _, how_g = _state(sc, "global", list_capacity=400000)
        assert how_g[0] == "global" and last_forward_status()[1] is False


def test_what_auto_picks(monkeypatch):
    """`auto`: per tile for frames of short lists (<= 256 Gaussians per tile on average; once a call of the shape has reported
    it, a longest list <= 4096), global for GGRt-like frames and in the sync-free mode; GGR_DEPTH_SORT overrides `auto` only."""
    from ggrt_official_amd.rasterizer import clear_list_hints
    clear_list_hints()
    monkeypatch.delenv("GGR_DEPTH_SORT", raising=False)
    sparse = make_scene(50000, 1920, 1080, sh_degree=0, profile="A", seed=10)      # 6 per tile
    dense = make_scene(300000, 480, 352, sh_degree=0, profile="B", seed=10)        # 455 per tile
    assert _state(sparse, "auto")[1][0] == "per_tile" and _state(dense, "auto")[1][0] == "global"
    assert _state(sparse, "auto", list_capacity=2000000)[1][0] == "global"          # sync-free: no read-back to fall back with
    # a frame whose average is low but whose longest list is beyond 4096: per tile on the first call (nothing known), global
    # once the shape's history says so
    clear_list_hints()
    pile = _pile(7000)
    first, second = _state(pile, "auto")[1], _state(pile, "auto")[1]
    assert first[0] == "per_tile" and first[1] > 4096 and second[0] == "global"
    monkeypatch.setenv("GGR_DEPTH_SORT", "global")
    assert _state(sparse, "auto")[1][0] == "global" and _state(sparse, "per_tile")[1][0] == "per_tile"


def _on_depth_planes(sc, planes=(2.0, 9.0, 25.0), seed=1):
    """the scene's Gaussians moved along their view rays onto a few planes of EXACTLY equal depth"""
    g = torch.Generator().manual_seed(seed)
    z = torch.tensor(planes)[torch.randint(0, len(planes), (sc.means3D.shape[0],), generator=g)]
    s = z / sc.means3D[:, 2]
    sc.means3D *= s[:, None]
    sc.cov3D *= (s * s)[:, None]
    return sc


def test_clustered_depths_send_a_shape_back_to_the_global_sort(monkeypatch):
    """Depths that cluster (planes: every tile's keys fall into two or three buckets) send the per-tile sort down its slow
    route; the sort counts those entries (ggr_sort_stats_async word 3) and the host keeps such a shape on the global sort —
    same lists either way.  A frame of spread-out depths stays per tile."""
    import ggrt_official_amd
    from ggrt_official_amd.rasterizer import clear_list_hints, sort_watch_stats
    monkeypatch.delenv("GGR_DEPTH_SORT", raising=False)
    clear_list_hints()
    P, W, H = 150000, 1280, 720
    planes = _on_depth_planes(make_scene(P, W, H, sh_degree=0, profile="A", seed=40))
    ref, how = _state(planes, "global")
    assert how[0] == "global"
    used = []
    for _ in range(4):
        out, how = _state(planes, "auto")
        torch.cuda.synchronize()   # (the look at the frame is a copy queued behind it: let it land before the next call)
        used.append(how[0])
        _same_lists(out, ref)
    (calls, share, left), = sort_watch_stats().values()
    assert used[0] == "per_tile" and used[-1] == "global", used
    assert calls == 4 and share > 0.5 and left > 200
    # the explicit setting is not overruled
    assert _state(planes, "per_tile")[1][0] == "per_tile"
    # spread-out depths: nothing to go back for
    clear_list_hints()
    spread = make_scene(P, W, H, sh_degree=0, profile="A", seed=41)
    for _ in range(4):
        how = _state(spread, "auto")[1]
        torch.cuda.synchronize()
        assert how[0] == "per_tile"
    (calls, share, left), = sort_watch_stats().values()
    assert share is not None and share < 0.05 and left == 0
    # the raw words, through the C ABI: N, status, longest list, slow-route entries
    from ggrt_official_amd import _lib
    lib = _lib.load()
    st = debug_state_geom(spread)
    words = torch.zeros(4, dtype=torch.int32).pin_memory()
    assert lib.ggr_sort_stats_async(st["geom"].data_ptr(), P, words.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert int(words[0]) == st["num_rendered"] and int(words[2]) == st["longest"] and 0 <= int(words[3]) <= int(words[0])
    assert ggrt_official_amd.sort_watch_stats is sort_watch_stats


def debug_state_geom(sc):
    """one per-tile forward through the module, its geometry buffer kept"""
    from ggrt_official_amd.rasterizer import _RasterizeGaussians, last_forward_binning

    class Ctx:
        needs_input_grad = (True,)
        def set_materialize_grads(self, v): pass
        def save_for_backward(self, *t): self.saved = t
        def mark_non_differentiable(self, *t): pass
    s = sc.to("cuda:0")
    rs = s.settings()._replace(depth_sort="per_tile")
    ctx = Ctx()
    _RasterizeGaussians.forward(ctx, s.means3D, torch.zeros_like(s.means3D), s.shs, None, s.opacities, None, None, s.cov3D,
                                rs.viewmatrix, rs.projmatrix, rs.campos, None, rs)
    return {"geom": ctx.saved[12], "num_rendered": ctx.num_rendered, "longest": last_forward_binning()[1]}


def test_views_and_sets_per_tile_equal_global():
    from ggrt_official_amd.rasterizer import rasterize_views
    from ggrt_official_amd.synthetic import camera_matrices
    dev = torch.device("cuda:0")
    B, V, P, W, H = 2, 4, 20000, 640, 480
    scs = [make_scene(P, W, H, sh_degree=1, profile="A", seed=20 + b) for b in range(B)]
    views, projs, cams = [], [], []
    for v in range(V):
        c2w = torch.eye(4, dtype=torch.float64)
        c2w[0, 3] = 0.05 * v
        view, full, campos, tfx, tfy, _, _ = camera_matrices(W, H, c2w=c2w)
        views.append(view); projs.append(full); cams.append(campos)
    views, projs, cams = (torch.stack(t).to(dev) for t in (views, projs, cams))
    bg = torch.zeros(V, 3, device=dev)
    tanfov = torch.tensor([[scs[0].tanfovx, scs[0].tanfovy]] * V, device=dev)
    stack = lambda k: torch.stack([getattr(s, k) for s in scs]).to(dev)
    dL = torch.stack([upstream_gradient(W, H, seed=30 + v) for v in range(V)]).to(dev)
    res = {}
    for mode in ("global", "per_tile"):
        leaves = {k: stack(k).requires_grad_(True) for k in ("means3D", "opacities", "shs", "cov3D")}
        rs = scs[0].to(dev).settings()._replace(depth_sort=mode)
        color, radii, depth = rasterize_views(leaves["means3D"], leaves["opacities"], views, projs, cams, bg, tanfov, rs,
                                              shs=leaves["shs"], cov3D_precomp=leaves["cov3D"])
        (color * dL).sum().backward()
        res[mode] = (color.detach(), radii, depth.detach(), {k: v.grad for k, v in leaves.items()})
    assert torch.equal(res["global"][0], res["per_tile"][0])
    assert torch.equal(res["global"][1], res["per_tile"][1])
    assert torch.equal(res["global"][2], res["per_tile"][2])
    for k, gg in res["global"][3].items():   # (atomic accumulation order: not bitwise)
        gt = res["per_tile"][3][k]
        assert (gg - gt).norm() <= 1e-5 * gg.norm(), k
