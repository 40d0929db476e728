"""The global depth sort's BUCKET form (csrc/binning.hip, ABI 11): one stable partition pass into equally full depth buckets +
every bucket sorted in LDS — the same order as the three stable radix passes, key for key, hence the same tile lists
(`-m gpu`).  The reference sorts all (tile << 32 | depth bits, id) pairs once (ties keep ascending id).

Covered: random scenes of both depth profiles against the three-pass form and the oracle; every key equal (one bucket of far
more keys than a workgroup sorts: copied out, no fault); planes of equal depth beside a spread; a thin slab of > 8192 DIFFERENT
keys inside one fine bin — the one case the form gives up: the call sorts again in three passes, says so, and the host keeps the
shape on three passes for a while; culled Gaussians; launch sets (one segment per view); the sync-free mode (three passes:
nothing to fall back with)."""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene
from tests.helpers import oracle_forward

pytestmark = pytest.mark.gpu


def _state(sc, mode, **kw):
    from ggrt_official_amd.rasterizer import debug_forward_state, last_forward_sort_form
    s = sc.to("cuda:0")
    out = debug_forward_state(s.means3D, s.opacities, s.settings()._replace(depth_sort=mode, **kw), shs=s.shs,
                              cov3D_precomp=s.cov3D)
    return out, last_forward_sort_form()


def _same_lists(a, b):
    assert a["num_rendered"] == b["num_rendered"]
    assert torch.equal(a["ranges"], b["ranges"])
    assert torch.equal(a["point_list"], b["point_list"])
    for k in ("color", "out_depth", "radii", "final_T", "n_contrib"):
        assert torch.equal(a[k], b[k]), k


@pytest.fixture(autouse=True)
def _fresh(monkeypatch):
    from ggrt_official_amd.rasterizer import clear_list_hints
    monkeypatch.delenv("GGR_GLOBAL_SORT", raising=False)
    monkeypatch.delenv("GGR_DEPTH_SORT", raising=False)
    clear_list_hints()
    yield
    clear_list_hints()


@pytest.mark.parametrize("P,W,H,profile,seed", [(20000, 256, 256, "A", 0), (300000, 504, 378, "A", 1), (150000, 1280, 720, "B", 2),
                                                (1, 64, 64, "A", 3), (777, 48, 33, "B", 4), (1100000, 480, 352, "B", 5)])
def test_bucket_form_equals_three_passes_and_the_oracle(P, W, H, profile, seed):
    sc = make_scene(P, W, H, sh_degree=1, profile=profile, seed=seed)
    buck, how_b = _state(sc, "global")
    three, how_3 = _state(sc, "global_3pass")
    assert how_b == "buckets" and how_3 == "3pass"
    _same_lists(buck, three)
    if P <= 300000:
        st = oracle_forward(sc)
        assert np.array_equal(buck["point_list"].cpu().numpy().astype(np.uint32), st.point_list)
        assert np.array_equal(buck["ranges"].cpu().numpy(), st.ranges)


def _at_depths(sc, z):
    """the scene's Gaussians moved along their view rays (identity camera pose of make_scene: depth = z) onto depths z"""
    z = z.to(sc.means3D.dtype)
    sc.means3D[:, :2] *= (z / sc.means3D[:, 2])[:, None]
    sc.means3D[:, 2] = z
    return sc


def test_every_key_equal_is_one_oversized_bucket_and_no_fault():
    P = 60000
    sc = _at_depths(make_scene(P, 320, 240, sh_degree=0, profile="A", seed=7), torch.full((P,), 5.0))
    buck, how = _state(sc, "global")
    three, _ = _state(sc, "global_3pass")
    assert how == "buckets"
    _same_lists(buck, three)
    # ties keep ascending id inside every tile
    pl, rg = buck["point_list"].cpu().numpy(), buck["ranges"].cpu().numpy()
    for a, b in rg[(rg[:, 1] - rg[:, 0]) > 1][:50]:
        assert (np.diff(pl[a:b].astype(np.int64)) > 0).all()


def test_planes_of_equal_depth_beside_a_spread():
    P = 120000
    sc = make_scene(P, 640, 480, sh_degree=0, profile="A", seed=8)
    z = sc.means3D[:, 2].clone()
    z[:30000] = 3.0          # 30 000 equal keys: one fine bin, far beyond a workgroup's 8192 — but all equal
    z[30000:50000] = 7.25
    sc = _at_depths(sc, z)
    buck, how = _state(sc, "global")
    three, _ = _state(sc, "global_3pass")
    assert how in ("buckets", "fell_back")
    _same_lists(buck, three)


def _thin_slab_scene(P=150000, slab=20000, seed=9):
    """`slab` Gaussians on CONSECUTIVE float32 depths from 6.0 up (20 000 different keys inside 20 000 ulps: at most two of the
    4096 fine bins of a frame whose depths spread over [0.5, 2000] — 24 000 ulps each), the rest spread over that range"""
    sc = make_scene(P, 640, 480, sh_degree=0, profile="A", seed=seed)
    g = torch.Generator().manual_seed(seed)
    z = 0.5 * torch.exp(torch.rand(P, generator=g) * np.log(4000.0))
    z[:slab] = torch.from_numpy(np.float32(6.0).view(np.uint32) + np.arange(slab, dtype=np.uint32)).view(torch.float32)[torch.randperm(slab, generator=g)]
    return _at_depths(sc, z)


def test_thin_slab_falls_back_to_three_passes_inside_the_call_and_the_host_remembers():
    from ggrt_official_amd.rasterizer import sort_watch_stats
    sc = _thin_slab_scene()
    assert len(torch.unique(sc.means3D[:20000, 2])) > 8192
    three, how_3 = _state(sc, "global_3pass")
    assert how_3 == "3pass"
    forms = []
    for _ in range(3):
        out, how = _state(sc, "global")
        forms.append(how)
        _same_lists(out, three)
    # the first frame pays for the discovery, the shape then keeps the three passes
    assert forms[0] == "fell_back" and forms[1] == "3pass" and forms[2] == "3pass", forms
    # … and so with the exact mode and hints off: every call falls back, every call is right
    from ggrt_official_amd.rasterizer import set_list_hint
    set_list_hint(False)
    try:
        for _ in range(2):
            out, how = _state(sc, "global")
            assert how == "fell_back"
            _same_lists(out, three)
    finally:
        set_list_hint(True)


def test_fault_in_the_guessed_buffer_mode_is_repaired_at_the_end_of_the_call():
    """The default mode enqueues scatter and blend behind a GUESSED list buffer and reads the counts at the end of the call: a
    bucket fault shows up there, with the frame already blended from mis-ordered lists — the call builds the lists again in
    three passes, scatters and blends once more.  (A shape's first frames establish the guess; the faulting frame has the same
    shape.)"""
    from ggrt_official_amd.rasterizer import list_hint_stats
    P = 150000
    spread = make_scene(P, 640, 480, sh_degree=0, profile="A", seed=21)
    for _ in range(2):
        _, how = _state(spread, "global")
        assert how == "buckets"
    slab = _thin_slab_scene(P=P, seed=22)
    three, _ = _state(slab, "global_3pass")
    list_hint_stats(reset=True)
    out, how = _state(slab, "global")
    assert how == "fell_back"
    assert list_hint_stats()["exact"] == 0        # (it was a guessed-buffer forward)
    _same_lists(out, three)


def test_concentrated_depths_are_sorted_the_slow_way_once_and_the_host_remembers():
    """A few far outliers and everything else inside a seventh of an octave: 4096 fine bins over the frame's range leave the bulk
    some forty of them, every bucket is one overfull fine bin and the small launch sorts them a few workgroups wide — correct,
    slow, and said so (GGR_DEPTH_SORT_GLOBAL_SLOW): the shape then keeps the three passes."""
    P = 200000
    sc = make_scene(P, 640, 480, sh_degree=0, profile="A", seed=31)
    g = torch.Generator().manual_seed(31)
    z = 5.0 + 0.5 * torch.rand(P, generator=g)
    z[:5] = 0.3
    z[5:10] = 3000.0
    sc = _at_depths(sc, z)
    three, _ = _state(sc, "global_3pass")
    forms = []
    for _ in range(3):
        out, how = _state(sc, "global")
        forms.append(how)
        _same_lists(out, three)
    assert forms == ["buckets_slow", "3pass", "3pass"], forms


def test_culled_gaussians_and_tiny_frames():
    sc = make_scene(50000, 200, 120, sh_degree=0, profile="A", seed=10)
    sc.means3D[::3, 2] = -1.0       # a third behind the camera: key 0, the bucket of its own
    buck, how = _state(sc, "global")
    three, _ = _state(sc, "global_3pass")
    assert how == "buckets"
    _same_lists(buck, three)
    allc = make_scene(5000, 64, 64, sh_degree=0, profile="A", seed=11)
    allc.means3D[:, 2] = -1.0       # nothing visible at all
    buck, how = _state(allc, "global")
    assert buck["num_rendered"] == 0


def test_sync_free_mode_keeps_the_three_passes():
    sc = make_scene(40000, 320, 240, sh_degree=0, profile="A", seed=12)
    out, how = _state(sc, "global", list_capacity=600000)
    assert how == "3pass"
    ref, how_b = _state(sc, "global")
    assert how_b == "buckets"
    for k in ("color", "out_depth", "radii", "ranges"):
        assert torch.equal(out[k], ref[k]), k


def test_launch_set_of_views_sorts_every_view_in_buckets(monkeypatch):
    """four views of one Gaussian set in one launch set: the bucket form sorts one segment per view — image, depth, radii
    bit-identical to the three-pass form (same lists, same blend), gradients within the suite's bars"""
    from ggrt_official_amd.synthetic import upstream_gradient
    from tests.helpers import check_grads
    from tests.test_gpu_views_batched import _cameras, _run
    P, W, H, V = 60000, 200, 144, 4
    sc = make_scene(P, W, H, sh_degree=2, profile="B", seed=13)
    cams = _cameras(W, H, V)
    g = torch.Generator().manual_seed(5)
    dLs = torch.stack([upstream_gradient(W, H, seed=10 + v) for v in range(V)])
    bgs = torch.rand(V, 3, generator=g)
    args = (sc, cams, dLs, None, None, True, True, bgs, None, None)
    monkeypatch.setenv("GGR_DEPTH_SORT", "global")
    res = {}
    for form in ("buckets", "3pass"):
        monkeypatch.setenv("GGR_GLOBAL_SORT", form)
        res[form] = _run(*args, batched=True)
    (ca, ra, da, ga), (cb, rb, db, gb) = res["buckets"], res["3pass"]
    assert np.array_equal(ra, rb) and np.array_equal(ca, cb) and np.array_equal(da, db)
    check_grads(ga, gb, [k for k in ga if ga[k] is not None], tag="views-buckets")
