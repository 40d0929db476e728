"""Edge cases of the HIP path against the CPU oracle (`-m gpu`): the sizes and degeneracies the
synthetic BASELINE scenes do not reach — multi-band tile tables (4K image), rects of thousands of
tiles, single-tile images, P = 1, exact depth ties (list stability), degenerate opacities, a partial
last chunk / last sort tile."""
import numpy as np
import pytest
import torch

from ggrt_official_amd.synthetic import make_scene, upstream_gradient
from oracle import c_oracle
from tests.helpers import check_grads, check_image, hip_forward_backward, oracle_forward, psnr, rel_l2

pytestmark = pytest.mark.gpu


def _state(sc):
    from ggrt_official_amd.rasterizer import debug_forward_state
    s = sc.to("cuda:0")
    out = debug_forward_state(s.means3D, s.opacities, s.settings(), shs=s.shs, cov3D_precomp=s.cov3D)
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}


def _check_lists(sc):
    st = oracle_forward(sc)
    cpu = _state(sc)
    assert cpu["num_rendered"] == st.num_rendered
    assert np.array_equal(cpu["radii"], st.radii)
    assert np.array_equal(cpu["point_list"].astype(np.uint32), st.point_list)
    assert np.array_equal(cpu["ranges"], st.ranges)
    check_image(cpu["color"], st.color, tag="edge_lists")
    return st, cpu


def test_4k_image_with_huge_splats():
    """3840×2160 → 32 400 tiles (8 count bands, 32 scatter bands); splats covering thousands of tiles."""
    sc = make_scene(1500, 3840, 2160, sh_degree=1, profile="A", seed=3)
    sc.cov3D *= 400.0          # σ × 20 → radii of several hundred pixels
    sc.opacities.clamp_(max=0.5)
    st, cpu = _check_lists(sc)
    assert st.tiles_touched.max() > 2000 and st.num_rendered > 300_000


def test_single_tile_image_and_single_gaussian():
    sc = make_scene(300, 16, 16, sh_degree=2, profile="A", seed=1)
    _check_lists(sc)
    sc1 = make_scene(1, 64, 48, sh_degree=0, profile="A", seed=5)
    sc1.means3D[:] = torch.tensor([[0.0, 0.0, 4.0]])
    st, cpu = _check_lists(sc1)
    assert st.num_rendered >= 1


def test_exact_depth_ties_keep_ascending_id_order():
    """All Gaussians on ONE depth plane: every list must be in ascending Gaussian id (stable sort)."""
    sc = make_scene(6000, 160, 128, sh_degree=0, profile="A", seed=2)
    sc.means3D[:, :2] *= 7.0 / sc.means3D[:, 2:3]
    sc.means3D[:, 2] = 7.0
    st, cpu = _check_lists(sc)
    for r0, r1 in st.ranges:
        seg = cpu["point_list"][r0:r1]
        assert np.all(np.diff(seg.astype(np.int64)) > 0)


def test_sizes_that_straddle_chunk_and_sort_tile_boundaries():
    for P in (1023, 1025, 4097, 8192 + 7):
        sc = make_scene(P, 176, 96, sh_degree=1, profile="A", seed=P)
        _check_lists(sc)


def test_degenerate_opacities_and_gradients():
    sc = make_scene(3000, 96, 80, sh_degree=1, profile="A", seed=9)
    sc.opacities[::5] = 0.0          # never contributes
    sc.opacities[1::5] = 1.0         # clamped to 0.99
    sc.opacities[2::5] = 1e-3        # below 1/255 everywhere
    dL = upstream_gradient(96, 80, seed=4)
    st = oracle_forward(sc)
    ref = c_oracle.backward(st, dL.numpy())
    color, radii, depth, grads = hip_forward_backward(sc, dL)
    assert np.array_equal(radii, st.radii)
    check_image(color, st.color, tag="edge_opacities")
    check_grads(grads, ref, ("means3D", "shs", "opacities", "cov3D_precomp"), tag="edge_opacities")
    assert np.all(grads["opacities"][2::5] == 0) and np.all(grads["means3D"][::5] == 0)


def test_colors_precomp_with_aux_feature():
    """colors_precomp (no SH) together with the aux channel: aux image = Σ aux·α·T, gradient to aux."""
    from ggrt_official_amd import GaussianRasterizer
    from oracle import torch_raster as tr
    sc = make_scene(2000, 80, 64, sh_degree=0, profile="A", seed=6)
    g = torch.Generator().manual_seed(1)
    colors, aux = torch.rand(2000, 3, generator=g), torch.randn(2000, generator=g)
    dL, dLa = upstream_gradient(80, 64, seed=2), upstream_gradient(80, 64, seed=3)[0]
    lf = lambda t: t.double().clone().requires_grad_(True)
    c64, a64, m64 = lf(colors), lf(aux), lf(sc.means3D)
    color, _, aimg = tr.rasterize(m64, sc.opacities.double(), sc.viewmatrix.double(), sc.projmatrix.double(),
                                  sc.campos.double(), sc.bg, 80, 64, sc.tanfovx, sc.tanfovy, 0, colors_precomp=c64,
                                  cov3D_precomp=sc.cov3D.double(), aux=a64)
    ((color * dL.double()).sum() + (aimg * dLa.double()).sum()).backward()
    dev = "cuda:0"
    s = sc.to(dev)
    ch, ah, mh = [t.clone().to(dev).requires_grad_(True) for t in (colors, aux, sc.means3D)]
    col_h, _, aimg_h = GaussianRasterizer(s.settings())(means3D=mh, means2D=torch.zeros_like(mh), opacities=s.opacities,
                                                       colors_precomp=ch, cov3D_precomp=s.cov3D, aux_precomp=ah)
    ((col_h * dL.to(dev)).sum() + (aimg_h * dLa.to(dev)).sum()).backward()
    assert np.abs(aimg_h.detach().cpu().numpy() - aimg.detach().numpy()).max() < 1e-4
    assert rel_l2(ah.grad.cpu().numpy(), a64.grad.numpy()) < 1e-3
    assert rel_l2(ch.grad.cpu().numpy(), c64.grad.numpy()) < 1e-3
    assert rel_l2(mh.grad.cpu().numpy(), m64.grad.numpy()) < 1e-3


@pytest.mark.parametrize("P", [1, 5, 83, 84, 85, 168, 169, 255, 257])
@pytest.mark.parametrize("cap", [4, 3])
def test_long_sh_rows_with_few_gaussians(P, cap):
    """25-coefficient SH rows (GGRt's) take the long-row paths of the preprocess kernels: rows through LDS a third at a
    time, gradient rows written in three row ranges of 84 / 84 / 88 Gaussians.  Block sizes that leave ranges empty,
    end inside a range or on its border; both `sh_max_degree` settings."""
    sc = make_scene(P, 64, 48, sh_degree=4, profile="A", seed=P)
    dL = upstream_gradient(64, 48, seed=3)
    st = oracle_forward(sc, sh_cap=cap)
    ref = c_oracle.backward(st, dL.numpy())
    color, radii, _, grads = hip_forward_backward(sc, dL, sh_max_degree=cap)
    assert np.array_equal(radii, st.radii)
    check_image(color, st.color, tag="few_gaussians")
    if st.num_rendered:
        assert rel_l2(grads["shs"], ref["shs"]) < 2e-5 and rel_l2(grads["means3D"], ref["means3D"]) < 2e-5
    K = (cap + 1) ** 2
    assert np.all(grads["shs"][:, K:, :] == 0)
    assert np.all(grads["shs"][st.radii <= 0] == 0)   # culled Gaussians: zero rows
