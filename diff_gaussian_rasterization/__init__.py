"""Import-name shim: `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` — the line GGRt has at ggrt/model/pixelsplat/decoder/cuda_splatting.py:6-9 —
resolves to the MI355X-native rasterizer when this repository is on PYTHONPATH.

One checkpoint, one answer: GGRt's own call site builds its settings without `sh_max_degree` (the field does not exist
upstream, cuda_splatting.py:101-113).  Through THIS package name such settings take the call-site layer's default for the
highest SH band (`ggrt_official_amd.splatting.SH_MAX_DEGREE`: `set_sh_max_degree` / `GGR_SH_MAX_DEGREE`, 4 unless chosen
otherwise — INTEGRATION.md §7), silently, exactly as `ggrt_official_amd.splatting.render_cuda` does: the two documented
integration paths render the same images.  (`ggrt_official_amd.GaussianRasterizer`, the raw rasterizer, keeps "not chosen
= bands 0..3 with one warning".)  Nothing else lives here."""
from ggrt_official_amd import rasterizer as _r
from ggrt_official_amd.rasterizer import GaussianRasterizationSettings  # noqa: F401


def _with_call_site_cap(raster_settings):
    if int(getattr(raster_settings, "sh_max_degree", 0) or 0) == 0:
        from ggrt_official_amd import splatting
        return raster_settings._replace(sh_max_degree=splatting.resolve_sh_max_degree(None))
    return raster_settings


class GaussianRasterizer(_r.GaussianRasterizer):
    """`ggrt_official_amd.GaussianRasterizer` with the call-site layer's SH-cap default for settings that leave it open."""

    def _settings_for_call(self):
        return _with_call_site_cap(self.raster_settings)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, aux_precomp=None):
    return _r.rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                  _with_call_site_cap(raster_settings), aux_precomp)


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
