"""Import-name shim: `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` — the line GGRt has at ggrt/model/pixelsplat/decoder/cuda_splatting.py:6-9 —
resolves to the MI355X-native rasterizer when this repository is on PYTHONPATH.  Nothing else lives here."""
from ggrt_official_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                          rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
