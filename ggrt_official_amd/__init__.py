"""ggrt_official_amd — MI355X-native differentiable 3D-Gaussian rasterizer, drop-in for the
``diff_gaussian_rasterization`` extension GGRt uses on its render hot path
(reference ``ggrt/model/pixelsplat/decoder/cuda_splatting.py``).

Scope: the rasterizer (forward + backward) behind a C ABI, the call-site glue
(`render_cuda`, `render_depth_cuda`, `DecoderSplattingCUDA`), one-frame-per-GPU sharding helpers and a
synthetic-scene generator for the benchmark.  Everything else of GGRt is out of scope.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, clear_list_hints, last_forward_status,
                         list_hint_stats, rasterize_gaussians, rasterize_views, set_list_hint, sort_watch_stats)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_views",
           "last_forward_status", "set_list_hint", "list_hint_stats", "clear_list_hints", "sort_watch_stats"]
__version__ = "0.1.0"
