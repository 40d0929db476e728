"""Deterministic synthetic scenes for the parity tests and the benchmark.

The reference ships no data, checkpoints or fixtures for the rasterizer boundary, so the
BASELINE.json configurations are restated as seeded random scenes:

* camera: identity pose at the origin looking down +z, horizontal FoV 60°, principal point at
  the image centre, near 1 / far 100, black background;  view / projection matrices are built
  exactly the way the reference call site builds them (``cuda_splatting.py:18-46,82-89``);
* profile **A** ("3DGS-like"): means uniformly over the image (+5 % margin), depth log-uniform in
  [1.5, 50], screen-space σ log-uniform in [0.3, 8] px, anisotropy U[1,4], opacity U[0.05, 1];
* profile **B** ("GGRt-like", reference ``encoder_epipolar.py:189-195``, ``gaussian_adapter.py:62-69``):
  pixel-aligned means, σ log-uniform in [0.1, 3] px, opacity = max softmax prob over 32 buckets / 3;
* SH: DC ~ N(0,1), band ℓ ~ N(0, (0.1·0.25^ℓ)²) (mirrors ``gaussian_adapter.py:45-46``).

Everything is generated on the CPU with a seeded ``torch.Generator`` (bit-reproducible across
machines) and then moved to the requested device.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class Scene:
    means3D: torch.Tensor        # [P,3]
    cov3D: torch.Tensor          # [P,6]  (00,01,02,11,12,22)
    scales: torch.Tensor         # [P,3]
    rotations: torch.Tensor      # [P,4]  unit quaternion (r,x,y,z)
    opacities: torch.Tensor      # [P,1]
    shs: torch.Tensor            # [P,M,3]
    viewmatrix: torch.Tensor     # [4,4]
    projmatrix: torch.Tensor     # [4,4]
    campos: torch.Tensor         # [3]
    bg: torch.Tensor             # [3]
    tanfovx: float
    tanfovy: float
    width: int
    height: int
    sh_degree: int

    def to(self, device) -> "Scene":
        kw = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()}
        return Scene(**kw)

    def settings(self, debug: bool = False):
        from .rasterizer import GaussianRasterizationSettings
        return GaussianRasterizationSettings(
            image_height=self.height, image_width=self.width, tanfovx=self.tanfovx, tanfovy=self.tanfovy,
            bg=self.bg, scale_modifier=1.0, viewmatrix=self.viewmatrix, projmatrix=self.projmatrix,
            sh_degree=self.sh_degree, campos=self.campos, prefiltered=False, debug=debug,
            sh_max_degree=3)  # (explicit: the synthetic D = 4 / M = 25 shapes evaluate bands 0..3)


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


def camera_matrices(width: int, height: int, fov_x_deg: float = 60.0, near: float = 1.0, far: float = 100.0,
                    c2w: torch.Tensor | None = None, cx: float = 0.5, cy: float = 0.5):
    """view / full-projection matrices in the reference's convention (``cuda_splatting.py:18-46,86-89``):
    viewmatrix = (c2w^-1)^T, projmatrix = viewmatrix @ P^T with the GGRt-modified P."""
    tanfovx = math.tan(math.radians(fov_x_deg) / 2)
    fx_n = 0.5 / tanfovx                    # focal length normalised by the width
    fy_n = fx_n * width / height            # square pixels
    tanfovy = 0.5 / fy_n
    Pm = torch.zeros(4, 4, dtype=torch.float64)
    Pm[0, 0] = 2 * near * fx_n
    Pm[1, 1] = 2 * near * fy_n
    Pm[0, 2] = 2 * cx - 1
    Pm[1, 2] = 2 * cy - 1
    Pm[3, 2] = 1
    Pm[2, 2] = far / (far - near)
    Pm[2, 3] = -(far * near) / (far - near)
    if c2w is None:
        c2w = torch.eye(4, dtype=torch.float64)
    c2w = c2w.double()
    view = torch.linalg.inv(c2w).T
    full = view @ Pm.T
    return view.float(), full.float(), c2w[:3, 3].float().clone(), tanfovx, tanfovy, fx_n, fy_n


def make_scene(num_points: int, width: int, height: int, sh_degree: int = 3, profile: str = "A", seed: int = 0,
               sh_stride: int | None = None, c2w: torch.Tensor | None = None, device="cpu",
               layout: str = "uniform") -> Scene:
    """``layout="lower_half"``: the same Gaussians squeezed into the lower half of the frame (upper half empty — a
    frame with sky): the spatially NON-uniform variant the benchmark reports next to the uniform one."""
    g = torch.Generator().manual_seed(seed)
    P = num_points
    view, full, campos, tanfovx, tanfovy, fx_n, fy_n = camera_matrices(width, height, c2w=c2w)
    fpx = fx_n * width  # focal length in pixels (same for y: square pixels)

    def rand(*s):
        return torch.rand(*s, generator=g, dtype=torch.float64)

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float64)

    if profile == "A":
        u = (rand(P) * 1.10 - 0.05) * width
        v = (rand(P) * 1.10 - 0.05) * height
        sig_px = torch.exp(math.log(0.3) + rand(P) * (math.log(8.0) - math.log(0.3)))
        opacity = 0.05 + 0.95 * rand(P)
    elif profile == "B":
        # pixel-aligned: tile the image with (roughly) P / (W·H) Gaussians per pixel
        idx = torch.arange(P, dtype=torch.float64)
        pix = idx % (width * height)
        u = pix % width + 0.5
        v = torch.floor(pix / width) + 0.5
        sig_px = torch.exp(math.log(0.1) + rand(P) * (math.log(3.0) - math.log(0.1)))
        logits = randn(P, 32) * 2.0
        opacity = torch.softmax(logits, -1).max(-1).values / 3.0
    else:
        raise ValueError(f"unknown profile {profile!r}")
    if layout == "lower_half":
        v = 0.5 * height + 0.5 * v
    elif layout != "uniform":
        raise ValueError(f"unknown layout {layout!r}")
    z = torch.exp(math.log(1.5) + rand(P) * (math.log(50.0) - math.log(1.5)))
    # unproject (camera frame == world frame for the identity pose; otherwise transform by c2w)
    xc = (u - 0.5 * width) / fpx * z
    yc = (v - 0.5 * height) / fpx * z
    pts_cam = torch.stack([xc, yc, z], -1)
    ratio = 1.0 + 3.0 * rand(P)
    s1 = sig_px * z / fpx
    s2 = s1 * ratio
    s3 = torch.sqrt(s1 * s2)
    scales = torch.stack([s1, s2, s3], -1)
    q = randn(P, 4)
    q = q / q.norm(dim=-1, keepdim=True)
    R = quat_to_rotmat(q)
    if c2w is not None:
        c2w64 = c2w.double()
        pts = pts_cam @ c2w64[:3, :3].T + c2w64[:3, 3]
    else:
        pts = pts_cam
    Mx = R * scales[:, None, :]
    S = Mx @ Mx.transpose(1, 2)
    cov6 = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)
    M = (sh_degree + 1) ** 2 if sh_stride is None else sh_stride
    shs = torch.zeros(P, M, 3, dtype=torch.float64)
    shs[:, 0] = randn(P, 3)
    k = 1
    for l in range(1, sh_degree + 1):
        n = 2 * l + 1
        if k + n > M:
            break
        shs[:, k:k + n] = randn(P, n, 3) * (0.1 * 0.25 ** l)
        k += n
    sc = Scene(means3D=pts.float(), cov3D=cov6.float(), scales=scales.float(), rotations=q.float(),
               opacities=opacity.float()[:, None], shs=shs.float(), viewmatrix=view, projmatrix=full, campos=campos,
               bg=torch.zeros(3), tanfovx=float(tanfovx), tanfovy=float(tanfovy), width=width, height=height,
               sh_degree=sh_degree)
    return sc.to(device) if str(device) != "cpu" else sc


def upstream_gradient(width: int, height: int, seed: int = 1234, device="cpu") -> torch.Tensor:
    """MSE-like upstream gradient dL/dcolor ~ N(0,1)/(3HW) with its own seed."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(3, height, width, generator=g) / (3.0 * height * width)).to(device)


# BASELINE.json configurations (C1..C3) and the GGRt-shaped stand-ins (C4', C5') — SURVEY.md §8a/§8d
CONFIGS = {
    "C1": dict(num_points=10_000, width=256, height=256, sh_degree=0, profile="A"),
    "C2": dict(num_points=200_000, width=504, height=378, sh_degree=3, profile="A"),
    "C3": dict(num_points=1_000_000, width=1920, height=1080, sh_degree=3, profile="A"),
    "C3_lower_half": dict(num_points=1_000_000, width=1920, height=1080, sh_degree=3, profile="A", layout="lower_half"),
    "C4p": dict(num_points=1_146_880, width=448, height=320, sh_degree=4, profile="B"),
    "C5p": dict(num_points=1_013_760, width=480, height=352, sh_degree=4, profile="B"),
    # Waymo eval shape (reference waymo.py:88-90: 640×960, 5 source views → 4·2·640·960 Gaussians): scale check
    "C6p": dict(num_points=4_915_200, width=960, height=640, sh_degree=4, profile="B"),
}
