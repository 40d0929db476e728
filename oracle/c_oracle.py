"""ctypes front-end for oracle/ggr_oracle.c — TEST INFRASTRUCTURE ONLY.

The C file is the sequential CPU restatement of the rasterizer GGRt calls at
``ggrt/model/pixelsplat/decoder/cuda_splatting.py:101-125`` of the reference
(parity unpinned — see the header of ggr_oracle.c).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
the product package ``ggrt_official_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libggr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ggr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libggr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ggo_preprocess.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


@dataclass
class OracleState:
    """Everything the oracle's forward produced (intermediates included, so that each HIP
    kernel can be checked stage by stage)."""
    P: int
    W: int
    H: int
    D: int
    M: int
    depth: np.ndarray
    radii: np.ndarray
    xy: np.ndarray
    conic_opacity: np.ndarray
    rgb: np.ndarray
    clamped: np.ndarray
    tiles_touched: np.ndarray
    cov3D: np.ndarray
    num_rendered: int
    point_list: np.ndarray
    keys_sorted: np.ndarray
    ranges: np.ndarray
    color: np.ndarray
    final_T: np.ndarray
    n_contrib: np.ndarray
    out_depth: np.ndarray
    inputs: dict


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy,
            sh_degree=0, shs=None, colors_precomp=None, cov3D_precomp=None, scales=None,
            rotations=None, scale_modifier=1.0, sh_cap=3, tight_rects=False) -> OracleState:
    """``sh_cap``: highest SH band evaluated (3 = graphdeco / w-depth family, the default; 4 = band 4 as well; see
    the header of ggr_oracle.c).  ``tight_rects``: False = the reference's tile rects (the restatement proper); True =
    the build's tight rects (ggr_oracle.c `tighten_rect`): sub-lists of the reference's, every output unchanged."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    opac = _f32(np.asarray(opacities).reshape(-1))
    shs = _f32(shs)
    colors_precomp = _f32(colors_precomp)
    cov3D_precomp = _f32(cov3D_precomp)
    scales = _f32(scales)
    rotations = _f32(rotations)
    V = _f32(np.asarray(viewmatrix).reshape(16))
    PM = _f32(np.asarray(projmatrix).reshape(16))
    cam = _f32(np.asarray(campos).reshape(3))
    bg = _f32(np.asarray(bg).reshape(3))
    assert (shs is None) != (colors_precomp is None)
    assert (cov3D_precomp is None) != (scales is None or rotations is None)
    M = 0 if shs is None else shs.shape[1]
    depth = np.zeros(P, np.float32)
    radii = np.zeros(P, np.int32)
    xy = np.zeros((P, 2), np.float32)
    co = np.zeros((P, 4), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    clamped = np.zeros((P, 3), np.uint8)
    tiles = np.zeros(P, np.int32)
    cov_used = np.zeros((P, 6), np.float32)
    rect = np.zeros((P, 4), np.int32)
    N = L.ggo_preprocess(
        C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(shs), _p(colors_precomp), _p(opac),
        _p(scales), _p(rotations), C.c_float(scale_modifier), _p(cov3D_precomp), _p(V), _p(PM), _p(cam),
        C.c_int(W), C.c_int(H), C.c_float(tanfovx), C.c_float(tanfovy), _p(depth), _p(radii), _p(xy),
        _p(co), _p(rgb), _p(clamped), _p(tiles), _p(cov_used), C.c_int(sh_cap), _p(rect), C.c_int(int(bool(tight_rects))))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    point_list = np.zeros(max(N, 1), np.uint32)
    keys = np.zeros(max(N, 1), np.uint64)
    ranges = np.zeros((gx * gy, 2), np.int32)
    L.ggo_bin(C.c_int(P), C.c_int(W), C.c_int(H), _p(depth), _p(radii), _p(xy), C.c_int64(N),
              _p(point_list), _p(keys), _p(ranges), _p(rect))
    color = np.zeros((3, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.int32)
    out_depth = np.zeros((H, W), np.float32)
    L.ggo_blend_forward(C.c_int(W), C.c_int(H), _p(ranges), _p(point_list), _p(xy), _p(co), _p(rgb),
                        _p(depth), _p(bg), _p(color), _p(final_T), _p(n_contrib), _p(out_depth))
    return OracleState(
        P=P, W=W, H=H, D=sh_degree, M=M, depth=depth, radii=radii, xy=xy, conic_opacity=co, rgb=rgb,
        clamped=clamped, tiles_touched=tiles, cov3D=cov_used, num_rendered=int(N),
        point_list=point_list[:N], keys_sorted=keys[:N], ranges=ranges, color=color, final_T=final_T,
        n_contrib=n_contrib, out_depth=out_depth,
        inputs=dict(means3D=means3D, opac=opac, shs=shs, colors_precomp=colors_precomp,
                    cov3D_precomp=cov3D_precomp, scales=scales, rotations=rotations,
                    scale_modifier=float(scale_modifier), V=V, PM=PM, cam=cam, bg=bg,
                    tanfovx=float(tanfovx), tanfovy=float(tanfovy), sh_cap=int(sh_cap)))


def backward(st: OracleState, dL_dcolor) -> dict:
    """Analytic backward (Appendix A.4) from the saved forward state.  Returns a dict with the
    gradients in the autograd order of the boundary."""
    L = lib()
    inp = st.inputs
    P, W, H = st.P, st.W, st.H
    dpix = _f32(np.asarray(dL_dcolor).reshape(3, H, W))
    d2 = np.zeros((P, 2), np.float64)
    dcon = np.zeros((P, 3), np.float64)
    dop = np.zeros(P, np.float64)
    drgb = np.zeros((P, 3), np.float64)
    L.ggo_blend_backward(C.c_int(P), C.c_int(W), C.c_int(H), _p(st.ranges), _p(st.point_list), _p(st.xy),
                         _p(st.conic_opacity), _p(st.rgb), _p(inp["bg"]), _p(st.final_T), _p(st.n_contrib),
                         _p(dpix), _p(d2), _p(dcon), _p(dop), _p(drgb))
    dmeans3D = np.zeros((P, 3), np.float32)
    dmeans2D = np.zeros((P, 3), np.float32)
    has_cp = inp["colors_precomp"] is not None
    dsh = None if has_cp else np.zeros((P, st.M, 3), np.float32)
    dcp = np.zeros((P, 3), np.float32) if has_cp else None
    dcov = np.zeros((P, 6), np.float32)
    has_sr = inp["scales"] is not None
    dsc = np.zeros((P, 3), np.float32) if has_sr else None
    drot = np.zeros((P, 4), np.float32) if has_sr else None
    L.ggo_preprocess_backward(
        C.c_int(P), C.c_int(st.D), C.c_int(st.M), _p(inp["means3D"]), _p(inp["shs"]), C.c_int(int(has_cp)),
        _p(inp["scales"]), _p(inp["rotations"]), C.c_float(inp["scale_modifier"]), _p(st.cov3D), _p(inp["V"]),
        _p(inp["PM"]), _p(inp["cam"]), C.c_int(W), C.c_int(H), C.c_float(inp["tanfovx"]),
        C.c_float(inp["tanfovy"]), _p(st.radii), _p(st.clamped), _p(d2), _p(dcon), _p(drgb), _p(dmeans3D),
        _p(dmeans2D), _p(dsh), _p(dcp), _p(dcov), _p(dsc), _p(drot), C.c_int(inp["sh_cap"]))
    return dict(means3D=dmeans3D, means2D=dmeans2D, shs=dsh, colors_precomp=dcp,
                opacities=dop.astype(np.float32).reshape(P, 1), cov3D_precomp=None if has_sr else dcov,
                scales=dsc, rotations=drot, _dL_dconic=dcon.astype(np.float32), _dL_drgb=drgb.astype(np.float32),
                _dL_dcov3D=dcov)
