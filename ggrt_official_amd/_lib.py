"""ctypes binding of the C ABI declared in ``include/ggr_raster.h``.

There is deliberately NO fallback: if ``libggr_raster.so`` is missing or does not export the
symbols the header declares, importing the rasterizer raises.  (The CPU restatements under
``oracle/`` are test infrastructure and are never reachable from here.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libggr_raster.so")
ABI_VERSION = 11

c_float_p = C.c_void_p  # device pointers travel as integers


class GgrSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32), ("sh_degree", C.c_int32),
        ("sh_stride", C.c_int32), ("num_points", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float), ("bg", C.c_void_p), ("viewmatrix", C.c_void_p),
        ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("tanfov_dev", C.c_void_p), ("sh_max_degree", C.c_int32), ("scissor", C.c_int32 * 4), ("reference_rects", C.c_int32),
        ("depth_sort", C.c_int32),
    ]


class GgrForwardIn(C.Structure):
    _fields_ = [
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p),
        ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("aux_precomp", C.c_void_p),
        ("input_scale", C.c_void_p), ("cov3D_full", C.c_int32), ("sh_channel_major", C.c_int32),
        ("aux_affine", C.c_int32), ("aux_a", C.c_float), ("aux_b", C.c_float),
    ]


class GgrViews(C.Structure):
    _fields_ = [
        ("num_views", C.c_int32), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("bg", C.c_void_p), ("tanfov", C.c_void_p), ("input_scale", C.c_void_p), ("num_sets", C.c_int32),
    ]


class GgrForwardOut(C.Structure):
    _fields_ = [
        ("out_color", C.c_void_p), ("radii", C.c_void_p), ("out_depth", C.c_void_p), ("geom_buffer", C.c_void_p),
        ("image_buffer", C.c_void_p), ("binning_buffer", C.c_void_p), ("num_rendered", C.c_int64),
        ("stage_ms", C.c_void_p), ("binning_capacity", C.c_int64), ("no_backward", C.c_int32),
        ("backward_scratch", C.c_void_p), ("capacity_is_hint", C.c_int32), ("max_list_len", C.c_int32),
        ("depth_sort_used", C.c_int32),
    ]


class GgrBackwardIn(C.Structure):
    _fields_ = [
        ("fwd", GgrForwardIn), ("radii", C.c_void_p), ("geom_buffer", C.c_void_p), ("image_buffer", C.c_void_p),
        ("binning_buffer", C.c_void_p), ("num_rendered", C.c_int64), ("dL_dout_color", C.c_void_p),
        ("dL_dout_depth", C.c_void_p), ("scratch", C.c_void_p), ("scratch_zeroed", C.c_int32),
    ]


class GgrBackwardOut(C.Structure):
    _fields_ = [
        ("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
        ("dL_dcolors_precomp", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dcov3D", C.c_void_p),
        ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p), ("dL_daux", C.c_void_p),
        ("dL_dviewmatrix", C.c_void_p),
        ("dL_dprojmatrix", C.c_void_p), ("dL_dcampos", C.c_void_p), ("stage_ms", C.c_void_p),
    ]

FWD_STAGES = ["preprocess", "depth_sort", "tile_count", "tile_scatter", "blend", "colour_side_stream", "tile_sort"]
DEPTH_SORT = {"auto": 0, "global": 1, "per_tile": 2, "global_3pass": 0x101}
DEPTH_SORT_NO_BUCKETS = 0x100      # IN flag: never the global sort's bucket form (include/ggr_raster.h)
DEPTH_SORT_FELL_BACK = 0x201       # OUT: the bucket form gave a frame up; its lists were built again in three passes
DEPTH_SORT_SLOW = 0x401            # OUT: the bucket form built the lists, the slow way (depths concentrated: include/ggr_raster.h)
BWD_STAGES = ["clear", "blend", "preprocess"]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

# every symbol include/ggr_raster.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ggr_abi_version", C.c_int, []),
    ("ggr_last_error", C.c_char_p, []),
    ("ggr_source_hash", C.c_char_p, []),
    ("ggr_geom_bytes", C.c_size_t, [C.c_int32]),
    ("ggr_image_bytes", C.c_size_t, [C.c_int32, C.c_int32]),
    ("ggr_binning_bytes", C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    ("ggr_work_bytes", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("ggr_backward_scratch_bytes", C.c_size_t, [C.c_int32]),
    ("ggr_forward", C.c_int, [C.POINTER(GgrSettings), C.POINTER(GgrForwardIn), C.POINTER(GgrForwardOut),
                              ALLOC_FN, C.c_void_p, C.c_void_p]),
    ("ggr_backward", C.c_int, [C.POINTER(GgrSettings), C.POINTER(GgrBackwardIn), C.POINTER(GgrBackwardOut),
                               C.c_void_p]),
    ("ggr_image_bytes_inference", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("ggr_geom_bytes_inference", C.c_size_t, [C.c_int32, C.c_int32]),
    ("ggr_geom_bytes_views", C.c_size_t, [C.c_int32, C.c_int32]),
    ("ggr_image_bytes_views", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("ggr_work_bytes_views", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("ggr_backward_scratch_bytes_views", C.c_size_t, [C.c_int32, C.c_int32]),
    ("ggr_forward_views", C.c_int, [C.POINTER(GgrSettings), C.POINTER(GgrViews), C.POINTER(GgrForwardIn),
                                    C.POINTER(GgrForwardOut), ALLOC_FN, C.c_void_p, C.c_void_p]),
    ("ggr_backward_views", C.c_int, [C.POINTER(GgrSettings), C.POINTER(GgrViews), C.POINTER(GgrBackwardIn),
                                     C.POINTER(GgrBackwardOut), C.c_void_p]),
    ("ggr_camera_setup", C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ggr_forward_status", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_void_p]),
    ("ggr_sort_stats_async", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    ("ggr_mark_visible", C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ggr_debug_readback_wait", C.c_int, [C.c_int32, C.c_double, C.POINTER(C.c_uint32)]),
    ("ggr_debug_counters", C.c_int, [C.POINTER(C.c_uint64), C.c_int32]),
    ("ggr_debug_host_slots", C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("ggr_debug_copy", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("ggr_debug_unpack_geom", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ggr_debug_unpack_binning", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
]

_lib = None


def load():
    """dlopen the HIP library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f"{LIB_PATH} does not export {name}")
        fn.restype = restype
        fn.argtypes = argtypes
    v = lib.ggr_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {v}, binding expects {ABI_VERSION}")
    # the library must have been built from the csrc/ tree next to it (a stale .so would otherwise be what gets tested
    # and measured).  GGR_SKIP_SOURCE_HASH=1: dev builds only (scripts/build_variants.sh ships several libraries).
    if os.environ.get("GGR_SKIP_SOURCE_HASH", "0") != "1":
        from . import _build
        built = lib.ggr_source_hash().decode("ascii", "replace")
        try:
            want = _build.source_hash()
        except FileNotFoundError:   # a binary-only installation (no csrc/ next to the library): nothing to compare with
            want = built
        if built != want:
            raise ImportError(
                f"{LIB_PATH} was built from other sources than {_build.CSRC} holds now (library {built[:12]}…, tree "
                f"{want[:12]}…): run `python __graft_entry__.py build`. There is no CPU fallback.")
    _lib = lib
    return lib


def last_error() -> str:
    return load().ggr_last_error().decode("utf-8", "replace")
