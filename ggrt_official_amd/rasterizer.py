"""PyTorch front-end of the MI355X rasterizer — same surface as ``diff_gaussian_rasterization``.

Mirrors what GGRt imports and calls at
``ggrt/model/pixelsplat/decoder/cuda_splatting.py:6-9,101-125`` of the reference:

    settings   = GaussianRasterizationSettings(image_height=…, image_width=…, tanfovx=…, tanfovy=…,
                     bg=…, scale_modifier=…, viewmatrix=…, projmatrix=…, sh_degree=…, campos=…,
                     prefiltered=False)                      # `debug` optional, as at :101-113
    rasterizer = GaussianRasterizer(settings)
    image, radii, depth = rasterizer(means3D=…, means2D=…, shs=… | colors_precomp=…, opacities=…,
                                     cov3D_precomp=… | scales=…, rotations=…)

Same names, argument meaning, error behaviour (``Exception`` with the upstream messages when both /
neither of ``shs``/``colors_precomp`` or of ``scales+rotations``/``cov3D_precomp`` are given) and
gradient order.  The arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/ggr_raster.h``; PyTorch only provides device memory, the stream and autograd plumbing.
There is no CPU path: tensors must live on a ROCm device and the HIP library must be built.

Extension beyond the reference: if ``viewmatrix`` / ``projmatrix`` / ``campos`` of
the settings require grad, their gradients are produced too (the reference's extension cannot:
it receives them inside a NamedTuple).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool = False
    list_capacity: int = 0  # extension.  0: exact mode (one 4-byte read-back + host sync per forward, like
    #                         upstream).  > 0: SYNC-FREE mode — the per-tile lists go into a buffer of this many
    #                         entries, nothing is read back, forward + backward are hipGraph-capturable; check
    #                         `last_forward_status()` (count, overflow) whenever a sync is affordable
    # ---- input forms (extensions; defaults = upstream's forms).  They move the torch operations the reference's
    # render_cuda runs over the P-sized tensors before every call (cuda_splatting.py:66-77,116,124) into the
    # kernels' loads, see include/ggr_raster.h ----
    input_scale: Optional[torch.Tensor] = None  # device scalar s: means·s, cov·s², scales·s (the 1/near renorm)
    sh_channel_major: bool = False              # shs given as [P,3,M] (GGRt's harmonics layout) instead of [P,M,3]
    aux_affine: Optional[tuple] = None          # (a, b): depth output = Σ max(a + b·z/s, 0)·α·T (GGRt's depth pass)
    tanfov: Optional[torch.Tensor] = None       # device [2]: overrides tanfovx / tanfovy without a read-back (camera_setup)
    scissor: Optional[tuple] = None  # (x0, y0, x1, y1) pixels, half-open: only the tiles overlapping the window are binned
    #                         and blended (the fine-tune loop's per-cell re-render, finetune_ggrt_stable.py:126-142); inside
    #                         them the outputs equal the full-frame render bit for bit, other tiles come out as background
    reference_rects: bool = False    # False: tight tile rects (a Gaussian is listed only where its α ≥ 1/255 ellipse can reach:
    #                         same outputs bit for bit, shorter internal lists); True: the reference's rects — lists identical
    #                         to the reference's, entry for entry
    sh_max_degree: int = 0  # highest SH band evaluated — an EXPLICIT choice where it matters.
    #                         3: graphdeco and its w-depth forks — the family the live call site's signature belongs to
    #                         (3-tuple return, no `debug`: cuda_splatting.py:101-118) — ignore coefficients 16.. (zero
    #                         gradient); 4: the nine degree-4 terms are evaluated when sh_degree >= 4 with >= 25
    #                         coefficients.  0 = not chosen (also settable process-wide: GGR_SH_MAX_DEGREE=3|4): bands
    #                         0..3, and the first call that thereby leaves coefficients 16.. unused warns once
    depth_sort: str = "auto"  # how the tile lists get their (depth, index) order — identical lists either way
    #                         (include/ggr_raster.h GgrSettings.depth_sort): "global" = one depth sort of the Gaussians in front of
    #                         the tile-list build; "per_tile" = lists built in index order, then every tile's list sorted by
    #                         depth in LDS (no global dependency on the forward's critical path); "auto" = per tile for
    #                         frames with <= 256 Gaussians per tile on average and (once known) a longest list <= 4096,
    #                         outside the sync-free mode; global otherwise (GGRt's 660-tile frames).  The global sort runs in
    #                         its BUCKET form where it can (ABI 11: one partition pass + every bucket sorted in LDS; falls
    #                         back inside the call); "global_3pass" keeps the three radix passes


class StageProfile:
    """Per-stage HIP-event timings (ms, accumulated over calls) filled by the library when a
    ``profile_stages()`` context is active.  Profiling synchronises the stream — never use it inside a
    timed region."""

    def __init__(self):
        self.fwd = (C.c_float * len(_lib.FWD_STAGES))()
        self.bwd = (C.c_float * len(_lib.BWD_STAGES))()
        self.fwd_calls = 0
        self.bwd_calls = 0

    def as_dict(self):
        d = {f"fwd_{n}_ms": self.fwd[i] / max(self.fwd_calls, 1) for i, n in enumerate(_lib.FWD_STAGES)}
        d.update({f"bwd_{n}_ms": self.bwd[i] / max(self.bwd_calls, 1) for i, n in enumerate(_lib.BWD_STAGES)})
        return d


_tls = threading.local()


@contextlib.contextmanager
def profile_stages():
    prof = StageProfile()
    prev = getattr(_tls, "prof", None)
    _tls.prof = prof
    _active_profiles.append(prof)
    try:
        yield prof
    finally:
        _tls.prof = prev
        _active_profiles.remove(prof)


_active_profiles: list = []


def _current_profile():
    # autograd runs backward on a worker thread: fall back to the most recent active profile
    prof = getattr(_tls, "prof", None)
    if prof is None and _active_profiles:
        prof = _active_profiles[-1]
    return prof


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _none_if_empty(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor) and t.numel() == 0 and t.dim() <= 1:
        return None
    return t


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {_lib.last_error()}")


# ---- the default mode's list buffer, sized from what the same shape needed before --------------------------------
# Upstream's forward reads num_rendered back in the middle, allocates the list buffer to that size and only then
# launches the rest — the device idles for as long as the host takes (a slow or busy host: 0.96 instead of 0.88 ms per
# 1080p frame on one box of the pool).  With a guess of the size — 1.25 × the largest num_rendered of the last calls of
# the same (device, P, W, H, views) — everything is enqueued at once (GgrForwardOut.capacity_is_hint), the call still
# returns the exact num_rendered, and a guess that was too small costs a repeat in upstream's order.  The guess is a
# HINT: no result depends on it; `GGR_LIST_HINT=0` in the environment turns it off.
_GGR_E_CAPACITY = 5
_hint_lock = threading.Lock()
_hints: dict = {}
_HINTS_ON = os.environ.get("GGR_LIST_HINT", "1") != "0"


def set_list_hint(enabled: bool) -> bool:
    """Turns the list-size guess of the default mode on or off (process-wide); returns the previous setting.  Off, every
    forward runs in upstream's order: read num_rendered back, allocate, launch the rest."""
    global _HINTS_ON
    with _hint_lock:
        prev, _HINTS_ON = _HINTS_ON, bool(enabled)
        if not enabled:
            _hints.clear()
            _sort_watch.clear()
    return prev


# what the default mode did with its guesses (process-wide): calls that ran with a guessed list buffer that held
# ("hinted"), whose guess did not hold — repaired inside the call, or repeated in upstream's order ("missed") —, and that ran in upstream's order from the start
# ("exact": first call of a shape, hints off, stage profiling)
_hint_stats = {"hinted": 0, "missed": 0, "exact": 0}


def list_hint_stats(reset: bool = False) -> dict:
    with _hint_lock:
        out = dict(_hint_stats)
        if reset:
            for k in _hint_stats:
                _hint_stats[k] = 0
    return out


def clear_list_hints() -> None:
    """Forgets every shape's list-size history (the next forward of each shape runs in upstream's order)."""
    with _hint_lock:
        _hints.clear()
        _sort_watch.clear()


def _scissor_key(rs):
    sc = getattr(rs, "scissor", None)
    return tuple(int(v) for v in sc) if sc else None


def _capacity_guess(key):
    """(list entries, longest tile list) to plan for: 1.25 × the largest of the last calls of the same shape; (0, 0) = no idea"""
    if not _HINTS_ON:
        return 0, 0
    with _hint_lock:
        h = _hints.get(key)
        if not h:
            return 0, 0
        return int(1.25 * max(n for n, _ in h)) + 4096, int(1.25 * max(l for _, l in h))


def _note_rendered(key, n: int, longest: int = 0):
    with _hint_lock:
        if len(_hints) > 256 and key not in _hints:   # (shapes that keep changing: start over rather than grow)
            _hints.clear()
        h = _hints.setdefault(key, [])
        h.append((int(n), max(int(longest), 0)))
        del h[:-8]


def _forward_with_guess(call, fout, holder, lib, dev, W, H, key, user_capacity, profiling):
    """Runs `call()` (the ggr_forward / ggr_forward_views invocation over `fout`): sync-free with the caller's
    `list_capacity`, or exact — with a guessed list buffer when this shape has been seen, upstream's order otherwise and
    after a guess that did not hold.  Returns what `last_forward_status` needs to know."""
    if user_capacity > 0:  # sync-free mode: bring the list buffer, no read-back inside the call
        holder["bin"] = torch.empty((lib.ggr_binning_bytes(user_capacity, W, H),), dtype=torch.uint8, device=dev)
        fout.binning_buffer = holder["bin"].data_ptr()
        fout.binning_capacity = user_capacity
        _check(call(), "ggr_forward")
        return
    # (stage timing keeps upstream's order: the read-back is a stage)
    guess, len_guess = (0, 0) if profiling else _capacity_guess(key)
    if guess > 0:
        holder["bin"] = torch.empty((lib.ggr_binning_bytes(guess, W, H),), dtype=torch.uint8, device=dev)
        fout.binning_buffer = holder["bin"].data_ptr()
        fout.binning_capacity = guess
        fout.capacity_is_hint = 1
        fout.max_list_len = len_guess
        rc = call()
        if rc != _GGR_E_CAPACITY:
            _check(rc, "ggr_forward")
            _note_rendered(key, int(fout.num_rendered), int(fout.max_list_len))
            # capacity_is_hint == 2: the guess did not hold and the call repaired itself (exact buffer through the allocator —
            # `holder["bin"]` is that buffer now —, scatter and blend once more: csrc/api.hip); == 3: the list buffer held, the
            # guess of the longest tile list did not (the long lists were sorted and the frame blended once more)
            _hint_stats["missed" if int(fout.capacity_is_hint) >= 2 else "hinted"] += 1
            return
        _hint_stats["missed"] += 1
        # the guess did not hold: once more, in upstream's order (the buffers of the first attempt are released to the
        # stream-ordered allocator: what is still running on them was enqueued before what follows)
        _note_rendered(key, int(fout.num_rendered), int(fout.max_list_len))
        holder.clear()
        fout.binning_buffer = None
        fout.binning_capacity = 0
        fout.capacity_is_hint = 0
    fout.max_list_len = 0
    _check(call(), "ggr_forward")
    _note_rendered(key, int(fout.num_rendered), int(fout.max_list_len))
    if guess == 0:
        _hint_stats["exact"] += 1


# ---- which depth sort "auto" means for a shape ----------------------------------------------------------------------------------
# The library's AUTO picks the per-tile sort from sizes alone (csrc/api.hip).  What it cannot see is how the depths are
# spread: tiles whose depth keys cluster in few buckets (a surface in front of a background) take the sort kernel's slow route,
# and a frame made of such tiles renders faster through the global sort (NOTES r6 "clustered depths": forward 0.46-0.53 against
# 0.39 ms at C3's sizes).  The sort kernel counts the entries of those tiles; the host looks at the count now and then —
# ggr_sort_stats_async: a 16-byte copy queued behind the forward, read at a LATER call of the shape, never waited for — and
# keeps a shape whose frames are mostly of that kind on the global sort for a while.  A hint like the list sizes: no result
# depends on it.  Off with GGR_LIST_HINT=0 / set_list_hint(False), with an explicit depth_sort, or with GGR_DEPTH_SORT set.
_SORT_LOOK_EVERY = 64      # after a shape's first two per-tile forwards: every 64th
_SORT_KEEP_GLOBAL = 256    # calls for which a shape then stays on the global sort before the per-tile one is tried again
_SORT_SLOW_SHARE = 0.30    # share of a frame's list entries in slow-route tiles from which the global sort is chosen
_sort_watch: dict = {}


def _sort_choice(key, rs, st) -> None:
    """Before a forward: reads a finished look at an earlier frame of this shape, and turns AUTO into GLOBAL while the shape
    is known to cluster."""
    if not _HINTS_ON or st.depth_sort != _lib.DEPTH_SORT["auto"] or os.environ.get("GGR_DEPTH_SORT"):
        return
    with _hint_lock:
        if len(_sort_watch) > 256 and key not in _sort_watch:
            _sort_watch.clear()
        w = _sort_watch.setdefault(key, {"calls": 0, "global_until": 0, "pending": None, "words": None, "slow_share": None})
        pend = w["pending"]
        if pend is not None and pend[0].query():
            w["pending"] = None
            n, slow = int(w["words"][0]), int(w["words"][3])
            if n > 0:
                w["slow_share"] = slow / n
                if slow > _SORT_SLOW_SHARE * n:
                    w["global_until"] = w["calls"] + _SORT_KEEP_GLOBAL
        w["calls"] += 1
        if w["calls"] <= w["global_until"]:
            st.depth_sort = _lib.DEPTH_SORT["global"]


def _sort_no_buckets(key, st) -> None:
    """Before a forward: a shape whose frames the global sort's bucket form recently gave up keeps the three passes for a while."""
    if not _HINTS_ON or os.environ.get("GGR_GLOBAL_SORT"):
        return
    with _hint_lock:
        w = _sort_watch.get(key)
        if w is not None and w.get("no_buckets_left", 0) > 0:
            w["no_buckets_left"] -= 1
            st.depth_sort |= _lib.DEPTH_SORT_NO_BUCKETS


def _sort_fell_back(key, fout) -> None:
    """Behind a forward: the global sort's bucket form gave this frame up (a bucket of > 8192 different keys inside 1/4096 of
    the frame's depth range) and the call built the lists again in three passes — about one binning more.  Frames of a shape
    tend to look alike: the shape keeps the three passes for a while.  A hint like the others: no result depends on it."""
    if int(fout.depth_sort_used) not in (_lib.DEPTH_SORT_FELL_BACK, _lib.DEPTH_SORT_SLOW) or not _HINTS_ON:
        return
    with _hint_lock:
        w = _sort_watch.setdefault(key, {"calls": 0, "global_until": 0, "pending": None, "words": None, "slow_share": None})
        w["no_buckets_left"] = _SORT_KEEP_GLOBAL
        w["fell_back"] = w.get("fell_back", 0) + 1


def _sort_look(key, lib, fout, geom, rows: int, stream: int) -> None:
    """Behind a forward that sorted per tile: now and then, the 16-byte copy of its counters (read by a later `_sort_choice`)."""
    if int(fout.depth_sort_used) != _lib.DEPTH_SORT["per_tile"]:
        return
    with _hint_lock:
        w = _sort_watch.get(key)
        if w is None or w["pending"] is not None or (w["calls"] > 2 and w["calls"] % _SORT_LOOK_EVERY):
            return
        if torch.cuda.is_current_stream_capturing():
            return
        if w["words"] is None:
            w["words"] = torch.zeros(4, dtype=torch.int32).pin_memory()
        _check(lib.ggr_sort_stats_async(geom.data_ptr(), rows, w["words"].data_ptr(), stream), "ggr_sort_stats_async")
        ev = torch.cuda.Event()
        ev.record()
        w["pending"] = (ev, geom)   # (the buffer stays referenced until the copy has run)


def sort_watch_stats() -> dict:
    """{shape key: (calls, share of the last looked-at frame's entries in slow-route tiles or None, calls left on the global sort)}"""
    with _hint_lock:
        return {k: (w["calls"], w["slow_share"], max(0, w["global_until"] - w["calls"])) for k, w in _sort_watch.items()}


_sh_warned = False


def _sh_cap(rs, M: int) -> int:
    """GgrSettings.sh_max_degree for this call: the settings' explicit 3 / 4, else GGR_SH_MAX_DEGREE, else 0 (= bands
    0..3) — and ONE warning per process when that undecided default leaves coefficients unused (GGRt passes
    sh_degree 4 with 25 coefficients per channel; what the replaced extension does with band 4 is not verifiable from
    the reference tree)."""
    global _sh_warned
    cap = int(getattr(rs, "sh_max_degree", 0) or 0)
    if cap == 0:
        cap = int(os.environ.get("GGR_SH_MAX_DEGREE", "0") or 0)
    if cap not in (0, 3, 4):
        raise RuntimeError("sh_max_degree must be 3 or 4 (0 = not chosen)")
    if cap == 0 and not _sh_warned and int(rs.sh_degree) >= 4 and M >= 25:
        _sh_warned = True
        import warnings
        warnings.warn(
            "ggrt_official_amd: sh_degree >= 4 with >= 25 SH coefficients per channel, and sh_max_degree was not chosen: "
            "bands 0..3 are evaluated, coefficients 16.. are ignored and get zero gradient (the behaviour of the "
            "rasterizer family GGRt's live call site is written against).  If the extension you are replacing evaluates "
            "band 4, pass sh_max_degree=4 (GaussianRasterizationSettings / DecoderSplattingCUDA / GGR_SH_MAX_DEGREE=4); "
            "pass 3 to keep this behaviour silently.  See INTEGRATION.md §7 and scripts/export_upstream_goldens.py.",
            UserWarning, stacklevel=3)
    return cap


def _depth_sort(rs) -> int:
    v = getattr(rs, "depth_sort", "auto") or "auto"
    if v not in _lib.DEPTH_SORT:
        raise RuntimeError(f"depth_sort must be one of {sorted(_lib.DEPTH_SORT)}, not {v!r}")
    return _lib.DEPTH_SORT[v]


def _settings_struct(rs: GaussianRasterizationSettings, P: int, M: int, bg, view, proj, campos) -> _lib.GgrSettings:
    tf = getattr(rs, "tanfov", None)
    if tf is not None and (tf.dtype != torch.float32 or not tf.is_contiguous() or (bg is not None and tf.device != bg.device)):
        raise RuntimeError("settings.tanfov must be a contiguous float32 [2] tensor on the rasterizer's device")
    return _lib.GgrSettings(
        image_height=int(rs.image_height), image_width=int(rs.image_width), sh_degree=int(rs.sh_degree),
        sh_stride=int(M), num_points=int(P), tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
        scale_modifier=float(rs.scale_modifier), bg=_ptr(bg), viewmatrix=_ptr(view), projmatrix=_ptr(proj),
        campos=_ptr(campos), prefiltered=int(bool(rs.prefiltered)), debug=int(bool(rs.debug)), tanfov_dev=_ptr(tf),
        sh_max_degree=_sh_cap(rs, int(M)),
        scissor=(C.c_int32 * 4)(*[int(v) for v in (getattr(rs, "scissor", None) or (0, 0, 0, 0))]),
        reference_rects=int(bool(getattr(rs, "reference_rects", False))),
        depth_sort=_depth_sort(rs))


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                viewmatrix, projmatrix, campos, aux, raster_settings, grad_mode=True):
        lib = _lib.load()
        rs = raster_settings
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError(
                "ggrt_official_amd rasterizer needs tensors on a ROCm GPU (device 'cuda'); there is no CPU path")
        ctx.set_materialize_grads(False)
        means3D_c = _f32c(means3D)
        P = means3D_c.shape[0]
        sh_c, cp_c = _f32c(sh), _f32c(colors_precomp)
        op_c = _f32c(opacities)
        sc_c, rot_c, cov_c = _f32c(scales), _f32c(rotations), _f32c(cov3Ds_precomp)
        aux_c = _f32c(aux)
        sh_cm = bool(getattr(rs, "sh_channel_major", False)) and sh_c is not None
        if sh_cm and (sh_c.dim() != 3 or sh_c.shape[1] != 3):
            raise RuntimeError("sh_channel_major expects shs of shape [P,3,M]")
        M = 0 if sh_c is None else int(sh_c.shape[2] if sh_cm else sh_c.shape[1])
        cov_full = cov_c is not None and cov_c.dim() == 3  # [P,3,3]: the upper triangle is gathered on load
        in_scale = getattr(rs, "input_scale", None)
        in_scale = None if in_scale is None else _f32c(in_scale.to(dev)).reshape(-1)[:1]
        aux_aff = getattr(rs, "aux_affine", None) if aux_c is None else None
        form = dict(input_scale=_ptr(in_scale), cov3D_full=int(cov_full), sh_channel_major=int(sh_cm),
                    aux_affine=int(aux_aff is not None), aux_a=float(aux_aff[0]) if aux_aff else 0.0,
                    aux_b=float(aux_aff[1]) if aux_aff else 0.0)
        bg = _f32c(rs.bg.to(dev))
        view, proj, cam = _f32c(viewmatrix.to(dev)), _f32c(projmatrix.to(dev)), _f32c(campos.to(dev))
        H, W = int(rs.image_height), int(rs.image_width)

        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            depth = torch.empty((H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            # nothing requires grad (torch.no_grad() / inference): no backward will replay this forward, so the
            # per-pixel checkpoints of the segmented backward are neither written nor allocated
            # (needs_input_grad mirrors tensor.requires_grad even under no_grad — the call site's means2D sink always
            #  requires grad — and inside a Function's forward the grad mode is always off: the caller's grad mode
            #  comes in as an argument)
            infer = (not grad_mode) or not any(getattr(ctx, "needs_input_grad", (True,)))  # (debug_forward_state: plain ctx)
            # (… nor is the SH colour's Jacobian — 48 B per Gaussian of the geometry buffer — written or allocated)
            geom = torch.empty((lib.ggr_geom_bytes_inference(P, 1) if infer else lib.ggr_geom_bytes(P),), dtype=torch.uint8,
                               device=dev)
            img = torch.empty((lib.ggr_image_bytes_inference(W, H, 1) if infer else lib.ggr_image_bytes(W, H),),
                              dtype=torch.uint8, device=dev)
            holder = {}

            def _alloc(_ctx, nbytes):
                # called twice (include/ggr_raster.h): 1st = transient work area, 2nd = tile lists (kept)
                try:
                    key = "work" if "work" not in holder else "bin"
                    holder[key] = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
                    return holder[key].data_ptr()
                except Exception:  # pragma: no cover - out of memory
                    return None

            cb = _lib.ALLOC_FN(_alloc)
            st = _settings_struct(rs, P, M, bg, view, proj, cam)
            fin = _lib.GgrForwardIn(means3D=_ptr(means3D_c), shs=_ptr(sh_c), colors_precomp=_ptr(cp_c),
                                    opacities=_ptr(op_c), scales=_ptr(sc_c), rotations=_ptr(rot_c),
                                    cov3D_precomp=_ptr(cov_c), aux_precomp=_ptr(aux_c), **form)
            # the backward's per-Gaussian gradient records: cleared by the forward on the side (inside its blend kernel)
            scratch = None if infer else torch.empty((lib.ggr_backward_scratch_bytes(P),), dtype=torch.uint8, device=dev)
            fout = _lib.GgrForwardOut(out_color=color.data_ptr(), radii=_ptr(radii), out_depth=depth.data_ptr(),
                                      geom_buffer=geom.data_ptr(), image_buffer=img.data_ptr(),
                                      binning_buffer=None, num_rendered=0, stage_ms=None, binning_capacity=0,
                                      no_backward=int(infer), backward_scratch=_ptr(scratch))
            capacity = int(getattr(rs, "list_capacity", 0) or 0)
            prof = _current_profile()
            if prof is not None:
                fout.stage_ms = C.cast(prof.fwd, C.c_void_p)
                prof.fwd_calls += 1
            key = (dev.index, P, W, H, 1, _scissor_key(rs))
            _sort_choice(key, rs, st)
            _sort_no_buckets(key, st)
            _forward_with_guess(lambda: lib.ggr_forward(C.byref(st), C.byref(fin), C.byref(fout), cb, None, stream),
                                fout, holder, lib, dev, W, H, key, capacity, prof is not None)
            _sort_look(key, lib, fout, geom, P, stream)
            _sort_fell_back(key, fout)

        # exact mode: count known, nothing to keep.  Sync-free mode: count + flags live in the geometry buffer on the
        # device, so that (≈100 MB at P = 1 M) buffer stays referenced until this thread's next forward
        _tls.last_forward = (geom, P) if capacity > 0 else (None, int(fout.num_rendered))
        _tls.last_binning = (int(fout.depth_sort_used), int(fout.max_list_len))
        ctx.raster_settings = rs
        ctx.num_rendered = int(fout.num_rendered)
        ctx.dims = (P, M, H, W)
        ctx.form = (form, in_scale, cov_full, sh_cm)  # in_scale kept alive for backward
        ctx.in_shapes = (means3D.shape, None if sh is None else sh.shape, opacities.shape,
                         None if aux is None else aux.shape)
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
        ctx.save_for_backward(means3D_c, sh_c, cp_c, op_c, sc_c, rot_c, cov_c, bg, view, proj, cam, radii, geom,
                              img, holder.get("bin"), aux_c, scratch)
        ctx.scratch_fresh = scratch is not None  # (a second backward over this forward clears a scratch of its own)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_depth):
        lib = _lib.load()
        rs = ctx.raster_settings
        (means3D, sh, cp, op, sc, rot, cov, bg, view, proj, cam, radii, geom, img, binb, aux, fwd_scratch) = ctx.saved_tensors
        P, M, H, W = ctx.dims
        dev = means3D.device
        need_pose = any(ctx.needs_input_grad[8:11])
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if grad_color is None:
                grad_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
            grad_color = _f32c(grad_color)
            grad_depth = _f32c(grad_depth)
            d_means3D = torch.empty((P, 3), dtype=torch.float32, device=dev)
            d_means2D = torch.empty((P, 3), dtype=torch.float32, device=dev)
            d_op = torch.empty((P,), dtype=torch.float32, device=dev)
            form, _in_scale, cov_full, sh_cm = ctx.form
            d_cov = torch.empty((P, 3, 3) if (cov_full and cov is not None) else (P, 6), dtype=torch.float32, device=dev)
            d_sh = (torch.empty((P, 3, M) if sh_cm else (P, M, 3), dtype=torch.float32, device=dev)
                    if sh is not None else None)
            d_cp = torch.empty((P, 3), dtype=torch.float32, device=dev) if cp is not None else None
            d_sc = torch.empty((P, 3), dtype=torch.float32, device=dev) if sc is not None else None
            d_rot = torch.empty((P, 4), dtype=torch.float32, device=dev) if rot is not None else None
            d_aux = torch.empty((P,), dtype=torch.float32, device=dev) if (aux is not None and grad_depth is not None) else None
            d_view = torch.empty((4, 4), dtype=torch.float32, device=dev) if need_pose else None
            d_proj = torch.empty((4, 4), dtype=torch.float32, device=dev) if need_pose else None
            d_cam = torch.empty((3,), dtype=torch.float32, device=dev) if need_pose else None
            zeroed = bool(getattr(ctx, "scratch_fresh", False)) and fwd_scratch is not None
            ctx.scratch_fresh = False
            scratch = fwd_scratch if zeroed else torch.empty((lib.ggr_backward_scratch_bytes(P),), dtype=torch.uint8, device=dev)

            st = _settings_struct(rs, P, M, bg, view, proj, cam)
            bin_ = _lib.GgrBackwardIn(
                fwd=_lib.GgrForwardIn(means3D=_ptr(means3D), shs=_ptr(sh), colors_precomp=_ptr(cp), opacities=_ptr(op),
                                      scales=_ptr(sc), rotations=_ptr(rot), cov3D_precomp=_ptr(cov),
                                      aux_precomp=_ptr(aux), **form),
                radii=_ptr(radii), geom_buffer=geom.data_ptr(), image_buffer=img.data_ptr(),
                binning_buffer=_ptr(binb), num_rendered=ctx.num_rendered, dL_dout_color=grad_color.data_ptr(),
                dL_dout_depth=_ptr(grad_depth), scratch=scratch.data_ptr(), scratch_zeroed=int(zeroed))
            bout = _lib.GgrBackwardOut(
                dL_dmeans3D=d_means3D.data_ptr(), dL_dmeans2D=d_means2D.data_ptr(), dL_dshs=_ptr(d_sh),
                dL_dcolors_precomp=_ptr(d_cp), dL_dopacities=d_op.data_ptr(), dL_dcov3D=d_cov.data_ptr(),
                dL_dscales=_ptr(d_sc), dL_drotations=_ptr(d_rot), dL_daux=_ptr(d_aux), dL_dviewmatrix=_ptr(d_view),
                dL_dprojmatrix=_ptr(d_proj), dL_dcampos=_ptr(d_cam), stage_ms=None)
            prof = _current_profile()
            if prof is not None:
                bout.stage_ms = C.cast(prof.bwd, C.c_void_p)
                prof.bwd_calls += 1
            _check(lib.ggr_backward(C.byref(st), C.byref(bin_), C.byref(bout), stream), "ggr_backward")

        means_shape, sh_shape, op_shape, aux_shape = ctx.in_shapes
        has_sh, has_cp, has_sc, has_cov = ctx.has
        return (
            d_means3D.reshape(means_shape),
            d_means2D,
            d_sh.reshape(sh_shape) if has_sh else None,
            d_cp if has_cp else None,
            d_op.reshape(op_shape),
            d_sc if has_sc else None,
            d_rot if has_sc else None,
            d_cov if has_cov else None,
            d_view if ctx.needs_input_grad[8] else None,
            d_proj if ctx.needs_input_grad[9] else None,
            d_cam.reshape(ctx.saved_tensors[10].shape) if ctx.needs_input_grad[10] else None,
            d_aux.reshape(aux_shape) if d_aux is not None else None,
            None, None,
        )


class _RasterizeViews(torch.autograd.Function):
    """V views of the SAME Gaussians in one launch set (``ggr_forward_views`` / ``ggr_backward_views``)."""

    @staticmethod
    def forward(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrices,
                projmatrices, campos, aux, means2D, raster_settings, bg, tanfov, input_scale, grad_mode=True):
        lib = _lib.load()
        rs = raster_settings
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError(
                "ggrt_official_amd rasterizer needs tensors on a ROCm GPU (device 'cuda'); there is no CPU path")
        ctx.set_materialize_grads(False)
        means3D_c = _f32c(means3D)
        view, proj, cam = _f32c(viewmatrices.to(dev)), _f32c(projmatrices.to(dev)), _f32c(campos.to(dev))
        V = int(view.shape[0])
        if view.shape != (V, 4, 4) or proj.shape != (V, 4, 4) or cam.shape != (V, 3):
            raise RuntimeError("rasterize_views: viewmatrices / projmatrices [V,4,4] and campos [V,3] expected")
        bg_c, tf_c = _f32c(bg.to(dev)).reshape(V, 3), _f32c(tanfov.to(dev)).reshape(V, 2)
        sc_in = None if input_scale is None else _f32c(input_scale.to(dev)).reshape(V)
        sh_c, cp_c, op_c = _f32c(sh), _f32c(colors_precomp), _f32c(opacities)
        sc_c, rot_c, cov_c = _f32c(scales), _f32c(rotations), _f32c(cov3Ds_precomp)
        aux_c = _f32c(aux)
        # several Gaussian SETS (GgrViews.num_sets): means3D [B,P,3] and every per-Gaussian input [B,P,…]; the V views
        # are B groups of V/B consecutive views.  Inside, the sets are rows [b·P, (b+1)·P) of flat [B·P, …] arrays.
        B = 1
        if means3D_c.dim() == 3:
            B = int(means3D_c.shape[0])
            if B < 1 or V % B != 0:
                raise RuntimeError(f"rasterize_views: {V} views cannot be split over {B} Gaussian sets")
            flat = lambda t: None if t is None else t.reshape(t.shape[0] * t.shape[1], *t.shape[2:])
            if any(t is not None and (t.dim() < 2 or t.shape[0] != B or t.shape[1] != means3D_c.shape[1])
                   for t in (sh_c, cp_c, op_c, sc_c, rot_c, cov_c)):
                raise RuntimeError("rasterize_views: with means3D [B,P,3] every per-Gaussian input must be [B,P,…]")
            means3D_c, sh_c, cp_c, op_c, sc_c, rot_c, cov_c = (flat(t) for t in (means3D_c, sh_c, cp_c, op_c, sc_c,
                                                                                   rot_c, cov_c))
        P = means3D_c.shape[0] // B   # Gaussians per set
        if aux_c is not None and aux_c.numel() != V * P:
            raise RuntimeError("rasterize_views: aux_precomp must be [V,P]")
        sh_cm = bool(getattr(rs, "sh_channel_major", False)) and sh_c is not None
        if sh_cm and (sh_c.dim() != 3 or sh_c.shape[1] != 3):
            raise RuntimeError("sh_channel_major expects shs of shape [P,3,M]")
        M = 0 if sh_c is None else int(sh_c.shape[2] if sh_cm else sh_c.shape[1])
        cov_full = cov_c is not None and cov_c.dim() == 3
        aux_aff = getattr(rs, "aux_affine", None) if aux_c is None else None
        form = dict(input_scale=None, cov3D_full=int(cov_full), sh_channel_major=int(sh_cm),
                    aux_affine=int(aux_aff is not None), aux_a=float(aux_aff[0]) if aux_aff else 0.0,
                    aux_b=float(aux_aff[1]) if aux_aff else 0.0)
        H, W = int(rs.image_height), int(rs.image_width)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            color = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
            depth = torch.empty((V, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((V, P), dtype=torch.int32, device=dev)
            infer = (not grad_mode) or not any(ctx.needs_input_grad)
            geom = torch.empty((lib.ggr_geom_bytes_inference(P, V) if infer else lib.ggr_geom_bytes_views(P, V),),
                               dtype=torch.uint8, device=dev)
            img = torch.empty((lib.ggr_image_bytes_inference(W, H, V) if infer else lib.ggr_image_bytes_views(W, H, V),),
                              dtype=torch.uint8, device=dev)
            holder = {}

            def _alloc(_ctx, nbytes):
                try:
                    key = "work" if "work" not in holder else "bin"
                    holder[key] = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
                    return holder[key].data_ptr()
                except Exception:  # pragma: no cover - out of memory
                    return None

            cb = _lib.ALLOC_FN(_alloc)
            st = _settings_struct(rs._replace(tanfovx=0.0, tanfovy=0.0, tanfov=None), P, M, None, None, None, None)
            vw = _lib.GgrViews(num_views=V, viewmatrix=view.data_ptr(), projmatrix=proj.data_ptr(), campos=cam.data_ptr(),
                               bg=bg_c.data_ptr(), tanfov=tf_c.data_ptr(), input_scale=_ptr(sc_in), num_sets=B)
            fin = _lib.GgrForwardIn(means3D=_ptr(means3D_c), shs=_ptr(sh_c), colors_precomp=_ptr(cp_c),
                                    opacities=_ptr(op_c), scales=_ptr(sc_c), rotations=_ptr(rot_c),
                                    cov3D_precomp=_ptr(cov_c), aux_precomp=_ptr(aux_c), **form)
            scratch = None if infer else torch.empty((lib.ggr_backward_scratch_bytes_views(P, V),), dtype=torch.uint8,
                                                     device=dev)
            fout = _lib.GgrForwardOut(out_color=color.data_ptr(), radii=_ptr(radii), out_depth=depth.data_ptr(),
                                      geom_buffer=geom.data_ptr(), image_buffer=img.data_ptr(),
                                      binning_buffer=None, num_rendered=0, stage_ms=None, binning_capacity=0,
                                      no_backward=int(infer), backward_scratch=_ptr(scratch))
            capacity = int(getattr(rs, "list_capacity", 0) or 0)   # (sync-free mode: the capacity covers the lists of ALL views)
            prof = _current_profile()
            if prof is not None:
                fout.stage_ms = C.cast(prof.fwd, C.c_void_p)
                prof.fwd_calls += 1
            key = (dev.index, P, W, H, V, _scissor_key(rs))
            _sort_choice(key, rs, st)
            _sort_no_buckets(key, st)
            _forward_with_guess(lambda: lib.ggr_forward_views(C.byref(st), C.byref(vw), C.byref(fin), C.byref(fout), cb,
                                                              None, stream),
                                fout, holder, lib, dev, W, H, key, capacity, prof is not None)
            _sort_look(key, lib, fout, geom, P * V, stream)
            _sort_fell_back(key, fout)
        _tls.last_forward = (geom, P * V) if capacity > 0 else (None, int(fout.num_rendered))
        _tls.last_binning = (int(fout.depth_sort_used), int(fout.max_list_len))
        ctx.raster_settings = rs
        ctx.num_rendered = int(fout.num_rendered)
        ctx.dims = (P, M, H, W, V, B)
        ctx.form = (form, cov_full, sh_cm)
        shp = lambda t: None if t is None else t.shape
        ctx.in_shapes = (means3D.shape, shp(sh), opacities.shape, shp(aux), campos.shape, shp(colors_precomp),
                         shp(scales), shp(rotations), shp(cov3Ds_precomp))
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None,
                   means2D is not None)
        ctx.save_for_backward(means3D_c, sh_c, cp_c, op_c, sc_c, rot_c, cov_c, bg_c, view, proj, cam, radii, geom,
                              img, holder.get("bin"), aux_c, tf_c, sc_in, scratch)
        ctx.scratch_fresh = scratch is not None
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_depth):
        lib = _lib.load()
        rs = ctx.raster_settings
        (means3D, sh, cp, op, sc, rot, cov, bg, view, proj, cam, radii, geom, img, binb, aux, tf, sc_in,
         fwd_scratch) = ctx.saved_tensors
        P, M, H, W, V, B = ctx.dims
        PT = P * B   # rows of the flat [B·P, …] gradient arrays
        dev = means3D.device
        need_pose = any(ctx.needs_input_grad[7:10])
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if grad_color is None:
                grad_color = torch.zeros((V, 3, H, W), dtype=torch.float32, device=dev)
            grad_color, grad_depth = _f32c(grad_color), _f32c(grad_depth)
            form, cov_full, sh_cm = ctx.form
            e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            d_means3D, d_means2D, d_op = e(PT, 3), e(V, P, 3), e(PT)
            d_cov = e(PT, 3, 3) if (cov_full and cov is not None) else e(PT, 6)
            d_sh = (e(PT, 3, M) if sh_cm else e(PT, M, 3)) if sh is not None else None
            d_cp = e(PT, 3) if cp is not None else None
            d_sc = e(PT, 3) if sc is not None else None
            d_rot = e(PT, 4) if rot is not None else None
            d_aux = e(V, P) if (aux is not None and grad_depth is not None) else None
            d_view = e(V, 4, 4) if need_pose else None
            d_proj = e(V, 4, 4) if need_pose else None
            d_cam = e(V, 3) if need_pose else None
            zeroed = bool(getattr(ctx, "scratch_fresh", False)) and fwd_scratch is not None
            ctx.scratch_fresh = False
            scratch = fwd_scratch if zeroed else torch.empty((lib.ggr_backward_scratch_bytes_views(P, V),),
                                                             dtype=torch.uint8, device=dev)
            st = _settings_struct(rs._replace(tanfovx=0.0, tanfovy=0.0, tanfov=None), P, M, None, None, None, None)
            vw = _lib.GgrViews(num_views=V, viewmatrix=view.data_ptr(), projmatrix=proj.data_ptr(), campos=cam.data_ptr(),
                               bg=bg.data_ptr(), tanfov=tf.data_ptr(), input_scale=_ptr(sc_in), num_sets=B)
            bin_ = _lib.GgrBackwardIn(
                fwd=_lib.GgrForwardIn(means3D=_ptr(means3D), shs=_ptr(sh), colors_precomp=_ptr(cp), opacities=_ptr(op),
                                      scales=_ptr(sc), rotations=_ptr(rot), cov3D_precomp=_ptr(cov),
                                      aux_precomp=_ptr(aux), **form),
                radii=_ptr(radii), geom_buffer=geom.data_ptr(), image_buffer=img.data_ptr(),
                binning_buffer=_ptr(binb), num_rendered=ctx.num_rendered, dL_dout_color=grad_color.data_ptr(),
                dL_dout_depth=_ptr(grad_depth), scratch=scratch.data_ptr(), scratch_zeroed=int(zeroed))
            bout = _lib.GgrBackwardOut(
                dL_dmeans3D=d_means3D.data_ptr(), dL_dmeans2D=d_means2D.data_ptr(), dL_dshs=_ptr(d_sh),
                dL_dcolors_precomp=_ptr(d_cp), dL_dopacities=d_op.data_ptr(), dL_dcov3D=d_cov.data_ptr(),
                dL_dscales=_ptr(d_sc), dL_drotations=_ptr(d_rot), dL_daux=_ptr(d_aux), dL_dviewmatrix=_ptr(d_view),
                dL_dprojmatrix=_ptr(d_proj), dL_dcampos=_ptr(d_cam), stage_ms=None)
            prof = _current_profile()
            if prof is not None:
                bout.stage_ms = C.cast(prof.bwd, C.c_void_p)
                prof.bwd_calls += 1
            _check(lib.ggr_backward_views(C.byref(st), C.byref(vw), C.byref(bin_), C.byref(bout), stream),
                   "ggr_backward_views")
        means_shape, sh_shape, op_shape, aux_shape, cam_shape, cp_shape, sc_shape, rot_shape, cov_shape = ctx.in_shapes
        has_sh, has_cp, has_sc, has_cov, has_m2d = ctx.has
        return (
            d_means3D.reshape(means_shape),
            d_sh.reshape(sh_shape) if has_sh else None,
            d_cp.reshape(cp_shape) if has_cp else None,
            d_op.reshape(op_shape),
            d_sc.reshape(sc_shape) if has_sc else None,
            d_rot.reshape(rot_shape) if has_sc else None,
            d_cov.reshape(cov_shape) if has_cov else None,
            d_view if ctx.needs_input_grad[7] else None,
            d_proj if ctx.needs_input_grad[8] else None,
            d_cam.reshape(cam_shape) if ctx.needs_input_grad[9] else None,
            d_aux.reshape(aux_shape) if d_aux is not None else None,
            d_means2D if has_m2d else None,
            None, None, None, None, None,
        )


def rasterize_views(means3D, opacities, viewmatrices, projmatrices, campos, bg, tanfov, raster_settings, shs=None,
                    colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, aux_precomp=None,
                    input_scale=None, means2D=None):
    """V views in ONE launch set — of the SAME Gaussians (``means3D [P,3]``), or, with ``means3D
    [B,P,3]`` and every per-Gaussian input ``[B,P,…]``, of B Gaussian sets with V/B consecutive views each (the
    reference's ``(b v)`` flattening with per-batch-element Gaussians, ``decoder_splatting_cuda.py:40-60``; gradients
    then come back ``[B,P,…]``, summed over the views of each set).  This is what the reference does with a Python loop
    of rasterizer calls over a v× repeated copy of the Gaussian tensors (``decoder_splatting_cuda.py:40-60``,
    ``cuda_splatting.py:93-127``).  ``viewmatrices / projmatrices [V,4,4]``, ``campos / bg [V,3]``, ``tanfov [V,2]``
    (device tensors: tan(fov/2) per view), ``input_scale [V]`` or None, ``aux_precomp [V,P]`` or None, ``means2D
    [V,P,3]`` or None (only a gradient sink, as at the reference's call site).  ``raster_settings`` supplies the image
    size, ``sh_degree``, ``scale_modifier``, ``debug`` and the extension fields; its per-view fields are ignored.
    Returns ``(color [V,3,H,W], radii [V,P], depth [V,H,W])``; gradients w.r.t. the Gaussians arrive summed over
    the views, per-view results equal ``GaussianRasterizer``'s (same lists, bit-identical images)."""
    shs, colors_precomp = _none_if_empty(shs), _none_if_empty(colors_precomp)
    scales, rotations, cov3D_precomp = _none_if_empty(scales), _none_if_empty(rotations), _none_if_empty(cov3D_precomp)
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    return _RasterizeViews.apply(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                 viewmatrices, projmatrices, campos, aux_precomp, means2D, raster_settings, bg, tanfov,
                                 input_scale, torch.is_grad_enabled())


def camera_setup(extrinsics: torch.Tensor, intrinsics: torch.Tensor, near: torch.Tensor, far: torch.Tensor,
                 scale_invariant: bool = True):
    """Per-view camera quantities of GGRt's call site (reference ``cuda_splatting.py:18-46,66-73,82-89``) from ONE
    kernel launch, all left on the device: returns ``(viewmatrix [n,4,4], projmatrix [n,4,4], campos [n,3],
    tanfov [n,2], scale [n])`` for ``extrinsics [n,4,4]`` (camera-to-world), normalised ``intrinsics [n,3,3]``,
    ``near/far [n]``.  No autograd (the reference does not differentiate through its settings either)."""
    lib = _lib.load()
    dev = extrinsics.device
    if dev.type != "cuda":
        raise RuntimeError("camera_setup needs ROCm GPU tensors; there is no CPU path")
    n = int(extrinsics.shape[0])
    with torch.no_grad(), torch.cuda.device(dev):
        e, k = _f32c(extrinsics.detach()), _f32c(intrinsics.detach())
        nr, fr = _f32c(near.detach()), _f32c(far.detach())
        view = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
        full = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
        campos = torch.empty((n, 3), dtype=torch.float32, device=dev)
        tanfov = torch.empty((n, 2), dtype=torch.float32, device=dev)
        scale = torch.empty((n,), dtype=torch.float32, device=dev)
        _check(lib.ggr_camera_setup(n, e.data_ptr(), k.data_ptr(), nr.data_ptr(), fr.data_ptr(), int(bool(scale_invariant)),
                                    view.data_ptr(), full.data_ptr(), campos.data_ptr(), tanfov.data_ptr(),
                                    scale.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "ggr_camera_setup")
    return view, full, campos, tanfov, scale


def last_forward_status():
    """(num_rendered, overflowed) of this thread's most recent forward — synchronises the stream.  Meant for
    the sync-free mode (``list_capacity > 0``), where ``ggr_forward`` itself reads nothing back: an overflowed
    frame was rendered from lists cut at the buffer's end and should be redone with a larger capacity."""
    last = getattr(_tls, "last_forward", None)
    if last is None:
        raise RuntimeError("no forward has run on this thread")
    geom, P = last
    if geom is None:  # exact mode: ggr_forward itself read the count back (and fails on a sort fault)
        return P, False
    lib = _lib.load()
    n, ov = C.c_int64(0), C.c_int32(0)
    with torch.cuda.device(geom.device):
        _check(lib.ggr_forward_status(geom.data_ptr(), P, C.byref(n), C.byref(ov),
                                      torch.cuda.current_stream(geom.device).cuda_stream), "ggr_forward_status")
    return int(n.value), bool(ov.value)


def last_forward_sort_form() -> str:
    """"per_tile" | "buckets" | "3pass" | "fell_back" | "buckets_slow": GgrForwardOut.depth_sort_used of this thread's most recent forward in full
    (the global sort's bucket form, its three-pass form, or three passes after the bucket form gave the frame up)."""
    last = getattr(_tls, "last_binning", None)
    if last is None:
        raise RuntimeError("no forward has run on this thread")
    return {2: "per_tile", 1: "buckets", 0x101: "3pass", _lib.DEPTH_SORT_FELL_BACK: "fell_back",
            _lib.DEPTH_SORT_SLOW: "buckets_slow"}.get(last[0], str(last[0]))


def last_forward_binning():
    """("global" | "per_tile", longest tile list or -1) of this thread's most recent forward: which form of the depth sort
    built its lists (GgrForwardOut.depth_sort_used — "global" also after a per-tile attempt met a list it could not hold)."""
    last = getattr(_tls, "last_binning", None)
    if last is None:
        raise RuntimeError("no forward has run on this thread")
    return ("per_tile" if last[0] == 2 else "global"), last[1]


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, aux_precomp=None):
    """Function form, argument order of upstream's ``rasterize_gaussians`` (+ the optional aux feature)."""
    rs = raster_settings
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.campos, aux_precomp, rs,
                                     torch.is_grad_enabled())


class GaussianRasterizer(nn.Module):
    """Drop-in for ``diff_gaussian_rasterization.GaussianRasterizer`` (call site:
    reference ``cuda_splatting.py:114-125``).  Returns the 3-tuple ``(color[3,H,W], radii[P],
    depth[H,W])`` that the live call site unpacks at ``:118``."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        rs = self.raster_settings
        dev = positions.device
        if dev.type != "cuda":
            raise RuntimeError("markVisible needs a ROCm GPU tensor; there is no CPU path")
        with torch.no_grad(), torch.cuda.device(dev):
            pos = _f32c(positions)
            view, proj = _f32c(rs.viewmatrix.to(dev)), _f32c(rs.projmatrix.to(dev))
            present = torch.empty((pos.shape[0],), dtype=torch.uint8, device=dev)
            _check(lib.ggr_mark_visible(pos.shape[0], pos.data_ptr(), view.data_ptr(), proj.data_ptr(),
                                        present.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "ggr_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, aux_precomp=None):
        """``aux_precomp`` [P] (extension, optional): a 4th feature blended like a colour channel; the third
        return value then is Σ aux·α·T instead of Σ z·α·T (one rasterization serves GGRt's colour AND depth
        pass — see ``splatting.render_color_and_depth``)."""
        shs, colors_precomp = _none_if_empty(shs), _none_if_empty(colors_precomp)
        scales, rotations, cov3D_precomp = _none_if_empty(scales), _none_if_empty(rotations), _none_if_empty(cov3D_precomp)
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self._settings_for_call(), aux_precomp)

    def _settings_for_call(self) -> GaussianRasterizationSettings:
        """The settings a forward runs with (the `diff_gaussian_rasterization` import shim fills in its SH-cap default here)."""
        return self.raster_settings


def debug_forward_state(means3D, opacities, raster_settings, shs=None, colors_precomp=None, cov3D_precomp=None,
                        scales=None, rotations=None):
    """Test helper: runs forward once and returns the kernel intermediates (device tensors) unpacked
    from the opaque buffers through ``ggr_debug_unpack_*`` — used by the stage-by-stage parity tests."""
    lib = _lib.load()
    # call the raw Function with a stand-in ctx to get hold of the saved buffers
    class _Ctx:  # minimal stand-in for the autograd ctx
        def set_materialize_grads(self, v): pass
        def save_for_backward(self, *t): self.saved = t
        def mark_non_differentiable(self, *t): pass
    ctx = _Ctx()
    rs = raster_settings
    with torch.no_grad():
        color, radii, depth = _RasterizeGaussians.forward(ctx, means3D, torch.zeros_like(means3D), shs, colors_precomp,
                                                          opacities, scales, rotations, cov3D_precomp, rs.viewmatrix,
                                                          rs.projmatrix, rs.campos, None, rs)
    P, M, H, W = ctx.dims
    dev = means3D.device
    geom, img, binb = ctx.saved[12], ctx.saved[13], ctx.saved[14]
    N = ctx.num_rendered
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(color=color, radii=radii, out_depth=depth, num_rendered=N,
               depth=torch.empty(P, device=dev), xy=torch.empty(P, 2, device=dev),
               conic_opacity=torch.empty(P, 4, device=dev), rgb=torch.empty(P, 3, device=dev),
               tiles_touched=torch.empty(P, dtype=torch.int32, device=dev),
               clamped=torch.empty(P, 3, dtype=torch.uint8, device=dev),
               point_list=torch.empty(max(N, 1), dtype=torch.int32, device=dev),
               ranges=torch.empty(max(tiles, 1), 2, dtype=torch.int32, device=dev),
               final_T=torch.empty(H, W, device=dev), n_contrib=torch.empty(H, W, dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(lib.ggr_debug_unpack_geom(geom.data_ptr(), P, out["depth"].data_ptr(), out["xy"].data_ptr(),
                                         out["conic_opacity"].data_ptr(), out["rgb"].data_ptr(),
                                         out["tiles_touched"].data_ptr(), out["clamped"].data_ptr(), stream),
               "ggr_debug_unpack_geom")
        _check(lib.ggr_debug_unpack_binning(_ptr(binb), img.data_ptr(), N, W, H, out["point_list"].data_ptr(),
                                            out["ranges"].data_ptr(), out["final_T"].data_ptr(),
                                            out["n_contrib"].data_ptr(), stream), "ggr_debug_unpack_binning")
    out["point_list"] = out["point_list"][:N]
    out["ranges"] = out["ranges"][:tiles]
    return out
