#!/usr/bin/env python3
"""Generates tests/golden/decoder_b2v3.npz and tests/golden/ply_export_scene.npz by running the REFERENCE's own
`DecoderSplattingCUDA` (/root/reference/ggrt/model/pixelsplat/decoder/decoder_splatting_cuda.py:19-85: `forward`,
`render_depth`) and `export_ply` (/root/reference/ggrt/model/pixelsplat/ply_export.py:26-92) on seeded CPU inputs.

Runs ONLY in the build container (needs /root/reference); nothing of the reference travels: the fixtures are plain
arrays (inputs, every argument that reached the rasterizer boundary in the order the reference issued the calls, the
returned `color[b,v,3,h,w]` / `depth[b,v,h,w]`; for the exporter the vertex table the reference handed to `plyfile`).

Import technique as in make_callsite_golden.py: `jaxtyping` and `diff_gaussian_rasterization`
are stubs, `..types` and `.decoder` are loaded by path next to `cuda_splatting.py`; `plyfile` is a recording stub
(`PlyElement.describe(elements, "vertex")` keeps the structured array, `PlyData.write` stores nothing).
The rasterizer boundary is served by the CPU oracle (oracle/torch_raster.py), so the recorded images pin the decoder's
`(b v)` flattening / `repeat` order and its depth-mode handling, not the rasterizer arithmetic.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_callsite_golden as mc  # noqa: E402  (stubs, reference loader, seeded inputs)

REF = mc.REF


def load_decoder():
    mc.install_stubs()
    mc.load_reference()  # registers placeholder packages and decoder.cuda_splatting

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    ty = load("ggrt.model.pixelsplat.types", "ggrt/model/pixelsplat/types.py")
    load("ggrt.model.pixelsplat.decoder.decoder", "ggrt/model/pixelsplat/decoder/decoder.py")
    dec = load("ggrt.model.pixelsplat.decoder.decoder_splatting_cuda",
               "ggrt/model/pixelsplat/decoder/decoder_splatting_cuda.py")
    return ty, dec


def decoder_inputs(seed=21, b=2, v=3, g_count=96, d_sh=25, h=32, w=40):
    """b DIFFERENT Gaussian sets, b·v DIFFERENT cameras (poses, near, far; a shared pinhole as in GGRt's batches —
    the reference's projection matrix reads intrinsics[0] for every view anyway), so that any mix-up of the
    (b v) order or of which set a view sees changes the images."""
    g = torch.Generator().manual_seed(seed)
    extr = torch.stack([mc.random_pose(g) for _ in range(b * v)]).reshape(b, v, 4, 4)
    fx = 0.9 + 0.2 * torch.rand(1, generator=g).item()
    intr = torch.eye(3).repeat(b, v, 1, 1)
    intr[..., 0, 0] = fx
    intr[..., 1, 1] = fx * w / h
    intr[..., 0, 2] = 0.47
    intr[..., 1, 2] = 0.54
    near = 0.7 + 0.8 * torch.rand(b, v, generator=g)
    far = 40.0 + 50.0 * torch.rand(b, v, generator=g)
    z = near.max() * (1.5 + 6 * torch.rand(b, g_count, generator=g))
    xy = (torch.rand(b, g_count, 2, generator=g) - 0.5) * z[..., None]
    means = torch.cat([xy, z[..., None]], -1)
    A = torch.randn(b, g_count, 3, 3, generator=g) * 0.04 * z[..., None, None]
    cov = A @ A.transpose(-1, -2) + 1e-5 * torch.eye(3)
    sh = torch.randn(b, g_count, 3, d_sh, generator=g) * 0.3
    sh[..., 0] = torch.randn(b, g_count, 3, generator=g)
    op = 0.05 + 0.9 * torch.rand(b, g_count, generator=g)
    return dict(extrinsics=extr, intrinsics=intr, near=near, far=far, means=means, covariances=cov, harmonics=sh,
                opacities=op), (h, w)


CALL_KEYS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
             "sh_degree", "campos", "prefiltered", "means3D", "opacities", "shs", "cov3D_precomp")


def decoder_golden():
    ty, dec = load_decoder()
    inp, shape = decoder_inputs()
    gs = ty.Gaussians(means=inp["means"], covariances=inp["covariances"], harmonics=inp["harmonics"],
                      opacities=inp["opacities"])
    module = dec.DecoderSplattingCUDA(dec.DecoderSplattingCUDACfg(name="splatting_cuda"))
    blob = {f"in_{k}": v.numpy() for k, v in inp.items()}
    blob["image_shape"] = np.asarray(shape)
    colour_calls = None
    for cap in (3, 4):
        for mode in (None, "depth", "log", "disparity"):
            mc.RECORD.clear()
            mc.SH_CAP[0] = cap
            out = module(gs, inp["extrinsics"], inp["intrinsics"], inp["near"], inp["far"], shape, depth_mode=mode)
            tag = f"cap{cap}_{mode or 'none'}"
            n_colour = inp["extrinsics"].shape[0] * inp["extrinsics"].shape[1]
            assert len(mc.RECORD) == n_colour * (1 if mode is None else 2)
            recs = [mc.to_np({k: r[k] for k in CALL_KEYS if r.get(k) is not None}) for r in mc.RECORD]
            # the colour pass's boundary calls do not depend on depth_mode or on what the rasterizer does with band 4:
            # kept once
            if colour_calls is None:
                colour_calls = recs[:n_colour]
                for i, r in enumerate(colour_calls):
                    for k, a in r.items():
                        blob[f"colour_call{i}_{k}"] = a
            else:
                for r0, r1 in zip(colour_calls, recs[:n_colour]):
                    assert all(np.array_equal(r0[k], r1[k]) for k in r0)
            # the colour image does not depend on depth_mode, the depth image (one SH coefficient) not on the cap:
            # each kept once, equality across the other axis asserted here
            ck = f"cap{cap}_color"
            if ck in blob:
                assert np.array_equal(blob[ck], out.color.detach().numpy())
            blob[ck] = out.color.detach().numpy()
            if mode is not None:
                dk = f"{mode}_depth"
                if dk in blob:
                    assert np.array_equal(blob[dk], out.depth.detach().numpy())
                blob[dk] = out.depth.detach().numpy()
                if cap == 3:
                    for i, r in enumerate(recs[n_colour:]):
                        for k, a in r.items():
                            if k in ("means3D", "cov3D_precomp", "opacities") and \
                                    np.array_equal(a, colour_calls[i][k]):
                                continue  # same tensors as the colour call of that view (asserted by the test)
                            blob[f"{mode}_call{i}_{k}"] = a
                # render_depth called directly must equal forward's depth
                d2 = module.render_depth(gs, inp["extrinsics"], inp["intrinsics"], inp["near"], inp["far"], shape,
                                         mode=mode)
                assert torch.equal(d2, out.depth)
    path = os.path.join(mc.OUT, "decoder_b2v3.npz")
    np.savez_compressed(path, **blob)
    print(f"decoder_b2v3: {os.path.getsize(path) / 1024:.0f} KiB, keys {len(blob)}")


# ---------------------------------------------------------------------------------------------------- export_ply
class _Recorder:
    elements = None
    name = None
    written_to = None


def install_plyfile_stub():
    pf = types.ModuleType("plyfile")

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            _Recorder.elements = np.array(elements, copy=True)
            _Recorder.name = name
            return ("element", name)

    class PlyData:
        def __init__(self, elements):
            self.elements = elements

        def write(self, path):
            _Recorder.written_to = str(path)

    pf.PlyElement, pf.PlyData = PlyElement, PlyData
    sys.modules["plyfile"] = pf


def ply_golden():
    mc.install_stubs()
    install_plyfile_stub()
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_ply_export", os.path.join(REF, "ggrt/model/pixelsplat/ply_export.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(31)
    n = 257
    extr = mc.random_pose(g, jitter=0.6)
    means = torch.randn(n, 3, generator=g) * torch.tensor([2.0, 0.7, 4.0]) + torch.tensor([0.3, -1.0, 5.0])
    scales = torch.exp(torch.randn(n, 3, generator=g) * 0.7 - 3.0)
    rot = torch.randn(n, 4, generator=g)
    rot = rot / rot.norm(dim=-1, keepdim=True)
    rot[:4] = torch.eye(4)  # pure-axis quaternions
    sh = torch.randn(n, 3, 9, generator=g)
    op = torch.rand(n, generator=g)
    import pathlib
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        mod.export_ply(extr, means, scales, rot, sh, op, pathlib.Path(td) / "sub" / "scene.ply")
    el = _Recorder.elements
    assert _Recorder.name == "vertex" and el.dtype.names is not None
    table = np.stack([el[c] for c in el.dtype.names], axis=1)
    assert table.dtype == np.float32 and el.tobytes() == table.tobytes()
    assert list(el.dtype.names) == mod.construct_list_of_attributes(0)
    path = os.path.join(mc.OUT, "ply_export_scene.npz")
    np.savez_compressed(path, in_extrinsics=extr.numpy(), in_means=means.numpy(), in_scales=scales.numpy(),
                        in_rotations=rot.numpy(), in_harmonics=sh.numpy(), in_opacities=op.numpy(),
                        columns=np.asarray(el.dtype.names), formats=np.asarray([el.dtype[c].str for c in el.dtype.names]),
                        table=table, element_name=np.asarray(_Recorder.name),
                        attributes_rest2=np.asarray(mod.construct_list_of_attributes(2)))
    print(f"ply_export_scene: {n} vertices × {table.shape[1]} columns, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    decoder_golden()
    ply_golden()
