"""Builds ``csrc/*.hip`` into ``libggr_raster.so`` (in-tree, gfx950 only) with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libggr_raster.so")
SOURCES = ["api.hip", "preprocess.hip", "binning.hip", "tile_lists.hip", "blend_fwd.hip", "blend_bwd.hip",
           "preprocess_bwd.hip", "camera.hip"]
HEADERS = ["ggr_common.h", "blend_common.h", "sh_stage.h", os.path.join("..", "..", "include", "ggr_raster.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]
# per-file additions.  The blend kernels are VALU-bound and the SLP vectoriser packs their fp32 math into v_pk_*
# instructions, which run at half rate on gfx950 (no gain) and need their operands moved into adjacent registers
# (51 extra v_mov in blend_bwd): without it blend_bwd is 7 % faster.  The streaming kernels are left alone.
EXTRA_FLAGS = {"blend_fwd.hip": ["-fno-slp-vectorize"], "blend_bwd.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    extra_env = os.environ.get("GGR_EXTRA_HIPCC_FLAGS", "").split()  # dev experiments only (e.g. -DGGR_XCD_SPLIT=4)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *extra_env, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError(f"link failed:\n{out.stdout}")
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
