"""Builds ``csrc/*.hip`` into ``libggr_raster.so`` (in-tree, gfx950 only) with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libggr_raster.so")
SOURCES = ["api.hip", "preprocess.hip", "binning.hip", "tile_lists.hip", "tile_sort.hip", "blend_fwd.hip", "blend_bwd.hip",
           "preprocess_bwd.hip", "camera.hip", "util.hip"]
HEADERS = ["ggr_common.h", "blend_common.h", "tile_sort.h", "sh_stage.h", "sh_terms.h", os.path.join("..", "..", "include", "ggr_raster.h")]
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


HASH_MARKER = b"ggr-source-hash:"   # followed by 64 hex digits inside the .so (api.hip, ggr_source_hash())

# what the last build_library() call of this process did: {"compiled": bool, "source_hash": str, "seconds": float}
last_build: dict = {}


def _code_only(text: str) -> str:
    """C / C++ source without comments and without insignificant white space (string and character literals kept as they
    are): what the compiler sees.  A comment edit must neither force a rebuild nor mark a profile as stale."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == "/" and i + 1 < n and text[i + 1] == "/":
            while i < n and text[i] != "\n":
                i += 1
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        else:
            out.append(c)
            i += 1
    lines = (" ".join(l.split()) for l in "".join(out).split("\n"))
    return "\n".join(l for l in lines if l)


def source_hash() -> str:
    """sha256 over the CODE of every translation unit and header of the library (name + contents without comments and
    insignificant white space) and the compiler flags.  The library carries the hash it was built from
    (``ggr_source_hash()``); ``_lib.load()`` refuses a library whose hash differs from ``csrc/`` as it is now — a stale
    ``.so`` can neither pass for a build nor be measured by accident."""
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        h.update(os.path.basename(name).encode() + b"\0")
        with open(os.path.join(CSRC, name), "r", encoding="utf-8") as f:
            h.update(_code_only(f.read()).encode("utf-8"))
        h.update(b"\0")
    h.update(" ".join(FLAGS).encode())
    for k in sorted(EXTRA_FLAGS):
        h.update((k + " " + " ".join(EXTRA_FLAGS[k])).encode())
    # dev variants (GGR_EXTRA_HIPCC_FLAGS, scripts/build_variants.sh) are OTHER libraries: they carry another hash than
    # the default build, so that one of them cannot pass for the tree's build (GGR_SKIP_SOURCE_HASH=1 loads it knowingly)
    extra = " ".join(os.environ.get("GGR_EXTRA_HIPCC_FLAGS", "").split())
    if extra:
        h.update(b"\0extra " + extra.encode())
    return h.hexdigest()


def embedded_hash(path: str = None):
    """The source hash a built library carries, read from the file's bytes (no dlopen); None if absent."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        blob = f.read()
    i = blob.find(HASH_MARKER)
    if i < 0:
        return None
    hx = blob[i + len(HASH_MARKER): i + len(HASH_MARKER) + 64]
    try:
        return hx.decode("ascii") if len(hx) == 64 and int(hx, 16) >= 0 else None
    except ValueError:
        return None


def needs_build() -> bool:
    return embedded_hash() != source_hash()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compiles csrc/*.hip for gfx950 and links ``libggr_raster.so`` unless the library in the tree was built from
    exactly these sources (its embedded hash equals ``source_hash()``) — `force` compiles regardless.  Objects go
    to a fresh temporary directory (no stale ``.o`` can be linked); ``last_build`` records what happened."""
    want = source_hash()
    if not force and embedded_hash() == want:
        last_build.clear()
        last_build.update(compiled=False, source_hash=want, seconds=0.0)
        return LIB
    t0 = time.time()
    hipcc = _hipcc()
    extra_env = os.environ.get("GGR_EXTRA_HIPCC_FLAGS", "").split()  # dev experiments only (e.g. -DGGR_XCD_SPLIT=4)
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    objdir = tempfile.mkdtemp(prefix="ggr_build_", dir=os.path.join(CSRC, "build"))  # (same filesystem as LIB: os.replace)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        stamp = [f'-DGGR_SOURCE_HASH="{want}"'] if src == "api.hip" else []
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *extra_env, *stamp, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            for _, q in procs:
                if q.poll() is None:
                    q.kill()
            shutil.rmtree(objdir, ignore_errors=True)
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp_lib = os.path.join(objdir, "libggr_raster.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp_lib]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError(f"link failed:\n{out.stdout}")
    if embedded_hash(tmp_lib) != want:
        raise RuntimeError("the linked library does not carry the source hash it was built with")
    os.replace(tmp_lib, LIB)   # (atomic: a process that has the old library mapped keeps its inode)
    shutil.rmtree(objdir, ignore_errors=True)
    last_build.clear()
    last_build.update(compiled=True, source_hash=want, seconds=round(time.time() - t0, 1))
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(last_build)
