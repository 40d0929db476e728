"""The drop-in boundary driven from plain C (`tests/c_abi/abi_smoke.c`, compiled with gcc as C11 — no C++, no
Python, no torch): the header is valid C, the library links, and (on the GPU) one forward + backward through
hipMalloc'd buffers reproduces the closed form of a single centred Gaussian."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c")
LIBDIR = os.path.join(ROOT, "ggrt_official_amd")


def _build(out):
    from ggrt_official_amd import _build
    _build.build_library()
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__", SRC,
           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L" + LIBDIR, "-L/opt/rocm/lib", "-lggr_raster",
           "-lamdhip64", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return out


def test_header_is_c11_and_the_library_links_from_c(tmp_path):
    exe = _build(str(tmp_path / "abi_smoke"))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_c_host_forward_backward_known_answer(tmp_path):
    exe = _build(str(tmp_path / "abi_smoke"))
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "C ABI SMOKE OK" in r.stdout, r.stdout
