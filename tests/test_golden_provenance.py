"""Where the reference is present (the build container: /root/reference), the committed fixtures under tests/golden/ must be
exactly what their generators produce from it — the fixtures ARE outputs of the reference's own call site, decoder and
exporter (served by the CPU oracle at the rasterizer boundary), not hand-edited arrays.  Skipped on the GPU box, where the
reference does not exist; the generators run in a subprocess (they inject stub modules into sys.modules)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/ggrt"), reason="the reference tree is not on this machine")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("script,patterns", [("make_callsite_golden.py", ["callsite_*.npz", "deferred_backprop_*.npz"]),
                                             ("make_decoder_golden.py", ["decoder_b2v3.npz", "ply_export_scene.npz"])])
def test_fixtures_are_what_the_generators_produce(tmp_path, script, patterns):
    env = dict(os.environ, GGR_GOLDEN_OUT=str(tmp_path), OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, os.path.join(GOLDEN, script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-3000:]
    names = sorted(os.path.basename(f) for pat in patterns for f in glob.glob(os.path.join(GOLDEN, pat)))
    assert names and names == sorted(os.path.basename(f) for pat in patterns for f in glob.glob(os.path.join(str(tmp_path), pat)))
    for n in names:
        a, b = np.load(os.path.join(GOLDEN, n), allow_pickle=False), np.load(os.path.join(str(tmp_path), n), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), n
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (n, k)
            if a[k].dtype.kind == "f":   # (images through the torch CPU oracle: thread count may change a summation order)
                np.testing.assert_allclose(a[k], b[k], rtol=0, atol=2e-6, err_msg=f"{n}:{k}")
            else:
                assert np.array_equal(a[k], b[k]), (n, k)
