"""flip_probe.py — are the pixels where the HIP image and the C oracle differ by more than the forward tolerance
threshold flips?  (dev tool; run on the GPU box:  python tests/tools/flip_probe.py 79 133 139 187 269 297)

For every such pixel of a random-sweep case `tests.helpers.threshold_flips` walks the tile's list in fp32 numpy with the oracle's
per-Gaussian values and reports the entry nearest to one of the two discrete decisions of the compositing rule: α against 1/255
and T·(1 − α) against 1e-4, as a relative distance.  A distance of a few 1e-7 is one ulp of the exponential: the two implementations may decide
differently there, and the pixel then moves by up to α·T·|c| — a discrete event, not an error."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import FWD_ATOL, hip_forward_backward, oracle_forward, threshold_flips   # noqa: E402
from tests.test_gpu_random_sweep import _case                              # noqa: E402
from ggrt_official_amd.synthetic import upstream_gradient                  # noqa: E402


def probe(i):
    sc, use_scale_rot = _case(i)
    cap = 3 + i % 2
    st = oracle_forward(sc, use_cov=not use_scale_rot, sh_cap=cap)
    dL = upstream_gradient(sc.width, sc.height, seed=300 + i)
    color, *_ = hip_forward_backward(sc, dL, use_cov=not use_scale_rot, sh_max_degree=cap)
    return sc, threshold_flips(st, color)


if __name__ == "__main__":
    for i in (int(a) for a in sys.argv[1:]):
        sc, rows = probe(i)
        print(f"case {i}: {sc.width}x{sc.height}, P {sc.means3D.shape[0]}: {len(rows)} pixel(s) beyond {FWD_ATOL}")
        for y, x, dd, g, rel, _ in rows:
            print(f"   pixel ({y},{x}) |d| {dd:.2e}   Gaussian {g}: relative distance to the nearest threshold {rel:.1e}")
