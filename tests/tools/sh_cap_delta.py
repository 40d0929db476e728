#!/usr/bin/env python3
"""What the choice `sh_max_degree` 3 vs 4 changes on a GGRt-like scene — measured on the CPU oracle.

Scene: profile B of ggrt_official_amd/synthetic.py (pixel-aligned Gaussians, opacity = top-1 bucket probability / 3),
GGRt's `sh_degree = 4` / 25 coefficients per channel whose band-ℓ coefficients carry GGRt's mask 0.1·0.25^ℓ
(reference encoder/common/gaussian_adapter.py:45-46), upstream gradient N(0,1)/(3HW).  Reported: max-abs / mean-abs
image difference, PSNR between the two renders, and the rel-L2 difference of every gradient tensor (for dL/dSH also on
the rows 0..15 both caps write).  Test infrastructure (uses oracle/): `python tests/tools/sh_cap_delta.py [P W H]`.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ggrt_official_amd.synthetic import make_scene, upstream_gradient  # noqa: E402
from oracle import c_oracle  # noqa: E402


def measure(P=337_920, W=480, H=352, seed=0):
    sc = make_scene(P, W, H, sh_degree=4, profile="B", seed=seed)
    dpix = upstream_gradient(W, H).numpy()
    res = {}
    for cap in (3, 4):
        st = c_oracle.forward(sc.means3D.numpy(), sc.opacities.numpy(), sc.viewmatrix.numpy(), sc.projmatrix.numpy(),
                              sc.campos.numpy(), sc.bg.numpy(), W, H, sc.tanfovx, sc.tanfovy, sh_degree=4,
                              shs=sc.shs.numpy(), cov3D_precomp=sc.cov3D.numpy(), sh_cap=cap)
        res[cap] = (st.color, c_oracle.backward(st, dpix))
    (c3, g3), (c4, g4) = res[3], res[4]
    d = np.abs(c3 - c4)
    out = dict(scene=f"profile B, P={P}, {W}x{H}, sh_degree 4 / 25 coefficients, band l ~ N(0, (0.1*0.25^l)^2), seed {seed}",
               image_max_abs=float(d.max()), image_mean_abs=float(d.mean()),
               image_psnr_dB=float(10 * np.log10(1.0 / max(float((d ** 2).mean()), 1e-30))),
               image_peak=float(np.abs(c3).max()))
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30))
    for k in ("means3D", "means2D", "opacities", "cov3D_precomp", "shs"):
        out[f"grad_rel_l2_{k}"] = rel(g3[k], g4[k])
    out["grad_rel_l2_shs_rows_0_15"] = rel(g3["shs"][:, :16], g4["shs"][:, :16])
    out["grad_shs_band4_share_of_norm"] = float(np.linalg.norm(g4["shs"][:, 16:]) / np.linalg.norm(g4["shs"]))
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]]
    print(json.dumps(measure(*a), indent=1))
