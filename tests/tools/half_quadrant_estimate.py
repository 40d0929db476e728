# Offline estimate: how many (wave, entry) slots would blend_bwd walk if every 8x8 quadrant kept two survivor lists
# (its two 8x4 halves) and paired them, instead of one list for the whole quadrant?
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ggrt_official_amd.synthetic import CONFIGS, make_scene
from helpers import oracle_forward
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(CONFIGS[cfgname]); layout = cfg.pop("layout", None)
sc = make_scene(cfg["num_points"], cfg["width"], cfg["height"], sh_degree=cfg["sh_degree"], profile=cfg["profile"], seed=0)
t = time.time(); st = oracle_forward(sc); print("oracle fwd", time.time() - t, "s; N =", st.num_rendered)
W, H = st.W, st.H
gx, gy = (W + 15) // 16, (H + 15) // 16
xy = st.xy; co = st.conic_opacity
ncon = st.n_contrib.reshape(H, W)
rng = np.random.default_rng(0)
tiles = rng.choice(gx * gy, size=min(400, gx * gy), replace=False)

def box_min_q(mx, my, cxx, cxy, cyy, x0, y0, x1, y1):
    dxl, dxh, dyl, dyh = mx - x1, mx - x0, my - y1, my - y0
    inside = (dxl <= 0) & (dxh >= 0) & (dyl <= 0) & (dyh >= 0)
    ry, rx = -cxy / cyy, -cxy / cxx
    qf = lambda dx, dy: cxx * dx * dx + 2 * cxy * dx * dy + cyy * dy * dy
    q1 = qf(dxl, np.clip(ry * dxl, dyl, dyh)); q2 = qf(dxh, np.clip(ry * dxh, dyl, dyh))
    q3 = qf(np.clip(rx * dyl, dxl, dxh), dyl); q4 = qf(np.clip(rx * dyh, dxl, dxh), dyh)
    return np.where(inside, 0.0, np.minimum(np.minimum(q1, q2), np.minimum(q3, q4)))

tot_full = tot_pair = tot_top = tot_bot = tot_units = 0
tot4 = tot4_units = 0
for tl in tiles:
    tx, ty = tl % gx, tl // gx
    a, b = st.ranges[tl]
    ids = st.point_list[a:b]
    if len(ids) == 0: continue
    mx, my = xy[ids, 0], xy[ids, 1]
    cxx, cxy, cyy, op = co[ids, 0], co[ids, 1], co[ids, 2], co[ids, 3]
    qmax = 2 * np.log(255 * op)
    pos = np.arange(len(ids))
    for q in range(4):
        qx0, qy0 = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        if qx0 >= W or qy0 >= H: continue
        def keep(x0, y0, x1, y1):
            x1 = min(x1, W - 1); y1 = min(y1, H - 1)
            if y0 > y1: return np.zeros(len(ids), bool)
            wl = ncon[y0:y1 + 1, x0:x1 + 1].max()
            return (box_min_q(mx, my, cxx, cxy, cyy, x0, y0, x1, y1) * 0.999 <= qmax + 1e-3) & (pos < wl)
        full = keep(qx0, qy0, qx0 + 7, qy0 + 7)
        top = keep(qx0, qy0, qx0 + 7, qy0 + 3); bot = keep(qx0, qy0 + 4, qx0 + 7, qy0 + 7)
        tot_full += full.sum(); tot_top += top.sum(); tot_bot += bot.sum()
        tot_pair += max(top.sum(), bot.sum()); tot_units += top.sum() + bot.sum()
        q4 = [keep(qx0 + 4 * (i & 1), qy0 + 4 * (i >> 1), qx0 + 4 * (i & 1) + 3, qy0 + 4 * (i >> 1) + 3).sum() for i in range(4)]
        tot4 += max(q4); tot4_units += sum(q4)
print(cfgname, "tiles sampled", len(tiles))
print("slots: one list per 8x8 quadrant", tot_full)
print("two 8x4 lists, paired: slots", tot_pair, "= %.3f of now; (Gaussian, half) units to commit %d = %.3f of now" % (tot_pair / tot_full, tot_units, tot_units / tot_full))
print("four 4x4 lists: slots", tot4, "= %.3f of now; units %d = %.3f" % (tot4 / tot_full, tot4_units, tot4_units / tot_full))
