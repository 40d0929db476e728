"""pose_cost.py — what the camera gradient (dL/dviewmatrix, dL/dprojmatrix, dL/dcampos) costs in the C3 step (dev tool)."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from ggrt_official_amd import synthetic
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
for pose in (True, False, True, False):
    wl = bench.Workload('C3', synthetic.CONFIGS['C3'], dev, seed=0, pose=pose)
    el, ps = bench.timed_steps(lambda i: wl.step(), 200, 20, dev, prewarm_ms=60)
    st = wl.stage_times(5)
    print(pose, round(el/200*1e3,4), bench.percentiles(ps)['median'], {k: round(v,4) for k,v in st.items() if k.startswith('bwd')}, flush=True)
    del wl
