"""dev: why are the GGRt-shape legs of bench.py slower behind its HIP-graph legs when the main workload runs per tile?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ggrt_official_amd import GaussianRasterizer
from ggrt_official_amd.synthetic import CONFIGS
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["none"]

def t_c5p(tag):
    wl = bench.Workload("C5p", CONFIGS["C5p"], dev)
    _, ev = bench.timed_steps(lambda i: wl.step(), 30, 10, dev, prewarm_ms=40)
    print(tag, "C5p median ms", bench.percentiles(ev)["median"], {k: round(v, 3) for k, v in wl.stage_times(3).items()}, flush=True)

wl = bench.Workload("C3", CONFIGS["C3"], dev)
for _ in range(50): wl.step()
torch.cuda.synchronize()
if "before" in which:
    t_c5p("before")
if "streams" in which:   # a host that creates streams before the library's first global-mode forward
    keep = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("NSTREAMS", "4")))]
    for st in keep:
        with torch.cuda.stream(st): torch.zeros(8, device=dev)
    torch.cuda.synchronize()
    t_c5p("after creating streams")
if "two" in which:
    wl2 = bench.Workload("C3", CONFIGS["C3"], dev, seed=1)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
    pair = (wl, wl2)
    for st in lanes: st.wait_stream(torch.cuda.current_stream(dev))
    for i in range(40):
        with torch.cuda.stream(lanes[i & 1]): pair[i & 1].step()
    for st in lanes: torch.cuda.current_stream(dev).wait_stream(st)
    torch.cuda.synchronize(); del wl2
    t_c5p("after two-streams")
if "graph" in which:
    rast_g = GaussianRasterizer(wl.rs._replace(list_capacity=14000000))
    def step_g():
        for t in wl.leaves: t.grad = None
        c, _, _ = rast_g(means3D=wl.means, means2D=wl.means2D, opacities=wl.op, shs=wl.shs, cov3D_precomp=wl.cov)
        c.backward(wl.dL)
    side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2): step_g()
    torch.cuda.current_stream(dev).wait_stream(side)
    if "capture" in which:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): step_g()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); del g
    t_c5p("after graph leg")
