"""Call-site benchmark at GGRt's shape (C5': 1.01 M pixel-aligned Gaussians, 480x352, d_sh 25, colour + depth,
fwd+bwd): the reference-literal call site (two rasterizations, torch pre-processing of the Gaussian tensors),
the one-pass call site, and the fully fused one (`render_views_fused`).  Run on the GPU box:
    python scripts/callsite_bench.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ggrt_official_amd import splatting as sp
from ggrt_official_amd.synthetic import make_scene, CONFIGS
dev="cuda:0"
cfg=CONFIGS["C5p"]; sc=make_scene(**cfg).to(dev)
P=sc.means3D.shape[0]; H,W=sc.height,sc.width
# call-site shaped inputs: b=1
c2w=torch.eye(4,device=dev)[None]
fx=0.5/sc.tanfovx; fy=0.5/sc.tanfovy
intr=torch.tensor([[fx,0,0.5],[0,fy,0.5],[0,0,1]],device=dev)[None]
near=torch.tensor([1.0],device=dev); far=torch.tensor([100.0],device=dev)
means=sc.means3D[None].clone().requires_grad_()
cov6=sc.cov3D
cov=torch.zeros(P,3,3,device=dev)
idx=[(0,0),(0,1),(0,2),(1,1),(1,2),(2,2)]
for k,(i,j) in enumerate(idx):
    cov[:,i,j]=cov6[:,k]; cov[:,j,i]=cov6[:,k]
cov=cov[None].clone().requires_grad_()
harm=sc.shs.permute(0,2,1).contiguous()[None].clone().requires_grad_()   # [b,g,3,d]
op=sc.opacities[:,0][None].clone().requires_grad_()
bg=torch.zeros(1,3,device=dev)
dL=torch.randn(1,3,H,W,device=dev)/(3*H*W); dD=torch.randn(1,H,W,device=dev)/(H*W)
def step():
    for t in (means,cov,harm,op): t.grad=None
    c,d=sp.render_color_and_depth(c2w,intr,near,far,(H,W),bg,means,cov,harm,op,"depth")
    torch.autograd.backward([c,d],[dL,dD])
for _ in range(5): step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("call-site fwd+bwd (colour+depth fused): %.3f ms"%((time.perf_counter()-t0)/20*1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))

# ---- fused call site ----
m2=means.detach().clone().requires_grad_(); c2=cov.detach().clone().requires_grad_(); h2=harm.detach().clone().requires_grad_(); o2=op.detach().clone().requires_grad_()
gs=sp.Gaussians(means=m2,covariances=c2,harmonics=h2,opacities=o2)
def step2():
    for t in (m2,c2,h2,o2): t.grad=None
    c,d=sp.render_views_fused(c2w,intr,near,far,(H,W),bg,gs,[0],"depth")
    torch.autograd.backward([c,d],[dL,dD])
for _ in range(5): step2()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): step2()
torch.cuda.synchronize(); print("FUSED call-site fwd+bwd (colour+depth): %.3f ms"%((time.perf_counter()-t0)/20*1e3))
# reference-literal: two rasterizations (colour pass + depth pass), torch pre-processing
def step3():
    for t in (means,cov,harm,op): t.grad=None
    c=sp.render_cuda(c2w,intr,near,far,(H,W),bg,means,cov,harm,op)
    d=sp.render_depth_cuda(c2w,intr,near,far,(H,W),means,cov,op,mode="depth")
    torch.autograd.backward([c,d],[dL,dD])
for _ in range(3): step3()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): step3()
torch.cuda.synchronize(); print("reference-literal call site (2 passes): %.3f ms"%((time.perf_counter()-t0)/10*1e3))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5): step2()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=55))
