"""ramp_probe.py — what the first ≈ 25 steps after an idle device pay (NOTES.md r4, "cold-start ramp").

Per-step HIP-event times of the C3 step after: 1 s of idle; the same preceded by ≈ 30 ms of memory-bound / ALU-bound
filler queued right in front of it; and after idle gaps of 1 … 100 ms inside a warm loop.  Dev tool, not a test.
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from ggrt_official_amd import synthetic

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload("C3", synthetic.CONFIGS["C3"], dev, seed=0)
for _ in range(3):
    wl.step()
torch.cuda.synchronize(dev)


def series(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        wl.step()
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    return [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(n)]


a = torch.empty(128 << 20, device=dev)   # 512 MiB
b = torch.empty_like(a)
m1 = torch.randn(8192, 8192, device=dev)
m2 = torch.randn(8192, 8192, device=dev)
torch.mm(m1, m2); b.copy_(a); torch.cuda.synchronize(dev)


def filler_mem(ms):
    for _ in range(int(ms / 0.2)):
        b.copy_(a)


def filler_alu(ms):
    for _ in range(max(1, int(ms / 12))):
        torch.mm(m1, m2)


def show(label, ms):
    print(f"{label:34s} first8 {ms[:8]}  9-16 med {sorted(ms[8:16])[4]:.3f}  17-24 {sorted(ms[16:24])[4]:.3f}  "
          f"25-32 {sorted(ms[24:32])[4]:.3f}  33-40 {sorted(ms[32:40])[4]:.3f}  last8 {sorted(ms[-8:])[4]:.3f}", flush=True)


for rep in range(2):
    time.sleep(1.0); show("idle 1 s", series(60))
    time.sleep(1.0); filler_mem(30); show("idle 1 s + 30 ms copies", series(60))
    time.sleep(1.0); filler_alu(30); show("idle 1 s + 30 ms sgemm", series(60))
    time.sleep(1.0); filler_alu(100); show("idle 1 s + 100 ms sgemm", series(60))
series(100)
for gap in (0.001, 0.003, 0.01, 0.03, 0.1, 0.3):
    series(60); torch.cuda.synchronize(dev); time.sleep(gap)
    show(f"warm, then idle {gap * 1e3:.0f} ms", series(60))
# step-count or time?  the same ramp with a 4x cheaper step (C2-size frame)
wl_small = bench.Workload("C2", synthetic.CONFIGS["C2"], dev, seed=0)
time.sleep(1.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
ev[0].record()
for i in range(200):
    wl_small.step(); ev[i + 1].record()
torch.cuda.synchronize(dev)
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(200)]
print("C2 after idle: steps 1-10 med %.4f, 41-50 %.4f, 91-100 %.4f, 191-200 %.4f; cumulative ms at step 50: %.1f" % (
    sorted(ms[:10])[5], sorted(ms[40:50])[5], sorted(ms[90:100])[5], sorted(ms[190:])[5], sum(ms[:50])))
