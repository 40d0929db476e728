import torch, time
dev=torch.device('cuda:0')
for _ in range(200): torch.mm(torch.randn(2048,2048,device=dev), torch.randn(2048,2048,device=dev))
torch.cuda.synchronize()
for mb in (16, 32, 64, 128, 192, 232, 256, 384, 512, 1024, 2048):
    n = mb*1024*1024//4
    x = torch.ones(n, device=dev)
    for _ in range(20): x.sum()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    reps=50
    e0.record()
    for _ in range(reps): x.sum()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps
    print(f"read {mb:5d} MB repeatedly: {ms*1e3:8.1f} us  {mb*1.048576/ms:8.1f} GB/s", flush=True)
