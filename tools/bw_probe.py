"""Practical HBM ceilings on the box: pure-write (fill), copy, pure-read (sum) with torch's vectorised kernels."""
import torch, time, sys
torch.cuda.set_device(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (35, 560, 4096, 9000):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    t = timeit(lambda: x.fill_(1.0)); print(f"{mb:6d} MB fill  {n*4/t/1e9:8.0f} GB/s")
    t = timeit(lambda: x.zero_()); print(f"{mb:6d} MB zero  {n*4/t/1e9:8.0f} GB/s")
    t = timeit(lambda: y.copy_(x)); print(f"{mb:6d} MB copy  {2*n*4/t/1e9:8.0f} GB/s (r+w)")
    t = timeit(lambda: x.sum()); print(f"{mb:6d} MB sum   {n*4/t/1e9:8.0f} GB/s (read)")
    del x, y
