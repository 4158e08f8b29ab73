#!/usr/bin/env python3
"""per-wave cycle budget of k_traj_wide (build with MPK_EXTRA_FLAGS=-DWIDE_TIME, load with MPK_LIB): python tools/dev/wide_time.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fancy_gym_amd import TrajectoryEngine
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = TrajectoryEngine("promp", "linear", "zero_rbf", 5, 1000, dt=0.02, duration=4.0, device=0, tau=4.0, num_basis_zero_start=1, num_basis_zero_goal=0)
g = torch.Generator().manual_seed(0)
params = torch.randn((B, eng.num_params), generator=g).to(dev)
ip = torch.zeros((B, 5), device=dev); iv = torch.zeros((B, 5), device=dev)
out = (torch.empty((B, 200, 5), device=dev), torch.empty((B, 200, 5), device=dev))
for _ in range(3):
    eng.trajectory(params, ip, iv, 0.0, out=out)
torch.cuda.synchronize()
v = out[1].flatten()[:512 * 4 * 8].cpu().numpy().reshape(-1, 8)
v = v[v[:, 7] > 0]
n = int(v[0, 7]) * 4
v = v[:n]
print(f"{n} waves ({int(v[0, 7])} workgroups); shader-clock cycles per wave (s_memtime):")
for name, i in (("total", 0), ("contraction", 1), ("barrier + commit + barrier", 2), ("fetch issue", 3), ("epilogue", 4)):
    c = v[:, i]
    print(f"  {name:28s} mean {c.mean():12.0f}  min {c.min():12.0f}  max {c.max():12.0f}  share of total {c.mean() / v[:, 0].mean() * 100:5.1f} %")
three = v[v[:, 5] < (1366 - 2 * int(v[0, 7]))]
print(f"  workgroups with 3 units: {len(three) // 4}; their total {three[:, 0].mean():.0f} vs others {v[v[:, 5] >= (1366 - 2 * int(v[0, 7]))][:, 0].mean():.0f}")
