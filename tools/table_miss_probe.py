#!/usr/bin/env python3
"""Time of a launch whose shared-phase tables miss the per-init_time cache (every call brings a new init_time: a replanning episode's plans):
cfg3 DMP on its response route (k_build_shared runs the Euler map's response rows) against a cached call.   python tools/table_miss_probe.py [T]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import TrajectoryEngine, _lib  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = TrajectoryEngine(device=0, mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=T * 0.02,
                       tau=T * 0.02, alpha_phase=2.0, dmp_alpha=25.0)
B = 256
g = torch.Generator().manual_seed(0)
params = torch.randn((B, eng.num_params), generator=g).cuda()
ip = torch.zeros((B, 7), device="cuda"); iv = torch.zeros((B, 7), device="cuda")
out = (torch.empty((B, T, 7), device="cuda"), torch.empty((B, T, 7), device="cuda"))
for _ in range(20):
    eng.trajectory(params, ip, iv, 0.0, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    eng.trajectory(params, ip, iv, 0.0, out=out)
torch.cuda.synchronize()
hit = (time.perf_counter() - t0) / 200
t0 = time.perf_counter()
for i in range(200):
    eng.trajectory(params, ip, iv, 0.001 * (i + 1), out=out)
torch.cuda.synchronize()
miss = (time.perf_counter() - t0) / 200
print(f"lib {_lib.LIB_PATH}: T = {T}, {eng.last_kernel()}: cached tables {hit * 1e6:.1f} us per call, a new init_time per call {miss * 1e6:.1f} us")
