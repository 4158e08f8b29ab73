#!/usr/bin/env python3
"""Launches, for a rocprofv3 --pmc pass, the three things the streaming row compares -- the fused episode-major kernel
(trajectory + actions, B episodes), the same kernel without actions, and a plain torch fill of the same output arrays --
`n` times each after a warm-up.     rocprofv3 --pmc ... -- python3 tools/stream_pmc_driver.py [B] [n]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
cp, cv = ip.double().contiguous(), iv.double().contiguous()
out = tuple(torch.empty((B, eng.num_steps, eng.num_dof), device=dev) for _ in range(3))
for _ in range(3):
    eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)
torch.cuda.synchronize()
for _ in range(n):
    eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)
    eng.trajectory(params, ip, iv, 0.0, out=out[:2])
    for o in out:
        o.fill_(1.0)
torch.cuda.synchronize()
print("done", eng.last_kernel())
