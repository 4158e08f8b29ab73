#!/usr/bin/env python3
"""mpk_pd_rollout (cfg2 shape) launched n times at batch B for a rocprofv3 --pmc pass:  ... -- python3 tools/rollout_pmc_driver.py [B] [n] [key=value ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from tools.closed_bench import CFG2, DG, PG  # noqa: E402

args = [a for a in sys.argv[1:] if "=" not in a]
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
B = int(args[0]) if args else 4096
n = int(args[1]) if len(args) > 1 else 30
torch.cuda.set_device(0)
eng = TrajectoryEngine(device=0, **CFG2)
g = torch.Generator().manual_seed(0)
params = torch.randn((B, eng.num_params), generator=g).cuda()
ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda()
iv = torch.zeros((B, 7), device="cuda")
pos, vel = eng.trajectory(params, ip, iv, 0.0)
spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
q, qd = ip.double().contiguous(), iv.double().contiguous()
act = torch.empty((B, 100, 7), device="cuda")
for _ in range(n):
    eng.pd_rollout(spec, pos, vel, q, qd, out=act)
torch.cuda.synchronize()
print("done")
