#!/usr/bin/env python3
"""cfg5's shape (ProMP 7 x 3 x 350 + PD actions, T D = 2 mod 4) around the ring's threshold: automatic choice against ring off / on.
    python tools/cfg5_choice.py [B ...]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import graph_time  # noqa: E402
from run_cfg import KW  # noqa: E402
import numpy as np  # noqa: E402
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(device=0, **KW["cfg5"])
D, T = eng.num_dof, eng.num_steps
spec = RolloutSpec("motor", D, np.full(D, 1.0), np.full(D, 0.1), -1.0, 1.0, plant="static")
print("| B | options | kernel | us | of 8 TB/s |")
for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [12288, 16384, 20480, 32768]:
    params = torch.randn((B, eng.num_params), generator=g).cuda()
    ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, D), device="cuda")
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, T, D), device="cuda") for _ in range(3))
    for opts in ({}, {"ring": 0}, {"ring": 1}):
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        t = graph_time(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out), reps=10, rounds=5)
        print(f"| {B} | {opts or 'auto'} | `{eng.last_kernel()}` | {t * 1e6:.1f} | {B * (140 + 3 * T * D * 4) / t / 8e12 * 100:.1f} % |", flush=True)
    _lib.reset_options()
    del out
