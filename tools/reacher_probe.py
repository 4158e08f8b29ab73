#!/usr/bin/env python3
"""Throughput of the SimpleReacher device rollout (controller + torque plant + reward): python tools/reacher_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import RolloutSpec, TrajectoryEngine  # noqa: E402
from tools.sweep import ev_time, row  # noqa: E402


def main():
    torch.cuda.set_device(0)
    D, T = 5, 200
    eng = TrajectoryEngine(device=0, mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=D, num_basis=5,
                           num_basis_zero_start=1, dt=0.01, duration=2.0, tau=2.0)
    spec = RolloutSpec("motor", D, 0.6, 0.075, -1000.0, 1000.0, plant="double_integrator", dt=0.01)
    g = torch.Generator().manual_seed(0)
    for B in (4096, 65536):
        params = torch.randn((B, eng.num_params), generator=g).cuda()
        ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
        iv = torch.zeros((B, D), device="cuda")
        pos, vel = eng.trajectory(params, ip, iv, 0.0)
        goal = (torch.rand((B, 2), generator=g, dtype=torch.float64) * 4 - 2).cuda()
        q0, qd0 = ip.double().contiguous(), iv.double().contiguous()

        q, qd = q0.clone(), qd0.clone()
        bufs = (torch.empty((B, T, D), device="cuda"), torch.empty((B, T), dtype=torch.float64, device="cuda"))

        def run():      # the state keeps integrating from launch to launch: same work per launch, no reset kernels
            eng.reacher_rollout(spec, pos, vel, q, qd, goal, out=bufs)

        def run_pd():
            eng.pd_rollout(spec, pos, vel, q, qd, out=bufs[0])
        t = ev_time(run, n=20, warm=3)
        row("LongSimpleReacher rollout + reward (D=5, T=200)", B, T, D, 0, t, 3 * T * D * 4 + T * 8 + 4 * D * 8, "k_reacher_rollout")
        t = ev_time(run_pd, n=20, warm=3)
        row("same without the reward", B, T, D, 0, t, 3 * T * D * 4 + 4 * D * 8, "k_pd_rollout")


if __name__ == "__main__":
    main()
