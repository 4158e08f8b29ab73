#!/usr/bin/env python3
"""Throughput of the per-episode-phase path (learned tau / delay): python tools/rows_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import TrajectoryEngine  # noqa: E402
from tools.sweep import ev_time, row  # noqa: E402

CFGS = [
    ("cfg5 ProMP TT learn tau+delay", dict(mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=7,
     num_basis=3, num_basis_zero_start=1, num_basis_zero_goal=1, dt=0.008, duration=2.8, tau=2.8, learn_tau=True,
     learn_delay=True, tau_bound=(0.5, 2.8), delay_bound=(0.05, 0.15)), [1024, 8192, 65536]),
    ("cfg2 ProDMP learn tau", dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=5,
     dt=0.02, duration=2.0, tau=1.5, alpha_phase=3.0, basis_bandwidth_factor=2.0, basis_alpha=10.0, learn_tau=True,
     tau_bound=(0.5, 2.0)), [4096, 65536]),
    ("cfg3 DMP learn tau", dict(mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02,
     duration=4.0, tau=4.0, alpha_phase=2.0, dmp_alpha=25.0, learn_tau=True, tau_bound=(1.0, 4.0)), [16384]),
]


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    for name, kw, batches in CFGS:
        eng = TrajectoryEngine(device=0, **kw)
        T, D, P = eng.num_steps, eng.num_dof, eng.num_params
        for B in batches:
            params = torch.randn((B, P), generator=g)
            n_ph = int(kw.get("learn_tau", False)) + int(kw.get("learn_delay", False))
            params[:, :n_ph] = torch.rand((B, n_ph), generator=g) * 0.5 + 0.8
            params = params.to(dev)
            ip = (torch.rand((B, D), generator=g) * 2 - 1).to(dev)
            iv = torch.zeros((B, D), device=dev)
            out = (torch.empty((B, T, D), device=dev), torch.empty((B, T, D), device=dev))
            t = ev_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out), n=20, warm=3)
            row(name, B, T, D, P, t, P * 4 + 2 * D * 4 + 2 * T * D * 4, eng.last_kernel())


if __name__ == "__main__":
    main()
