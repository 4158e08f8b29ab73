#!/usr/bin/env python3
"""For a rocprofv3 --pmc pass: the fused closed-loop step (cfg2 shape, all steps, B episodes) through k_traj_duo, k_traj_quad,
k_traj_mono and k_traj_pipe, and cfg3 (DMP) through k_traj_duo / k_traj_quad, n launches each.
   rocprofv3 --pmc ... -- python3 tools/closed_pmc_driver.py [B] [n]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="double_integrator", dt=0.02)
params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
q, qd = ip.double().contiguous(), iv.double().contiguous()
out = tuple(torch.empty((B, eng.num_steps, eng.num_dof), device=dev) for _ in range(3))
dmp = TrajectoryEngine(mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=4.0, tau=4.0,
                       alpha_phase=2.0, dmp_alpha=25.0, device=0)            # cfg3
g = torch.Generator().manual_seed(3)
dparams = torch.randn((B, dmp.num_params), generator=g).to(dev)
dout = tuple(torch.empty((B, dmp.num_steps, 7), device=dev) for _ in range(2))
cases = [("duo", {"quad": 3}), ("quad", {"quad": 2}), ("mono", {"quad": 4}), ("pipe", {"pipe": 1})]
for _ in range(3):
    eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out)
torch.cuda.synchronize()
for _ in range(n):
    for name, opts in cases:
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out)
        torch.cuda.synchronize()
    for name, opts in cases[:2]:
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        dmp.trajectory(dparams, ip, iv, out=dout)
        torch.cuda.synchronize()
_lib.reset_options()
print("order per round:", [c[0] for c in cases], eng.last_kernel())
