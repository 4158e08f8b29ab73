#!/usr/bin/env python3
"""Run one BASELINE configuration N times (for rocprofv3):  python tools/run_cfg.py <cfg2|cfg2act|cfg2closed|cfg3|cfg4|cfg5|cfg5tau|cfg2tau|ttprodmp[act|closed]|beerpong[act|closed]> B N [option=value ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import RolloutSpec, TrajectoryEngine  # noqa: E402

PG = 0.01 * np.array([120., 120., 120., 120., 50., 30., 10.])
DG = 0.01 * np.array([10., 10., 10., 10., 6., 5., 3.])
TT_P = 0.5 * np.array([1.0, 4.0, 2.0, 4.0, 1.0, 4.0, 1.0])
TT_D = 0.5 * np.array([0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1])
KW = {
    "cfg2": dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=5, dt=0.02,
                 duration=2.0, tau=1.5, alpha_phase=3.0, basis_bandwidth_factor=2.0, basis_alpha=10.0),
    "cfg3": dict(mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=4.0,
                 tau=4.0, alpha_phase=2.0, dmp_alpha=25.0),
    "cfg4": dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=5, dt=0.02,
                 duration=2.0, tau=1.5, alpha_phase=3.0, basis_bandwidth_factor=3.0, basis_alpha=10.0,
                 weights_scale=0.3, goal_scale=0.3, auto_scale_basis=True, disable_goal=True),
    "cfg5": dict(mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=7, num_basis=3,
                 num_basis_zero_start=1, num_basis_zero_goal=1, dt=0.008, duration=2.8, tau=2.8),
    # per-episode phase (learned tau / delay): the k_traj_phase kernels
    "cfg5tau": dict(mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=7, num_basis=3,
                    num_basis_zero_start=1, num_basis_zero_goal=1, dt=0.008, duration=2.8, tau=2.8, learn_tau=True,
                    learn_delay=True, tau_bound=(0.5, 2.8), delay_bound=(0.05, 0.15)),
    # the reference's learned-phase families at their real shapes (k_phase_fused: tools/learned_phase_bench.py has the citations)
    "ttprodmp": dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=3, dt=0.008, duration=2.8,
                     tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.8, 1.5), delay_bound=(0.05, 0.15),
                     basis_alpha=25.0, basis_bandwidth_factor=3.0, weights_scale=0.7, auto_scale_basis=True, relative_goal=True,
                     disable_goal=True),
    "beerpong": dict(mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=7, num_basis=2, num_basis_zero_start=2,
                     dt=0.01, duration=3.0, tau=3.0, learn_tau=True, tau_bound=(0.02, 3.0), basis_bandwidth_factor=3.0),
    "cfg2tau": dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=5, dt=0.02,
                    duration=2.0, tau=1.5, alpha_phase=3.0, basis_bandwidth_factor=2.0, basis_alpha=10.0,
                    learn_tau=True, tau_bound=(0.5, 2.0)),
}


def main():
    name, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from fancy_gym_amd import _lib
    for kv in sys.argv[4:]:                     # kernel-selection overrides: key=value (mpk_set_option)
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    base = name.replace("act", "").replace("closed", "")
    torch.cuda.set_device(0)
    eng = TrajectoryEngine(device=0, **KW[base])
    T, D, P = eng.num_steps, eng.num_dof, eng.num_params
    g = torch.Generator().manual_seed(0)
    params = torch.randn((B, P), generator=g)
    if base.endswith("tau"):
        n_ph = P - D * ((P - 0) // D)
        params[:, :n_ph] = torch.rand((B, n_ph), generator=g) * 0.5 + 0.8
    if base == "ttprodmp":
        params *= 0.3
        params[:, 0] = torch.rand(B, generator=g) * 0.7 + 0.8
        params[:, 1] = torch.rand(B, generator=g) * 0.1 + 0.05
    if base == "beerpong":
        params *= 0.3
        params[:, 0] = torch.rand(B, generator=g) * 2.0 + 1.0
    params = params.cuda()
    ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
    iv = torch.zeros((B, D), device="cuda")
    gains = (TT_P, TT_D) if base in ("cfg5", "ttprodmp", "beerpong") else (PG, DG)
    out = tuple(torch.empty((B, T, D), device="cuda") for _ in range(3))
    q, qd = ip.double().contiguous(), iv.double().contiguous()
    for _ in range(N):
        if name.endswith("closed"):
            spec = RolloutSpec("motor", D, gains[0], gains[1], -1.0, 1.0, plant="double_integrator", dt=KW[base]["dt"])
            eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out)
        elif name.endswith("act"):
            spec = RolloutSpec("motor", D, gains[0], gains[1], -1.0, 1.0, plant="static")
            eng.trajectory_actions(params, ip, iv, spec, q, qd, out=out)
        else:
            eng.trajectory(params, ip, iv, 0.0, out=out[:2])
    torch.cuda.synchronize()
    print(name, B, N, eng.last_kernel())


if __name__ == "__main__":
    main()
