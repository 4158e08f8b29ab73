#!/usr/bin/env python3
"""
The planning side of the reference's TableTennis-ProDMP Replan task for a whole batch of episodes on one MI355X:
raw policy actions -> (tau / delay frozen by the first plan) -> ProDMP plans conditioned on the current state -> validity
check against the joint limits and the tau / delay bounds -> PD tracking -- one launch per plan (`mpk_replan_step_gated`,
`k_phase_fused`), the step's flags in one more (`mpk_gate_flags`).

    python examples/batched_table_tennis_plans.py [--envs 4096] [--noise 0.6] [--verbose 1]

Constants: envs/mujoco/table_tennis/mp_wrapper.py:91-121 (7 DoF, 2 basis + goal, alpha 25, learned tau in [0.8, 1.5] and
delay in [0.05, 0.15], dt 0.008, 350 steps, replanning at `t % 50 == 0`, max_planning_times 3, PD gains 0.5 [1, 4, 2, 4, 1, 4, 1] /
0.5 [0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1]) and table_tennis_utils.py:3-10 (joint limits); the invalid-plan penalty is
table_tennis_env.py:282-289.  The MuJoCo arm is replaced by the torque double integrator (the GPU-resident plant of this
repository): what is shown is the movement-primitive path, not the game.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import BatchedBlackBox  # noqa: E402
from fancy_gym_amd.black_box.factory import (get_basis_generator, get_controller, get_phase_generator,  # noqa: E402
                                             get_trajectory_generator)

P_GAINS = 0.5 * np.array([1.0, 4.0, 2.0, 4.0, 1.0, 4.0, 1.0])
D_GAINS = 0.5 * np.array([0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1])
JNT_LOW = np.array([-2.6, -2.0, -2.8, -0.9, -4.8, -1.6, -2.2])
JNT_HIGH = np.array([2.6, 2.0, 2.8, 3.1, 1.3, 1.6, 2.2])


def make_batch(envs: int, verbose: int) -> BatchedBlackBox:
    phase = get_phase_generator("exp", tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=[0.8, 1.5],
                                delay_bound=[0.05, 0.15])
    basis = get_basis_generator("prodmp", phase, num_basis=2, alpha=25.0, basis_bandwidth_factor=3.0)
    traj = get_trajectory_generator("prodmp", 7, basis, auto_scale_basis=True, goal_offset=1.0)
    ctrl = get_controller("motor", p_gains=P_GAINS, d_gains=D_GAINS)
    return BatchedBlackBox(traj, ctrl, envs, dt=0.008, duration=2.8, act_low=-1.0, act_high=1.0, plant="double_integrator",
                           replanning_every=50, max_planning_times=3, pos_limits=(JNT_LOW, JNT_HIGH), check_tau_delay=True,
                           verbose=verbose)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--noise", type=float, default=0.6, help="standard deviation of the raw weights (larger: more invalid plans)")
    ap.add_argument("--verbose", type=int, default=2, help="2: plans and step actions are returned; 1: flags and state only")
    ap.add_argument("--episodes", type=int, default=20)
    args = ap.parse_args()
    bb = make_batch(args.envs, args.verbose)
    g = torch.Generator().manual_seed(0)
    B, P = args.envs, bb.engine.num_params
    q0 = (0.2 * (torch.rand((B, 7), generator=g, dtype=torch.float64) * 2 - 1)).cuda()

    def raw_action():
        a = args.noise * torch.randn((B, P), generator=g)
        a[:, 0] = torch.rand(B, generator=g) * 0.9 + 0.7        # tau: some outside [0.8, 1.5]
        a[:, 1] = torch.rand(B, generator=g) * 0.12 + 0.04      # delay: some outside [0.05, 0.15]
        return a.cuda()

    stats = np.zeros(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.episodes):
        bb.reset(q0)
        for k in range(3):
            out = bb.step(raw_action())
            stats += [float(out["valid"].float().mean()), float(out["terminated"].float().mean()),
                      float(out["invalid_penalty"].mean()), float(out["trajectory_length"].float().mean())]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats /= 3 * args.episodes
    print(f"{args.episodes} batches of {B} episodes x 3 plans in {dt:.2f} s = {args.episodes * B / dt:.3e} episodes/s (eager, random actions drawn on the host)")
    print(f"per plan: {100 * stats[0]:.1f} % valid, {100 * stats[1]:.1f} % of the episodes terminated by an invalid plan, mean penalty {stats[2]:.4f}, "
          f"mean executed steps {stats[3]:.1f}; kernel of the last plan: {bb.engine.last_kernel()}")
    if args.verbose >= 2:
        print("returned per plan:", {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v) and v.dim() >= 2})


if __name__ == "__main__":
    main()
