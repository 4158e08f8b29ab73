#!/usr/bin/env python3
"""
cfg4 end to end on one GPU: fancy_ProDMP/BoxPushingDenseReplan-style episodes (4 plans x 25 steps of a 100-step horizon,
schedule t % 25 == 0, condition_on_desired, P = 35) for B episodes with the reference's torque double integrator as the
GPU-resident plant, through BatchedBlackBox (integer state kernel + one fused plan/execute launch per plan).
    python tools/bench_replan.py [B] [episodes] [--graph]
"""
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

PG = 0.01 * np.array([120., 120., 120., 120., 50., 30., 10.])
DG = 0.01 * np.array([10., 10., 10., 10., 6., 5., 3.])


TT_P = 0.5 * np.array([1.0, 4.0, 2.0, 4.0, 1.0, 4.0, 1.0])
TT_D = 0.5 * np.array([0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1])
JNT_LOW = np.array([-2.6, -2.0, -2.8, -0.9, -4.8, -1.6, -2.2])      # table_tennis_utils.py:3-4
JNT_HIGH = np.array([2.6, 2.0, 2.8, 3.1, 1.3, 1.6, 2.2])


def table_tennis(B, episodes):
    """--tt [--gate] [--verbose1]: TableTennis-ProDMP Replan (envs/mujoco/table_tennis/mp_wrapper.py:91-121: learn tau + delay, 2 basis +
    goal, `t % 50 == 0`, max_planning_times 3, T = 350) with the double integrator as the plant: three plans per episode, tau / delay
    frozen by the first; --gate: joint limits + tau / delay bounds inside the launch (table_tennis_env.py:282-309)"""
    pg = get_phase_generator("exp", tau=1.5, alpha_phase=3, learn_tau=True, learn_delay=True, tau_bound=[0.8, 1.5], delay_bound=[0.05, 0.15])
    bg = get_basis_generator("prodmp", pg, num_basis=2, alpha=25, basis_bandwidth_factor=3)
    tg = get_trajectory_generator("prodmp", 7, bg, auto_scale_basis=True, goal_offset=1.0)
    gate = "--gate" in sys.argv
    bb = BatchedBlackBox(tg, get_controller("motor", p_gains=TT_P, d_gains=TT_D), B, 0.008, 2.8, act_low=-1.0, act_high=1.0,
                         plant="double_integrator", replanning_every=50, max_planning_times=3, condition_on_desired=False,
                         pos_limits=(JNT_LOW, JNT_HIGH) if gate else None, check_tau_delay=gate,
                         verbose=1 if "--verbose1" in sys.argv else 2)
    g = torch.Generator().manual_seed(0)
    q0 = (0.2 * (torch.rand((B, 7), generator=g, dtype=torch.float64) * 2 - 1)).cuda()
    P = bb.engine.num_params
    plans = []
    for _ in range(3):
        p = (1.0 if "--wild" in sys.argv else 0.3) * torch.randn((B, P), generator=g)     # --wild: a third of the plans leaves the joint limits
        p[:, 0] = torch.rand(B, generator=g) * 0.6 + 0.85
        p[:, 1] = torch.rand(B, generator=g) * 0.08 + 0.06
        plans.append(p.cuda())

    def episode():
        bb.reset(q0)
        for k in range(3):
            out = bb.step(plans[k])
        return out

    if "--graph" in sys.argv:
        ep = bb.capture_episode(3)
        ep.init_pos.copy_(q0)
        for k in range(3):
            ep.params[k].copy_(plans[k])

        def episode():     # noqa: F811
            return ep.replay()[-1]
    for _ in range(3):
        out = episode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(episodes):
        episode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / episodes
    valid = out.get("valid")
    print(f"TableTennis-ProDMP Replan episodes{' + validity gate' if gate else ''}{', verbose < 2' if '--verbose1' in sys.argv else ''}, B = {B}: "
          f"{dt * 1e3:.3f} ms per batch of episodes (3 plans) = {B / dt:.3e} episodes/s; kernel of the last plan: {bb.engine.last_kernel()}"
          f"{' (one hipGraph per episode batch)' if '--graph' in sys.argv else ''}"
          f"{'' if valid is None else f'; {int(valid.sum())} of {B} last plans valid'}")


def main():
    pos_args = [a for a in sys.argv[1:] if not a.startswith("--") and "=" not in a]
    for kv in sys.argv[1:]:
        if "=" in kv and not kv.startswith("--"):     # kernel-selection overrides: key=value (mpk_set_option)
            from fancy_gym_amd import _lib
            _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    B = int(pos_args[0]) if len(pos_args) > 0 else 8192
    episodes = int(pos_args[1]) if len(pos_args) > 1 else 50
    torch.cuda.set_device(0)
    if "--tt" in sys.argv:
        return table_tennis(B, episodes)
    pg = get_phase_generator("exp", tau=1.5, alpha_phase=3)
    bg = get_basis_generator("prodmp", pg, num_basis=5, alpha=10, basis_bandwidth_factor=3)
    tg = get_trajectory_generator("prodmp", 7, bg, weights_scale=0.3, goal_scale=0.3, auto_scale_basis=True,
                                  goal_offset=1.0, disable_goal=True)
    bb = BatchedBlackBox(tg, get_controller("motor", p_gains=PG, d_gains=DG), B, 0.02, 2.0, act_low=-1.0, act_high=1.0,
                         plant="double_integrator", replanning_every=25, max_planning_times=4,
                         condition_on_desired=True)
    g = torch.Generator().manual_seed(0)
    q0 = (torch.rand((B, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda()
    plans = [torch.randn((B, 35), generator=g).cuda() for _ in range(4)]

    def episode():
        bb.reset(q0)
        for k in range(4):
            out = bb.step(plans[k])
        return out

    if "--graph" in sys.argv:
        ep = bb.capture_episode(4)
        ep.init_pos.copy_(q0)
        for k in range(4):
            ep.params[k].copy_(plans[k])

        def episode():     # noqa: F811 - one graph launch per episode batch
            return ep.replay()[-1]
    for _ in range(3):
        out = episode()
    assert bool(out["done"].all()) and int(bb.traj_steps[0]) == 100
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(episodes):
        episode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / episodes
    print(f"cfg4 replanning episodes, B = {B}: {dt * 1e3:.3f} ms per batch of episodes (4 plans) = {B / dt:.3e} episodes/s"
          f" = {4 * B / dt:.3e} plans/s; kernel of the last plan: {bb.engine.last_kernel()}"
          f"{' (one hipGraph per episode batch)' if '--graph' in sys.argv else ''}")


if __name__ == "__main__":
    main()
