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


def main():
    pos_args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(pos_args[0]) if len(pos_args) > 0 else 8192
    episodes = int(pos_args[1]) if len(pos_args) > 1 else 50
    torch.cuda.set_device(0)
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
