#!/usr/bin/env python3
"""
Episode-based search on the reference's SimpleReacher family, entirely on one MI355X: a population of ProDMP parameter
vectors -> trajectories -> PD tracking on the torque plant -> per-step reward -> episode return, one hipGraph replay per
generation (`BatchedBlackBox.capture_episode`; with `verbose=1` the whole generation is ONE kernel that stores nothing per step:
`mpk_episode_return`), cross-entropy update on the host.

    python examples/batched_reacher_search.py [--pop 4096] [--iters 30] [--links 5]

The environment constants are those of `fancy_ProDMP/LongSimpleReacher-v0` (fancy_gym/envs/__init__.py:48-56,
classic_control/simple_reacher/mp_wrapper.py): dt = 0.01, 200 steps, reward = -(distance to the goal from step 199 on)
- sum(action^2).
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


def make_population(links: int, pop: int) -> BatchedBlackBox:
    phase = get_phase_generator("exp", tau=2.0, alpha_phase=3.0)
    basis = get_basis_generator("prodmp", phase, num_basis=5, alpha=10.0, basis_bandwidth_factor=3.0)
    traj = get_trajectory_generator("prodmp", links, basis, weights_scale=1.0, goal_scale=1.0)
    ctrl = get_controller("motor", p_gains=0.6, d_gains=0.075)
    # verbose = 1 (the reference's default, black_box_wrapper.py:21): a step returns the episode return, not the trajectories --
    # on the device ONE launch per generation that stores nothing per step (mpk_episode_return)
    return BatchedBlackBox(traj, ctrl, pop, dt=0.01, duration=2.0, act_low=-1000.0, act_high=1000.0,
                           plant="double_integrator", reward="simple_reacher", verbose=1)


def search(pop: int = 4096, iters: int = 30, links: int = 5, seed: int = 0, verbose: bool = True):
    bb = make_population(links, pop)
    P = bb.engine.num_params
    episode = bb.capture_episode(1)
    start = np.zeros(links); start[0] = np.pi / 2                  # the arm points straight up (base_reacher.py:33)
    episode.init_pos.copy_(torch.tensor(start).expand(pop, links))
    episode.goal.copy_(torch.tensor([0.6 * links, 0.4 * links], dtype=torch.float64).expand(pop, 2))
    g = torch.Generator(device="cpu").manual_seed(seed)
    mean, std = torch.zeros(P), torch.ones(P)
    history = []
    t0 = time.perf_counter()
    for it in range(iters):
        cand = mean + std * torch.randn((pop, P), generator=g)
        cand[0] = mean
        episode.params[0].copy_(cand)
        out = episode.replay()[0]
        ret = out["rewards"].cpu()
        elite = ret.topk(max(pop // 16, 4)).indices
        mean, std = cand[elite].mean(0), cand[elite].std(0) + 1e-3
        history.append((float(ret.max()), float(ret.mean())))
        if verbose:
            print(f"generation {it:3d}: best return {history[-1][0]:9.3f}   population mean {history[-1][1]:12.3f}")
    torch.cuda.synchronize()
    if verbose:
        dt = time.perf_counter() - t0
        print(f"{iters} generations x {pop} episodes x 200 steps in {dt:.2f} s = {iters * pop / dt:.3e} episodes/s "
              f"(host CEM update included)")
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--links", type=int, default=5)
    a = ap.parse_args()
    search(a.pop, a.iters, a.links)
