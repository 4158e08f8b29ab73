#!/usr/bin/env python3
"""
The rollout kernels on trajectories that already exist (mpk_pd_rollout / mpk_reacher_rollout): cfg2 shape (7 DoF, 100 steps,
PD gains of box_pushing/mp_wrapper.py:12-13, torque double integrator) and LongSimpleReacher (5 DoF, 200 steps, + reward).
Launches shorter than 400 us are timed as a captured graph of 20.  Algorithmic bytes: 2 T D 4 read + T D 4 (+ T 8) written.
    python tools/rollout_bench.py [B ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from tools.closed_bench import CFG2, DG, PG, graph_time  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 8192, 65536]
    extra = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("--"))}     # options for every row
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(0)
    print(f"lib: {_lib.LIB_PATH}")
    print("| rollout | B | options | us | episodes/s | GB/s (alg.) | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    eng2 = TrajectoryEngine(device=0, **CFG2)
    engr = TrajectoryEngine(device=0, mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=5, num_basis=5,
                            num_basis_zero_start=1, dt=0.01, duration=2.0, tau=2.0)
    for B in batches:
        for name, eng, D, T, spec, reward in (
                ("cfg2 PD + double integrator (7 x 100)", eng2, 7, 100,
                 RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02), False),
                ("LongSimpleReacher PD + plant (5 x 200)", engr, 5, 200,
                 RolloutSpec("motor", 5, 0.6, 0.075, -1000.0, 1000.0, plant="double_integrator", dt=0.01), False),
                ("LongSimpleReacher PD + plant + reward (5 x 200)", engr, 5, 200,
                 RolloutSpec("motor", 5, 0.6, 0.075, -1000.0, 1000.0, plant="double_integrator", dt=0.01), True)):
            params = torch.randn((B, eng.num_params), generator=g).cuda()
            ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
            iv = torch.zeros((B, D), device="cuda")
            pos, vel = eng.trajectory(params, ip, iv, 0.0)
            q, qd = ip.double().contiguous(), iv.double().contiguous()
            act = torch.empty((B, T, D), device="cuda")
            rew = torch.empty((B, T), dtype=torch.float64, device="cuda")
            goal = (torch.rand((B, 2), generator=g, dtype=torch.float64) * 4 - 2).cuda()
            nbytes = 3 * T * D * 4 + (T * 8 if reward else 0)
            if reward:
                fn = lambda: eng.reacher_rollout(spec, pos, vel, q, qd, goal, out=(act, rew))     # noqa: E731
            else:
                fn = lambda: eng.pd_rollout(spec, pos, vel, q, qd, out=act)                       # noqa: E731
            variants = ({}, {"pd_quad": 0}, {"pd_quad": 3}, {"pd_quad": 2})
            if "--waves" in sys.argv:   # groups per wave x resident workgroups per CU of the large launches
                variants = tuple(dict(pd_quad=q_, **({"phase_waves": w} if w else {})) for q_ in (2, 3) for w in (0, 4, 8, 16))
            if "--wt" in sys.argv:      # store policy A/B of the automatic geometry
                variants = ({}, {"write_through": 0}, {"write_through": 1}, {"pd_quad": 0, "write_through": 0}, {"pd_quad": 3, "write_through": 0})
            for opts in variants:
                _lib.reset_options()
                for k, v in list(extra.items()) + list(opts.items()):
                    _lib.set_option(k, v)
                t = graph_time(fn)
                print(f"| {name} | {B} | {opts or 'auto'} | {t * 1e6:.1f} | {B / t:.3e} | {B * nbytes / t / 1e9:.0f} | "
                      f"{B * nbytes / t / 8e12 * 100:.1f} % |")
            _lib.reset_options()
            del params, ip, iv, pos, vel, q, qd, act, rew


if __name__ == "__main__":
    main()
