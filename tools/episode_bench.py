#!/usr/bin/env python3
"""
The verbose < 2 step (mpk_episode_return: plan + controller + plant + reward + aggregation in one launch, nothing per step stored)
beside the verbose = 2 launches that compute the same step and materialise pos / vel / actions (/ step rewards):
  cfg2 shape, full horizon (7 x 100, no device reward)        mpk_episode_return   vs  mpk_trajectory_rollout
  cfg4 replanning step (25 of 100 steps, condition gather)     mpk_episode_return   vs  mpk_replan_step
  LongSimpleReacher (5 x 200) + SimpleReacher reward, sum      mpk_episode_return   vs  mpk_trajectory + mpk_reacher_rollout + mpk_reward_aggregate
What bounds the fused launch is the float64 chain, not HBM: the table reports episodes/s and, for orientation only, the bytes the
verbose = 2 path moves per episode.   python tools/episode_bench.py [B ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from tools.closed_bench import CFG2, DG, PG, graph_time  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 16384, 65536]
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(0)
    print(f"lib: {_lib.LIB_PATH}")
    print("| step | B | path | kernel(s) | us | episodes/s | speed-up |")
    print("|---|---|---|---|---|---|---|")
    eng2 = TrajectoryEngine(device=0, **CFG2)
    engr = TrajectoryEngine(device=0, mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=5, num_basis=5,
                            num_basis_zero_start=1, dt=0.01, duration=2.0, tau=2.0)
    spec2 = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
    specr = RolloutSpec("motor", 5, 0.6, 0.075, -1000.0, 1000.0, plant="double_integrator", dt=0.01)
    for B in batches:
        i32 = dict(dtype=torch.int32, device="cuda")

        def state():
            return torch.zeros(B, **i32), torch.zeros(B, **i32), torch.zeros(B, dtype=torch.uint8, device="cuda")
        for name, eng, D, T, spec, mode in (("cfg2 full horizon (7 x 100)", eng2, 7, 100, spec2, "full"),
                                            ("cfg2 shape, replanning step (25 of 100)", eng2, 7, 100, spec2, "replan"),
                                            ("LongSimpleReacher + reward, sum (5 x 200)", engr, 5, 200, specr, "reward")):
            params = torch.randn((B, eng.num_params), generator=g).cuda()
            ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
            iv = torch.zeros((B, D), device="cuda")
            q, qd = ip.double().contiguous(), iv.double().contiguous()
            goal = (torch.rand((B, 2), generator=g, dtype=torch.float64) * 4 - 2).cuda()
            out = tuple(torch.empty((B, T, D), device="cuda") for _ in range(3))
            rew = torch.empty((B, T), dtype=torch.float64, device="cuda")
            ts, ps, dn = state()
            if mode == "full":
                lean = lambda: eng.episode_return(params, ip, iv, spec, q, qd)                                        # noqa: E731
                full = lambda: eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out)                           # noqa: E731
            elif mode == "replan":
                def lean():
                    ts.zero_(); ps.zero_(); dn.zero_()
                    return eng.episode_return(params, ip, iv, spec, q, qd, replan=(ts, ps, dn, 25, 4, 100), condition=True)

                def full():
                    ts.zero_(); ps.zero_(); dn.zero_()
                    return eng.replan_step(params, ip, iv, spec, q, qd, ts, ps, dn, 25, 4, 100, condition=True, out=out)
            else:
                lean = lambda: eng.episode_return(params, ip, iv, spec, q, qd, reward="simple_reacher", goal=goal)    # noqa: E731

                def full():
                    eng.trajectory(params, ip, iv, 0.0, out=out[:2])
                    eng.reacher_rollout(spec, out[0], out[1], q, qd, goal, out=(out[2], rew))
                    return eng.reward_aggregate(rew, torch.full((B,), T, **i32), "sum")
            times = {}
            for rnd in range(2):
                for path, fn in (("verbose < 2", lean), ("verbose = 2", full)):
                    fn(); torch.cuda.synchronize()
                    kern = eng.last_kernel()
                    times.setdefault(path, []).append((graph_time(fn), kern))
            tl, tf = min(t for t, _ in times["verbose < 2"]), min(t for t, _ in times["verbose = 2"])
            for path, t in (("verbose < 2", tl), ("verbose = 2", tf)):
                print(f"| {name} | {B} | {path} | `{times[path][0][1]}` | {t * 1e6:.1f} | {B / t:.3e} | "
                      f"{tf / t:.2f} x |", flush=True)


if __name__ == "__main__":
    main()
