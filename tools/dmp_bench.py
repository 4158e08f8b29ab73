#!/usr/bin/env python3
"""
cfg3 (DMP 7 DoF x 5 basis x 200 steps, shared phase) through mpk_trajectory: the response route (contraction of the Euler map's
response rows on the matrix-core kernels, round 5) against the serial explicit-Euler kernels ("dmp_response" 0), alternating in one
process.  Also the fused actions and the closed-loop step, which DMP has on the response route only (two launches before).
Algorithmic bytes 11 424 per trajectory (+ 5 600 actions).   python tools/dmp_bench.py [B ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from tools.closed_bench import graph_time  # noqa: E402

CFG3 = dict(mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=4.0, tau=4.0,
            alpha_phase=2.0, dmp_alpha=25.0)


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 16384, 65536, 262144]
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(0)
    print(f"lib: {_lib.LIB_PATH}")
    print("| launch | B | route | kernel | us | trajectories/s | GB/s (alg.) | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    eng = TrajectoryEngine(device=0, **CFG3)
    T, D = eng.num_steps, 7
    spec_s = RolloutSpec("motor", 7, 1.0, 0.1, -1.0, 1.0, plant="static")
    spec_d = RolloutSpec("motor", 7, 1.0, 0.1, -1.0, 1.0, plant="double_integrator", dt=0.02)
    for B in batches:
        params = torch.randn((B, eng.num_params), generator=g).cuda()
        ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
        iv = torch.zeros((B, D), device="cuda")
        pos = torch.empty((B, T, D), device="cuda"); vel = torch.empty_like(pos); act = torch.empty_like(pos)
        cp, cv = ip.double().contiguous(), iv.double().contiguous()
        q, qd = cp.clone(), cv.clone()
        launches = (
            ("trajectory", 224 + 2 * T * D * 4, lambda: eng.trajectory(params, ip, iv, 0.0, out=(pos, vel))),
            ("trajectory + actions", 224 + 3 * T * D * 4, lambda: eng.trajectory_actions(params, ip, iv, spec_s, cp, cv, out=(pos, vel, act))),
            ("closed-loop step", 224 + 3 * T * D * 4, lambda: eng.trajectory_rollout(params, ip, iv, spec_d, q, qd, out=(pos, vel, act))),
        )
        for name, nbytes, fn in launches:
            for rnd in range(2):
                for route, opt in (("response", -1), ("serial Euler", 0)):
                    _lib.reset_options()
                    _lib.set_option("dmp_response", opt)
                    fn(); torch.cuda.synchronize()
                    kern = eng.last_kernel()
                    t = graph_time(fn)
                    print(f"| {name} | {B} | {route} | `{kern}` | {t * 1e6:.1f} | {B / t:.3e} | {B * nbytes / t / 1e9:.0f} | "
                          f"{B * nbytes / t / 8e12 * 100:.1f} % |", flush=True)
    _lib.reset_options()


if __name__ == "__main__":
    main()
