#!/usr/bin/env python3
"""
The reward rollout with its control-cost pass on helper waves ("pd_helper" 1, flag hand-over) against the pass on the chain waves: same
bits, and the time of both (LongSimpleReacher 5 x 200; captured graphs of 20).   python tools/helper_check.py [B ...] [key=value ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from tools.closed_bench import graph_time  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 8192, 16384, 65536, 200000]
    extra = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(0)
    D, T = 5, 200
    eng = TrajectoryEngine(device=0, mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=5, num_basis=5,
                           num_basis_zero_start=1, dt=0.01, duration=2.0, tau=2.0)
    spec = RolloutSpec("motor", D, 0.6, 0.075, -1000.0, 1000.0, plant="double_integrator", dt=0.01)
    print(f"lib: {_lib.LIB_PATH}")
    print("| B | pd_quad | no reward us | reward, pass on the chain waves us | on helper waves us | same bits |")
    print("|---|---|---|---|---|---|")
    for B in batches:
        params = torch.randn((B, eng.num_params), generator=g).cuda()
        ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
        iv = torch.zeros((B, D), device="cuda")
        pos, vel = eng.trajectory(params, ip, iv, 0.0)
        goal = (torch.rand((B, 2), generator=g, dtype=torch.float64) * 4 - 2).cuda()
        ns = torch.randint(0, T + 1, (B,), generator=g, dtype=torch.int32).cuda()
        for quad in (-1, 0, 3, 2):
            out = {}
            for hw in (0, 1):
                _lib.reset_options()
                for k, v in extra.items():
                    _lib.set_option(k, v)
                if quad >= 0:
                    _lib.set_option("pd_quad", quad)
                _lib.set_option("pd_helper", hw)
                act = torch.zeros((B, T, D), device="cuda"); rew = torch.zeros((B, T), dtype=torch.float64, device="cuda")
                q, qd = ip.double().contiguous(), iv.double().contiguous()
                eng.reacher_rollout(spec, pos, vel, q, qd, goal, n_steps=ns, out=(act, rew))
                torch.cuda.synchronize()
                eng.check_range()
                q2, qd2 = ip.double().contiguous(), iv.double().contiguous()
                a2 = torch.empty_like(act); r2 = torch.empty_like(rew)
                us = graph_time(lambda: eng.reacher_rollout(spec, pos, vel, q2, qd2, goal, out=(a2, r2)), 20)
                out[hw] = (act, rew, q, qd, us)
            same = all(torch.equal(out[0][i], out[1][i]) for i in range(4))
            _lib.reset_options()
            if quad >= 0:
                _lib.set_option("pd_quad", quad)
            a3 = torch.empty((B, T, D), device="cuda")
            q3, qd3 = ip.double().contiguous(), iv.double().contiguous()
            plain = graph_time(lambda: eng.pd_rollout(spec, pos, vel, q3, qd3, out=a3), 20)
            print(f"| {B} | {quad} | {plain:.1f} | {out[0][4]:.1f} (+{100 * (out[0][4] / plain - 1):.0f} %) | {out[1][4]:.1f} (+{100 * (out[1][4] / plain - 1):.0f} %) | {same} |", flush=True)
    _lib.reset_options()


if __name__ == "__main__":
    main()
