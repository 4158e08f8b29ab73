#!/usr/bin/env python3
"""
The serial-recurrence trajectory kernels at HBM-streaming sizes: the fused closed-loop step (cfg2 shape, all 100 steps) and cfg3
(DMP 7 x 5 x 200, Euler) through k_traj_duo / k_traj_quad / k_traj_mono (and whatever the launcher picks), HIP events over n
launches after 60 ms of GPU-busy warm-up per row, rounds alternating over the rows.
    [MPK_LIB=<other build>] python tools/serial_bench.py [B ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402

VARIANTS = [("auto", {}), ("duo", {"quad": 3}), ("quad", {"quad": 2}), ("mono", {"quad": 4})]


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


def rows(title, fn, kernel, nbytes, B):
    for _, opts in VARIANTS:                       # warm every row
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        busy = 0.0
        while busy < 0.06:
            busy += timed(fn, 4) * 4
    ts = {v[0]: [] for v in VARIANTS}
    names = {}
    for r in range(7):
        order = VARIANTS if r % 2 == 0 else VARIANTS[::-1]
        for vn, opts in order:
            _lib.reset_options()
            for k, v in opts.items():
                _lib.set_option(k, v)
            fn()
            ts[vn].append(timed(fn, 5))
            names[vn] = kernel()
    _lib.reset_options()
    for vn, _ in VARIANTS:
        t = float(np.median(ts[vn]))
        print(f"| {title} | {B} | {vn} | `{names[vn]}` | {t * 1e6:.1f} | {B * nbytes / t / 1e9:.0f} | {B * nbytes / t / 8e12 * 100:.1f} % |")


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384, 65536, 262144]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    print(f"lib: {_lib.LIB_PATH}")
    print("| launch | batch | variant | kernel | us | GB/s (alg.) | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    g = torch.Generator().manual_seed(0)
    for B in batches:
        if B <= 65536:
            eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
            spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="double_integrator", dt=0.02)
            params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
            q, qd = ip.double().contiguous(), iv.double().contiguous()
            out = tuple(torch.empty((B, eng.num_steps, eng.num_dof), device=dev) for _ in range(3))
            rows("closed loop, cfg2", lambda: eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out), eng.last_kernel,
                 224 + 3 * 2800, B)
            del eng, out
        dmp = TrajectoryEngine(mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=4.0,
                               tau=4.0, alpha_phase=2.0, dmp_alpha=25.0, device=0)
        dparams = torch.randn((B, dmp.num_params), generator=g).to(dev)
        ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
        iv = torch.zeros((B, 7), device=dev)
        dout = tuple(torch.empty((B, dmp.num_steps, 7), device=dev) for _ in range(2))
        rows("cfg3 DMP", lambda: dmp.trajectory(dparams, ip, iv, out=dout), dmp.last_kernel, 224 + 2 * 5600, B)
        del dmp, dout


if __name__ == "__main__":
    main()
