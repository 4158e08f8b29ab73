#!/usr/bin/env python3
"""k_traj_flat against the automatic choice below its 96 MB threshold (cfg2 fused trajectory + actions), graph-timed."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from tools.closed_bench import graph_time  # noqa: E402
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
print("| B | options | kernel | +actions us | of 8 TB/s | trajectory only us | of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for B in (2048, 4096, 8192, 16384, 32768):
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    for opts in ({}, {"flat": 1}, {"flat": 1, "write_through": 1}, {"flat": 1, "write_through": 0}, {"mapping": 1}):
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        t3 = graph_time(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)); k3 = eng.last_kernel()
        t2 = graph_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]))
        print(f"| {B} | {opts or 'auto'} | `{k3}` | {t3 * 1e6:.2f} | {B * 8624 / t3 / 8e12 * 100:.1f} % | {t2 * 1e6:.2f} | {B * 5824 / t2 / 8e12 * 100:.1f} % |")
    _lib.reset_options()
