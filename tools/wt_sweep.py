#!/usr/bin/env python3
"""write-through (sc1) against plain stores per kernel family and batch size (cfg2 fused trajectory + actions / trajectory
only), graph-timed, one process:  python tools/wt_sweep.py [B ...]"""
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
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384, 32768, 65536, 131072, 262144]
print("| B | options | kernel | +actions us | of 8 TB/s | kernel | trajectory only us | of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|")
for B in batches:
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    for opts in ({}, {"flat": 1, "write_through": 0}, {"flat": 1, "write_through": 1}, {"flat": 0, "write_through": 0},
                 {"flat": 0, "write_through": 1}, {}):
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        reps = 20 if B <= 32768 else 6
        t3 = graph_time(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out), reps=reps); k3 = eng.last_kernel()
        t2 = graph_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]), reps=reps); k2 = eng.last_kernel()
        print(f"| {B} | {opts or 'auto'} | `{k3}` | {t3 * 1e6:.1f} | {B * 8624 / t3 / 8e12 * 100:.1f} % | `{k2}` | {t2 * 1e6:.1f} | {B * 5824 / t2 / 8e12 * 100:.1f} % |")
    _lib.reset_options()
    del out, params
