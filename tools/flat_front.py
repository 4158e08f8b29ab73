#!/usr/bin/env python3
"""k_traj_flat with one instead of two resident workgroups per CU (mpk_set_option "lds_pad": the LDS-resident output that all
waves flush at about the same time -- the write front -- halves): python tools/flat_front.py [B ...]"""
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
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [65536, 262144, 1048576]
print("| B | lds_pad KB (workgroups per CU: +actions | trajectory only) | +actions us | of 8 TB/s | trajectory only us | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for B in batches:
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    for pad, wg in ((0, "2 | 3"), (20, "1 | 2"), (45, "1 | 1"), (0, "2 | 3"), (20, "1 | 2"), (45, "1 | 1")):
        _lib.reset_options()
        _lib.set_option("flat", 1)
        if pad:
            _lib.set_option("lds_pad", pad)
        t3 = graph_time(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out), reps=6)
        t2 = graph_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]), reps=6)
        print(f"| {B} | {pad} ({wg}) | {t3 * 1e6:.1f} | {B * 8624 / t3 / 8e12 * 100:.1f} % | {t2 * 1e6:.1f} | {B * 5824 / t2 / 8e12 * 100:.1f} % |", flush=True)
    _lib.reset_options()
    del out, params
