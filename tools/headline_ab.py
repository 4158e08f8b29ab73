#!/usr/bin/env python3
"""The headline launch (cfg2 fused trajectory + PD actions, B = 4096, outputs cache resident) under other work decompositions, one
process, rows alternated, each timed as a captured graph of 20 launches after ~40 ms of load:  python tools/headline_ab.py [B ...]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import graph_time  # noqa: E402

VARIANTS = [("automatic (k_traj_tiles, 7168 waves)", {}),
            ("tile-major, one item per wave", {"mapping": 1, "ipw": 1}),
            ("tile-major, four items per wave", {"mapping": 1, "ipw": 4}),
            ("tile-major, 128-thread workgroups", {"tiles_wpb": 2}),
            ("tile-major, 64-thread workgroups", {"tiles_wpb": 1}),
            ("tile-major, 128-thread workgroups, four items per wave", {"tiles_wpb": 2, "ipw": 4, "mapping": 1}),
            ("short-lived workgroup per group, 7 waves (one row tile each), whole-trajectory image, contiguous runs", {"ring": 2, "ring_m": 1, "ring_np": 7}),
            ("... A fragments from the cached table, no LDS copy", {"ring": 2, "ring_m": 1, "ring_np": 7, "ring_dbg": 16}),
            ("... 4 waves per group", {"ring": 2, "ring_m": 1, "ring_np": 4, "ring_dbg": 16}),
            ("... two groups per workgroup, 4 waves each", {"ring": 2, "ring_m": 2, "ring_np": 4, "ring_dbg": 16}),
            ("... four groups per workgroup, 2 waves each", {"ring": 2, "ring_m": 4, "ring_np": 2, "ring_dbg": 16}),
            ("k_traj_flat (persistent, whole-trajectory images per wave)", {"flat": 1}),
            ("k_traj_ring (persistent producers + store engine)", {"ring": 1}),
            ("episode-major k_traj_stream", {"mapping": 2, "flat": 0, "ring": 0})]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096]
print("| B | variant | kernel | us | of 8 TB/s |")
print("|---|---|---|---|---|")
for B in batches:
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 0))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    ref = None
    for rep in range(2):
        for name, opts in VARIANTS:
            _lib.reset_options()
            for k, v in opts.items():
                _lib.set_option(k, v)
            t = graph_time(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out))
            torch.cuda.synchronize()
            if ref is None:
                ref = [o.clone() for o in out]
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            print(f"| {B} | {name} | `{eng.last_kernel()}` | {t * 1e6:.2f} | {B * 8624 / t / 8e12 * 100:.1f} %{'' if same else '  OUTPUTS DIFFER'} |", flush=True)
    _lib.reset_options()
