#!/usr/bin/env python3
"""
The HBM-streaming launch of bench.py's `roofline_streaming` row (cfg2 fused trajectory + PD actions, B = 262144: 2.26 GB
per launch) under every kernel variant and two output placements, in ONE process on ONE box: the row differs by 35 %
between gpurun calls with identical clocks (profiles/r02_bench_kernel_stats.md); this tells which choices the slow
state is sensitive to.     python tools/stream_probe.py [B]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402


def timed(fn, n=12):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
    spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    T, D = eng.num_steps, eng.num_dof
    nbytes = B * (eng.num_params * 4 + 2 * D * 4 + 3 * T * D * 4)
    sep = tuple(torch.empty((B, T, D), device=dev) for _ in range(3))
    slab = torch.empty((3, B, T, D), device=dev)
    one = (slab[0], slab[1], slab[2])
    # warm the clocks
    t0 = timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=sep), n=60)
    fill = timed(lambda: [o.fill_(1.0) for o in sep])
    print(f"B = {B}: fill of the three arrays {3 * B * T * D * 4 / fill / 1e9:.0f} GB/s")
    print("| outputs | options | kernel | us | GB/s (alg.) | of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    variants = [{}, {"bulk": 0}, {"write_through": 1}, {"mapping": 1}, {"mapping": 1, "write_through": 0},
                {"mapping": 1, "ipw": 1}, {"mapping": 1, "ipw": 7}]
    for name, out in (("three allocations", sep), ("one slab", one)):
        for opts in variants:
            _lib.reset_options()
            for k, v in opts.items():
                _lib.set_option(k, v)
            t = timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out))
            print(f"| {name} | {opts or 'auto'} | `{eng.last_kernel()}` | {t * 1e6:.1f} | {nbytes / t / 1e9:.0f} | "
                  f"{nbytes / t / 8e12 * 100:.1f} % |")
    _lib.reset_options()
    # trajectory only (two output streams), and the pure rollout kernel on existing trajectories
    t = timed(lambda: eng.trajectory(params, ip, iv, 0.0, out=sep[:2]))
    nb2 = B * (eng.num_params * 4 + 2 * D * 4 + 2 * T * D * 4)
    print(f"| three allocations | trajectory only | `{eng.last_kernel()}` | {t * 1e6:.1f} | {nb2 / t / 1e9:.0f} | {nb2 / t / 8e12 * 100:.1f} % |")


if __name__ == "__main__":
    main()
