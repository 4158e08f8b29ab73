#!/usr/bin/env python3
"""Does the placement of the three output arrays (pos, vel, actions) matter for the fused cfg2 step at B = 4096?
Same launch as bench.py, outputs placed (a) in three separate torch allocations, (b) back to back in one allocation,
(c) in one allocation with a skew between the arrays.   python tools/alloc_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    B, T, D, P = 4096, bench.T_STEPS, bench.D, bench.P
    eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
    g = torch.Generator(device="cpu").manual_seed(0)
    params = torch.randn((B, P), generator=g).to(dev)
    ip = (torch.rand((B, D), generator=g) * 2 - 1).to(dev)
    iv = torch.zeros((B, D), device=dev)
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    spec = RolloutSpec("motor", D, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
    lib = _lib.load()
    n = B * T * D

    def run(name, outs):
        ptrs = [o.data_ptr() for o in outs]

        def step(sp):
            rc = lib.mpk_trajectory_actions(eng._h, params.data_ptr(), ip.data_ptr(), iv.data_ptr(), 0.0, C.byref(spec.c),
                                            cp.data_ptr(), cv.data_ptr(), ptrs[0], ptrs[1], ptrs[2], B, sp)
            assert rc == 0, _lib.last_error()
        K = 1000
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(side.cuda_stream)
            with torch.cuda.graph(graph, stream=side):
                for _ in range(K):
                    step(side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / K)
        offs = [(p - ptrs[0]) for p in ptrs]
        print(f"| {name} | offsets {offs} | {best:.2f} us | {bench.BYTES_PER_TRAJ * B / best / 1e3:.0f} GB/s |")

    run("three torch.empty", [torch.empty((B, T, D), device=dev) for _ in range(3)])
    one = torch.empty(3 * n + 3 * 65536, device=dev)
    run("one allocation, back to back", [one[k * n:(k + 1) * n] for k in range(3)])
    for skew in (64, 1024, 4096 + 64, 16384 + 1024, 65536 + 4096):      # floats
        run(f"one allocation, skew {skew * 4} B", [one[k * (n + skew):k * (n + skew) + n] for k in range(3)])


if __name__ == "__main__":
    main()
