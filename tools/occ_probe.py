#!/usr/bin/env python3
"""How much does the tile-major kernel need its 7 waves per SIMD?  The headline launch (cfg2, fused PD actions, B = 4096 and
8192) with dynamic-LDS padding that caps the workgroups per CU ("lds_pad" option; static 16 KB + pad out of 160 KB)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import CFG2, DG, PG, graph_time  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    print("| batch | lds_pad KB | workgroups / CU | waves / SIMD | us |")
    print("|---|---|---|---|---|")
    for B in (4096, 8192):
        eng = TrajectoryEngine(device=0, **CFG2)
        params = torch.randn((B, 42), generator=g).to(dev)
        ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
        iv = torch.zeros((B, 7), device=dev)
        spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
        cp, cv = ip.double().contiguous(), iv.double().contiguous()
        out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
        for pad in (0, 4, 6, 10, 16, 24, 37):
            _lib.set_option("lds_pad", pad)
            t = graph_time(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, 0.0, out=out))
            wg = min(8, 160 // (16 + pad))
            print(f"| {B} | {pad} | {wg} | {wg} | {t * 1e6:.2f} |")
        _lib.reset_options()


if __name__ == "__main__":
    main()
