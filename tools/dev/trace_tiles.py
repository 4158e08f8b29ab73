#!/usr/bin/env python3
"""Per-phase timeline of ONE wave of the headline launch (k_traj_tiles<prodmp,act>, cfg2, B = 4096) from a -DMPK_TRACE build:
MPK_LIB=fancy_gym_amd/libmpk_trace.so python tools/dev/trace_tiles.py [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import CFG2, DG, PG  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = C.CDLL(_lib.LIB_PATH)
    lib.mpk_debug_trace.argtypes = [C.c_void_p, C.c_int]
    eng = TrajectoryEngine(device=0, **CFG2)
    g = torch.Generator().manual_seed(0)
    params = torch.randn((B, 42), generator=g).to(dev)
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
    iv = torch.zeros((B, 7), device=dev)
    spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    fn = lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, 0.0, out=out)   # noqa: E731
    buf = np.zeros(512, np.int64)
    for _ in range(500):
        fn()
    torch.cuda.synchronize()
    lib.mpk_debug_trace(buf.ctypes.data, 256)
    tot = []
    for rep in range(8):
        for _ in range(20):           # keep the clocks up, trace the last launch
            fn()
        torch.cuda.synchronize()
        lib.mpk_debug_trace(buf.ctypes.data, 256)
        fn()
        torch.cuda.synchronize()
        n = lib.mpk_debug_trace(buf.ctypes.data, 256)
        st = buf[:2 * n].reshape(n, 2)
        tot.append(st)
    st = tot[-1]
    print(f"{eng.last_kernel()} B={B}: {len(st)} stamps")
    prev = st[0, 1]
    for tag, c in st:
        print(f"  tag {tag:3d}  +{c - prev:7d}  (t = {c - st[0, 1]:7d})")
        prev = c
    print("totals over 8 traced launches:", [int(s[-1, 1] - s[0, 1]) for s in tot])


if __name__ == "__main__":
    main()
