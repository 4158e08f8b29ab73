#!/usr/bin/env python3
"""
Per-phase timeline of wave 0 of workgroup 0 of the per-episode-phase kernel (development build with -DMPK_TRACE), last
chunk the wave processed.   MPK_LIB=fancy_gym_amd/libmpk_trace.so python tools/dev/trace_phase.py [B] [key=value ...]
tags: 1 chunk start | per episode e (+40 e): 2 start, 3 columns built, 10/20 row gathered (round 0/1),
12/22 contracted + staged, 13/23 flushed
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import TrajectoryEngine, _lib  # noqa: E402
from run_cfg import KW  # noqa: E402


def main():
    args = sys.argv[1:]
    B = int(args[0]) if args and args[0].isdigit() else 65536
    for kv in args:
        if "=" in kv:
            k, v = kv.split("=")
            _lib.set_option(k, int(v))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = C.CDLL(_lib.LIB_PATH)
    lib.mpk_debug_trace.argtypes = [C.c_void_p, C.c_int]
    name = next((x for x in args if x in KW), "cfg2tau")      # cfg2tau (prodmp) | cfg5tau (promp)
    kw = KW[name]
    eng = TrajectoryEngine(device=0, **kw)
    g = torch.Generator().manual_seed(0)
    lo, hi = kw["tau_bound"]
    params = torch.randn((B, eng.num_params), generator=g)
    params[:, 0] = torch.rand(B, generator=g) * (hi - lo) * 0.6 + lo + 0.3 * (hi - lo)
    if kw.get("learn_delay"):
        params[:, 1] = torch.rand(B, generator=g) * 0.1 + 0.05
    params = params.to(dev)
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
    iv = torch.zeros((B, 7), device=dev)
    out = (torch.empty((B, eng.num_steps, 7), device=dev), torch.empty((B, eng.num_steps, 7), device=dev))
    buf = np.zeros(512, np.int64)
    for _ in range(20):
        eng.trajectory(params, ip, iv, 0.0, out=out)
    torch.cuda.synchronize()
    lib.mpk_debug_trace(buf.ctypes.data, 256)
    for rep in range(2):
        eng.trajectory(params, ip, iv, 0.0, out=out)
        torch.cuda.synchronize()
        n = lib.mpk_debug_trace(buf.ctypes.data, 256)
        st = buf[:2 * n].reshape(n, 2)
        print(f"--- {eng.last_kernel()} B={B}: {n} stamps, span {st[-1, 1] - st[0, 1]} cycles")
        if rep == 1:
            prev = st[0, 1]
            for tag, c in st:
                print(f"  tag {tag:3d}  +{c - prev:7d}  (t = {c - st[0, 1]:7d})")
                prev = c


if __name__ == "__main__":
    main()
