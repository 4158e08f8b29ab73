#!/usr/bin/env python3
"""
Per-phase timeline of ONE wave of a kernel (development build with -DMPK_TRACE: wave 0 of one workgroup stamps the shader
clock at labelled points).   MPK_LIB=fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py [B] [full|plan] [key=value ...]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import CFG2, CFG4, DG, PG  # noqa: E402


def main():
    args = sys.argv[1:]
    B = int(args[0]) if args and args[0].isdigit() else 4096
    mode = "plan" if "plan" in args else "full"
    for kv in args:
        if "=" in kv:
            k, v = kv.split("=")
            _lib.set_option(k, int(v))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = C.CDLL(_lib.LIB_PATH)
    lib.mpk_debug_trace.argtypes = [C.c_void_p, C.c_int]
    kw, P = (CFG2, 42) if mode == "full" else (CFG4, 35)
    eng = TrajectoryEngine(device=0, **kw)
    g = torch.Generator().manual_seed(0)
    params = torch.randn((B, P), generator=g).to(dev)
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
    iv = torch.zeros((B, 7), device=dev)
    spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
    q, qd = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    ts = torch.zeros(B, dtype=torch.int32, device=dev); ps = torch.zeros_like(ts)
    dn = torch.zeros(B, dtype=torch.uint8, device=dev)

    def fn():
        if mode == "full":
            eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out)
        else:
            ts.zero_(); ps.zero_(); dn.zero_()
            eng.replan_step(params, ip, iv, spec, q, qd, ts, ps, dn, 25, 4, 100, condition=True, out=out)
    buf = np.zeros(512, np.int64)
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    lib.mpk_debug_trace(buf.ctypes.data, 256)
    for rep in range(3):
        fn()
        torch.cuda.synchronize()
        n = lib.mpk_debug_trace(buf.ctypes.data, 256)
        st = buf[:2 * n].reshape(n, 2)
        print(f"--- {eng.last_kernel()} B={B} {mode}: {n} stamps, total {st[-1, 1] - st[0, 1]} cycles")
        if rep == 2:
            st = st[np.argsort(st[:, 1], kind="stable")]
            prev = {0: st[0, 1], 1: st[0, 1]}
            for tag, c in st:
                w = int(tag >= 100)                      # tags >= 100: a second wave (MPK_STAMP_AT)
                print(f"  {'                         ' * w}tag {tag:3d}  +{c - prev[w]:7d}  (t = {c - st[0, 1]:7d})")
                prev[w] = c


if __name__ == "__main__":
    main()
