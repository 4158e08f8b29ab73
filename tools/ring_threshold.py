#!/usr/bin/env python3
"""Where k_traj_ring should take over from the kernel the launcher picks otherwise, per shape: cfg5 (350 x 7, T D = 2 mod 4: no
k_traj_flat), cfg1 (200 x 5: images beyond k_traj_flat's budget), cfg2 / cfg4 (k_traj_flat applies).  + actions and trajectory only.
    python tools/ring_threshold.py"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from run_cfg import KW as CFGS  # noqa: E402
CFGS = dict(CFGS)
CFGS["cfg1"] = dict(mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=5, num_basis=5, num_basis_zero_start=1,
                    dt=0.02, duration=4.0, tau=4.0)

PG = 0.01 * np.array([120., 120., 120., 120., 50., 30., 10.])
DG = 0.01 * np.array([10., 10., 10., 10., 6., 5., 3.])


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
print("| shape | B | MB out | launch | auto kernel | auto us | ring us | ring / auto |")
print("|---|---|---|---|---|---|---|---|")
g = torch.Generator().manual_seed(0)
for name, batches in (("cfg5", (4096, 8192, 16384, 32768)), ("cfg1", (8192, 16384, 32768, 65536, 131072)), ("cfg2", (32768, 49152, 65536, 98304))):
    kw = dict(CFGS[name])
    eng = TrajectoryEngine(device=0, **kw)
    D, T = eng.num_dof, eng.num_steps
    for B in batches:
        params = torch.randn((B, eng.num_params), generator=g).to(dev)
        ip = (torch.rand((B, D), generator=g) * 2 - 1).to(dev)
        iv = torch.zeros((B, D), device=dev)
        cp, cv = ip.double().contiguous(), iv.double().contiguous()
        spec = RolloutSpec("motor", D, PG[:D], DG[:D], -1.0, 1.0, plant="static")
        out = tuple(torch.empty((B, T, D), device=dev) for _ in range(3))
        for launch, fn, narr in (("+actions", lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out), 3),
                                 ("trajectory", lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]), 2)):
            mb = B * T * D * 4 * narr / 1e6
            n = max(4, int(3e-3 / (mb * 1e6 / 5e12)))
            res = {}
            busy = 0.0
            while busy < 0.05:
                busy += timed(fn, n) * n
            for rep in range(3):
                for key, opts in (("auto", {"ring": 0}), ("ring", {"ring": 1})):
                    _lib.reset_options()
                    for k, v in opts.items():
                        _lib.set_option(k, v)
                    fn()
                    res.setdefault(key, []).append(timed(fn, n))
                    res[key + "_k"] = eng.last_kernel()
            _lib.reset_options()
            a, r = float(np.median(res["auto"])), float(np.median(res["ring"]))
            print(f"| {name} | {B} | {mb:.0f} | {launch} | `{res['auto_k']}` | {a * 1e6:.1f} | {r * 1e6:.1f} (`{res['ring_k']}`) | {r / a:.3f} |", flush=True)
    del eng
