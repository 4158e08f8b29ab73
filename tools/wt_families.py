#!/usr/bin/env python3
"""write-through against plain stores for the other kernel families at the sizes where their outputs are between the L2 and
the memory-side cache (cfg3 DMP, fused closed loop, replanning plan, per-episode phase, rollout):
    python tools/wt_families.py"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import graph_time  # noqa: E402
from run_cfg import KW  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def inputs(eng, B, kw=None):
    P, D = eng.num_params, eng.num_dof
    params = torch.randn((B, P), generator=g)
    if kw and kw.get("learn_tau"):
        lo, hi = kw["tau_bound"]
        params[:, 0] = torch.rand(B, generator=g) * (hi - lo) * 0.6 + lo + 0.3 * (hi - lo)
    if kw and kw.get("learn_delay"):
        params[:, 1] = torch.rand(B, generator=g) * 0.1 + 0.05
    return params.to(dev), (torch.rand((B, D), generator=g) * 2 - 1).to(dev), torch.zeros((B, D), device=dev)


def row(name, B, out_mb, fn, eng):
    res = []
    for wt in (0, 1, -1, 0, 1):
        _lib.reset_options()
        if wt >= 0:
            _lib.set_option("write_through", wt)
        t = graph_time(fn, reps=20 if out_mb < 300 else 6)
        res.append((wt, t, eng.last_kernel()))
    _lib.reset_options()
    cells = " | ".join(f"{t * 1e6:.1f}" for _, t, _ in res)
    print(f"| {name} | {B} | {out_mb:.0f} | `{res[2][2]}` | {cells} |", flush=True)


print("| config | B | output MB | kernel (auto) | wt=0 us | wt=1 us | auto us | wt=0 again | wt=1 again |")
print("|---|---|---|---|---|---|---|---|---|")
# cfg3 DMP
eng = TrajectoryEngine(device=0, **KW["cfg3"])
for B in (8192, 16384, 24576, 32768, 65536):
    p, ip, iv = inputs(eng, B)
    out = tuple(torch.empty((B, eng.num_steps, eng.num_dof), device=dev) for _ in range(2))
    row("cfg3 dmp", B, B * eng.num_steps * eng.num_dof * 8 / 2**20, lambda: eng.trajectory(p, ip, iv, 0.0, out=out), eng)
    del out
# cfg2 fused closed loop, full horizon
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="double_integrator", dt=0.02)
for B in (8192, 16384, 32768, 65536):
    p, ip, iv = inputs(eng, B)
    q0, qd0 = ip.double().contiguous(), iv.double().contiguous()
    q, qd = q0.clone(), qd0.clone()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))

    def step():
        q.copy_(q0); qd.copy_(qd0)
        eng.trajectory_rollout(p, ip, iv, spec, q, qd, out=out)
    try:
        row("cfg2 closed loop (+ 2 state copies)", B, B * 8400 / 2**20, step, eng)
    except Exception as e:  # noqa: BLE001
        print(f"| cfg2 closed loop | {B} | - | {type(e).__name__}: {e} |")
    del out
# per-episode phase kernels
for nm, key in (("cfg2' prodmp learn_tau", "cfg2tau"), ("cfg5' promp learn_tau+delay", "cfg5tau"),
                ("cfg3' dmp learn_tau", dict(KW["cfg3"], learn_tau=True, tau_bound=(2.0, 4.0)))):
    kw = KW[key] if isinstance(key, str) else key
    eng = TrajectoryEngine(device=0, **kw)
    for B in (8192, 16384, 32768, 65536):
        if eng.num_steps > 300 and B > 16384:
            continue
        p, ip, iv = inputs(eng, B, kw)
        out = tuple(torch.empty((B, eng.num_steps, eng.num_dof), device=dev) for _ in range(2))
        row(nm, B, B * eng.num_steps * eng.num_dof * 8 / 2**20, lambda: eng.trajectory(p, ip, iv, 0.0, out=out), eng)
        del out
# rollout kernel alone
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
for B in (8192, 16384, 32768, 65536):
    p, ip, iv = inputs(eng, B)
    pos, vel = eng.trajectory(p, ip, iv, 0.0)
    q0 = ip.double().contiguous()
    q, qd = q0.clone(), torch.zeros_like(q0)
    act = torch.empty((B, 100, 7), device=dev)

    def roll():
        q.copy_(q0); qd.zero_()
        eng.pd_rollout(spec, pos, vel, q, qd, out=act)
    row("cfg2 rollout alone (+ 2 state copies)", B, B * 2800 * 3 / 2**20, roll, eng)
