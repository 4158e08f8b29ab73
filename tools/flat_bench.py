#!/usr/bin/env python3
"""k_traj_flat with the DoF count compiled in (k_traj_flat_d) against the generic kernel, the interleaved flush, and k_traj_ring, one
process, rows alternated; first a bitwise parity check of every row against the generic kernel.
    python tools/flat_bench.py [B ...]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402

ROWS = [("flat, generic", {"flat": 1, "ring": 0, "ring_dbg": 64}), ("flat, compile-time DoF", {"flat": 1, "ring": 0}),
        ("flat, compile-time DoF, interleaved flush", {"flat": 1, "ring": 0, "ring_dbg": 32}), ("ring", {"ring": 1}), ("auto", {})]


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


def setup(opts):
    _lib.reset_options()
    for k, v in opts.items():
        _lib.set_option(k, v)


batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384, 32768, 65536, 131072]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
for B in (5, 4099, 40000):
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 3))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    setup(ROWS[0][1])
    ref = [t.clone() for t in eng.trajectory_actions(params, ip, iv, spec, cp, cv)]
    ref2 = [t.clone() for t in eng.trajectory(params, ip, iv, 0.0)]
    for name, opts in ROWS[1:]:
        setup(opts)
        out = tuple(torch.full((B, 100, 7), float("nan"), device=dev) for _ in range(3))
        eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)
        k = eng.last_kernel()
        out2 = tuple(torch.full((B, 100, 7), float("nan"), device=dev) for _ in range(2))
        eng.trajectory(params, ip, iv, 0.0, out=out2)
        torch.cuda.synchronize()
        ok = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(out + out2, ref + ref2))
        print(f"parity B={B} {name} [{k}] {'ok' if ok else 'MISMATCH'}", flush=True)
        assert ok
print()
print("| B | row | kernel | +actions us | of 8 TB/s | trajectory only us | of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for B in batches:
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    fa = lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)      # noqa: E731
    ft = lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2])                    # noqa: E731
    n = max(5, int(2e-3 / (B * 2e-9)))
    busy = 0.0
    while busy < 0.06:
        busy += timed(fa, n) * n
    ta = {r[0]: [] for r in ROWS}; tt = {r[0]: [] for r in ROWS}; kn = {}
    for rep in range(5):
        for name, opts in (ROWS if rep % 2 == 0 else ROWS[::-1]):
            setup(opts)
            fa(); ta[name].append(timed(fa, n)); kn[name] = eng.last_kernel()
            ft(); tt[name].append(timed(ft, n))
    for name, _ in ROWS:
        a, t = float(np.median(ta[name])), float(np.median(tt[name]))
        print(f"| {B} | {name} | `{kn[name]}` | {a * 1e6:.1f} | {B * 8624 / a / 8e12 * 100:.1f} % | {t * 1e6:.1f} | {B * 5824 / t / 8e12 * 100:.1f} % |", flush=True)
_lib.reset_options()
