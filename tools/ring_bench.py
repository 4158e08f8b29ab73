#!/usr/bin/env python3
"""k_traj_ring (wave-specialised store engine) against k_traj_flat on the HBM-streaming open-loop step, one process, rows
alternated: python tools/ring_bench.py [B ...] [--variants np:m:ns,...] [--quick].  First checks the ring kernel's outputs against
k_traj_flat's bit for bit (every variant)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402


def timed(fn, n=20, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


args = [a for a in sys.argv[1:]]
variants = [(8, 4, 1), (4, 4, 1), (8, 2, 1), (8, 4, 2), (12, 4, 1)]
for a in args:
    if a.startswith("--variants="):
        variants = [tuple(int(x) for x in v.split(":")) for v in a.split("=", 1)[1].split(",")]
batches = [int(a) for a in args if a.isdigit()] or [262144, 65536]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")


def setup(v):
    _lib.reset_options()
    if v is None:
        _lib.set_option("flat", 1)
    else:
        _lib.set_option("ring", 2 if v[2] == 0 else 1); _lib.set_option("ring_np", v[0]); _lib.set_option("ring_m", v[1]); _lib.set_option("ring_ns", max(v[2], 1))
        if len(v) > 3:
            _lib.set_option("ring_dbg", v[3])


# ---- parity: every variant against k_traj_flat, bit for bit, incl. a ragged batch -----------------------------------------------
for B in (() if "--no-parity" in args else (5, 4099, 40000)):
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 3))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    setup(None)
    ref = [t.clone() for t in eng.trajectory_actions(params, ip, iv, spec, cp, cv)]
    ref2 = [t.clone() for t in eng.trajectory(params, ip, iv, 0.0)]
    for v in [x for x in variants if len(x) < 4 or x[3] in (0, 4)]:
        setup(v)
        out = tuple(torch.full((B, 100, 7), float("nan"), device=dev) for _ in range(3))
        eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)
        k = eng.last_kernel()
        torch.cuda.synchronize()
        ok3 = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(out, ref))
        out2 = tuple(torch.full((B, 100, 7), float("nan"), device=dev) for _ in range(2))
        eng.trajectory(params, ip, iv, 0.0, out=out2)
        torch.cuda.synchronize()
        ok2 = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(out2, ref2))
        print(f"parity B={B} np:m:ns={v} kernel={k} +actions {'ok' if ok3 else 'MISMATCH'} traj-only {'ok' if ok2 else 'MISMATCH'}", flush=True)
        assert k.startswith("k_traj_ring") or k.startswith("k_traj_burst"), k
        assert ok3 and ok2
_lib.reset_options()
print()
print("| B | kernel (np:m:ns) | +actions us | of 8 TB/s | trajectory only us | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for B in batches:
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out), n=40)       # clocks
    for rep in range(2):
        for v in [None] + variants:
            setup(v)
            t3 = timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)); k = eng.last_kernel()
            t2 = timed(lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]))
            print(f"| {B} | `{k}` {'' if v is None else v} | {t3 * 1e6:.1f} | {B * 8624 / t3 / 8e12 * 100:.1f} % | {t2 * 1e6:.1f} | "
                  f"{B * 5824 / t2 / 8e12 * 100:.1f} % |", flush=True)
    for o in out:
        o.fill_(1.0)
    tf = timed(lambda: [o.fill_(1.0) for o in out])
    print(f"| {B} | torch fill_ of the three arrays | {tf * 1e6:.1f} | {B * 8400 / tf / 8e12 * 100:.1f} % | | |", flush=True)
    _lib.reset_options()
    del out, params, ip, iv, cp, cv
