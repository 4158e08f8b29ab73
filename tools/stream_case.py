#!/usr/bin/env python3
"""One line of timings of the HBM-streaming open-loop step for the library MPK_LIB names (or the shipped one):
B = 262144 and 65536, trajectory + actions and trajectory only, auto kernel selection.  Used by tools/ab_libs.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402


def timed(fn, n=20, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
cols = []
for B in (262144, 65536):
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    if B == 262144:
        timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out), n=60)
    t3 = timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)); k3 = eng.last_kernel()
    t2 = timed(lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]))
    cols += [f"{t3:.0f}", f"{B * 8624 / t3 / 8e6 * 100:.1f} %", f"{t2:.0f}", f"{B * 5824 / t2 / 8e6 * 100:.1f} %"]
    del params, ip, iv, cp, cv, out
print("| " + os.path.basename(os.environ.get("MPK_LIB", "libmpk.so")) + " " + " ".join(sys.argv[1:]) + f" | `{k3}` | " + " | ".join(cols) + " |")
