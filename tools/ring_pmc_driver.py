#!/usr/bin/env python3
"""For a rocprofv3 --pmc pass: the streaming row (cfg2 + actions, B episodes) through k_traj_flat, k_traj_ring, the ring's
production-only and store-only ablations, and a torch fill of the three arrays, n launches each.
   rocprofv3 --pmc ... -- python3 tools/ring_pmc_driver.py [B] [n]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
cp, cv = ip.double().contiguous(), iv.double().contiguous()
out = tuple(torch.empty((B, eng.num_steps, eng.num_dof), device=dev) for _ in range(3))
cases = [("flat", {"flat": 1}), ("ring", {"ring": 1}), ("ring production only", {"ring": 1, "ring_dbg": 2}),
         ("ring engine only", {"ring": 1, "ring_dbg": 9})]
_lib.set_option("flat", 1)                      # warm-up launches must not disturb the ring's dispatch cycle
for _ in range(3):
    eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)
torch.cuda.synchronize()
for _ in range(n):
    for name, opts in cases:
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)
        torch.cuda.synchronize()
    for o in out:
        o.fill_(1.0)
torch.cuda.synchronize()
print("order per round:", [c[0] for c in cases] + ["fill x3"])
