#!/usr/bin/env python3
"""For a rocprofv3 --pmc pass: the fused closed-loop step at B episodes through k_traj_ring<.., closed> (full horizon and a
25-of-100-step replanning step) and through k_traj_duo, n launches each.
   rocprofv3 --pmc ... -- python3 tools/ringc_pmc_driver.py [B] [n] [key=value ...]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from closed_bench import CFG2, CFG4, PG, DG  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 65536
n = nums[1] if len(nums) > 1 else 3
ring_opts = {"ring": 1}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        ring_opts[k] = int(v)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
for name, kw, P in (("full", CFG2, 42), ("plan", CFG4, 35)):
    eng = TrajectoryEngine(device=0, **kw)
    params = torch.randn((B, P), generator=g).to(dev)
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
    iv = torch.zeros((B, 7), device=dev)
    q, qd = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    ts = torch.zeros(B, dtype=torch.int32, device=dev)
    ps = torch.zeros(B, dtype=torch.int32, device=dev)
    dn = torch.zeros(B, dtype=torch.uint8, device=dev)
    for opts in (ring_opts, {"quad": 3}):
        _lib.reset_options()
        for k, v in opts.items():
            _lib.set_option(k, v)
        for _ in range(n):
            if name == "full":
                eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out)
            else:
                eng.replan_step(params, ip, iv, spec, q, qd, ts, ps, dn, 25, 2 ** 30, 2 ** 30, condition=True, out=out)
            torch.cuda.synchronize()
        print(name, opts, eng.last_kernel())
_lib.reset_options()
