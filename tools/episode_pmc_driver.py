#!/usr/bin/env python3
"""
The float64-chain kernels launched n times at batch B for a rocprofv3 --pmc pass:
    which = lean      mpk_episode_return, cfg2 full horizon, no reward           (k_episode_return<prodmp>)
            lean_rw   mpk_episode_return, LongSimpleReacher 5 x 200 + reward      (k_episode_return<promp,reacher>)
            roll      mpk_pd_rollout on existing trajectories, LongSimpleReacher  (k_pd_rollout_tiles<..>)
            roll_rw   mpk_reacher_rollout (the same + reward)                    (k_pd_rollout_tiles<.., reward>)
            tt_lean   mpk_episode_return, TableTennis-ProDMP learned phase       (k_phase_fused<prodmp,..,closed,lean>)
            tt_step   mpk_replan_step, TableTennis-ProDMP learned phase          (k_phase_fused<prodmp,..,closed>)
  ... -- python3 tools/episode_pmc_driver.py <which> [B] [n] [key=value ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import CFG2, DG, PG  # noqa: E402

args = [a for a in sys.argv[1:] if "=" not in a]
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
which = args[0]
B = int(args[1]) if len(args) > 1 else 4096
n = int(args[2]) if len(args) > 2 else 30
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(0)
if which.startswith("tt_"):
    from learned_phase_bench import CASES, make_params
    case = CASES["TT-ProDMP"]
    eng = TrajectoryEngine(device=0, **case["kw"])
    D, T = eng.num_dof, eng.num_steps
    params = make_params(case, B, eng.num_params, g).cuda()
    spec = RolloutSpec("motor", D, case["gains"][0], case["gains"][1], -1.0, 1.0, plant="double_integrator", dt=case["kw"]["dt"])
    ip = (0.2 * (torch.rand((B, D), generator=g) * 2 - 1)).cuda()
elif which in ("lean",):
    eng = TrajectoryEngine(device=0, **CFG2)
    D, T = 7, 100
    params = torch.randn((B, eng.num_params), generator=g).cuda()
    spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
    ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
else:
    eng = TrajectoryEngine(device=0, mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=5, num_basis=5,
                           num_basis_zero_start=1, dt=0.01, duration=2.0, tau=2.0)
    D, T = 5, 200
    params = torch.randn((B, eng.num_params), generator=g).cuda()
    spec = RolloutSpec("motor", 5, 0.6, 0.075, -1000.0, 1000.0, plant="double_integrator", dt=0.01)
    ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
iv = torch.zeros((B, D), device="cuda")
q, qd = ip.double().contiguous(), iv.double().contiguous()
goal = (torch.rand((B, 2), generator=g, dtype=torch.float64) * 4 - 2).cuda()
out = tuple(torch.empty((B, T, D), device="cuda") for _ in range(3))
rew = torch.empty((B, T), dtype=torch.float64, device="cuda")
i32 = dict(dtype=torch.int32, device="cuda")
ts, ps, dn = torch.zeros(B, **i32), torch.zeros(B, **i32), torch.zeros(B, dtype=torch.uint8, device="cuda")
pos, vel = eng.trajectory(params, ip, iv, 0.0)
for _ in range(n):
    if which == "lean":
        eng.episode_return(params, ip, iv, spec, q, qd)
    elif which == "lean_rw":
        eng.episode_return(params, ip, iv, spec, q, qd, reward="simple_reacher", goal=goal)
    elif which == "roll":
        eng.pd_rollout(spec, pos, vel, q, qd, out=out[2])
    elif which == "roll_rw":
        eng.reacher_rollout(spec, pos, vel, q, qd, goal, out=(out[2], rew))
    elif which == "tt_lean":
        eng.episode_return(params, ip, iv, spec, q, qd, replan=(ts, ps, dn, T, 2 ** 30, 2 ** 30), condition=True)
    elif which == "tt_step":
        eng.replan_step(params, ip, iv, spec, q, qd, ts, ps, dn, T, 2 ** 30, 2 ** 30, condition=True, out=out)
    else:
        raise SystemExit(f"unknown {which}")
torch.cuda.synchronize()
print("done", which, B, n, eng.last_kernel())
