#!/usr/bin/env python3
"""
Batches whose outputs exceed 2^31 elements (the sizes 288 GB of HBM invite): every kernel family addresses them with
64-bit offsets.  First / last / random rows against the oracle; needs ~60 GB of device memory.
    python tools/big_batch_check.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from fancy_gym_amd import RolloutSpec  # noqa: E402
from oracle import mp_oracle as O  # noqa: E402  (checker only: this is a test tool)
from tests.test_gpu_trajectory import CFG2, CFG3, CFG5, close, fd_atol, inputs, make_engine  # noqa: E402


def main():
    for name, cfg, B in (("cfg2", CFG2, 5 * 2**20), ("cfg5", CFG5, 2**20), ("cfg3", CFG3, 2**21)):
        pc, bc, tc, dt, dur = cfg
        eng = make_engine(pc, bc, tc, dt, dur)
        T, D, P = eng.num_steps, eng.num_dof, eng.num_params
        print(name, "B", B, "elements", B * T * D, "> 2^31:", B * T * D > 2**31)
        g = torch.Generator().manual_seed(0)
        params = torch.randn((B, P), generator=g).cuda()
        ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda()
        iv = torch.zeros((B, D), device="cuda")
        pos, vel = eng.trajectory(params, ip, iv, 0.0)
        torch.cuda.synchronize()
        rows = np.concatenate([np.arange(8), np.random.default_rng(0).choice(B, 48, replace=False), np.arange(B - 8, B)])
        rp, rv = O.get_trajectory(pc, bc, tc, params[rows].cpu().numpy(), dur, dt, 0.0, ip[rows].cpu().numpy(), iv[rows].cpu().numpy(), dtype=np.float64)
        close(pos[rows].cpu().numpy(), rp, "pos")
        close(vel[rows].cpu().numpy(), rv, "vel", atol=fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0)
        print("  ok", eng.last_kernel())
        if name == "cfg2":
            spec = RolloutSpec("motor", D, 1.0, 0.1, -1.0, 1.0, plant="double_integrator", dt=dt)
            q, qd = ip.double().contiguous(), iv.double().contiguous()
            p2, v2, act = eng.trajectory_rollout(params, ip, iv, spec, q, qd)
            torch.cuda.synchronize()
            assert torch.equal(p2[rows], pos[rows])
            ra, rq, rqd = O.rollout(pos[rows].cpu().numpy(), vel[rows].cpu().numpy(), "motor", 1.0, 0.1, -1.0, 1.0, "double_integrator", dt, ip[rows].double().cpu().numpy(), iv[rows].double().cpu().numpy())
            assert np.array_equal(act[rows].cpu().numpy(), ra.astype(np.float32)), "actions"
            assert np.array_equal(q[rows].cpu().numpy(), rq)
            print("  closed loop ok", eng.last_kernel())
            del p2, v2, act
        del pos, vel, params
        torch.cuda.empty_cache()

    # per-episode phase, validity, unfused rollout, condition gather above 2^31 elements
    from tests.test_gpu_trajectory import PER_ROW
    name = [n for n in PER_ROW if "prodmp" in n][0]
    pc, bc, tc, dt, dur = PER_ROW[name]
    eng = make_engine(pc, bc, tc, dt, dur)
    T, D, P = eng.num_steps, eng.num_dof, eng.num_params
    B = (2**31 // (T * D)) + 4096
    print(name, "B", B, "elements", B * T * D)
    params, ip, iv = inputs(pc, bc, tc, 4096, seed=5)
    reps = (B + 4095) // 4096
    params = torch.tensor(params).repeat(reps, 1)[:B].contiguous().cuda()
    ip = torch.tensor(ip).repeat(reps, 1)[:B].contiguous().cuda(); iv = torch.tensor(iv).repeat(reps, 1)[:B].contiguous().cuda()
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    torch.cuda.synchronize()
    print("  kernel", eng.last_kernel())
    k = (B // 4096 - 1) * 4096
    assert torch.equal(pos[:4096], pos[k:k + 4096]) and torch.equal(vel[:4096], vel[k:k + 4096]), "periodic batch differs"
    rows = np.arange(B - 16, B)
    rp, rv = O.get_trajectory(pc, bc, tc, params[rows].cpu().numpy(), dur, dt, 0.0, ip[rows].cpu().numpy(), iv[rows].cpu().numpy(), dtype=np.float64)
    close(pos[rows].cpu().numpy(), rp, "pos"); close(vel[rows].cpu().numpy(), rv, "vel")
    valid = eng.traj_validity(pos, np.full(D, -1e9), np.full(D, 1e9))
    assert bool(valid.all())
    pos[B - 1, T - 1, D - 1] = 2e9
    valid = eng.traj_validity(pos, np.full(D, -1e9), np.full(D, 1e9))
    assert not bool(valid[B - 1]) and bool(valid[:B - 1].all())
    pos[B - 1, T - 1, D - 1] = 0.0
    spec = RolloutSpec("motor", D, 1.0, 0.1, -1.0, 1.0, plant="double_integrator", dt=dt)
    q, qd = ip.double().contiguous(), iv.double().contiguous()
    act = eng.pd_rollout(spec, pos, vel, q, qd)
    torch.cuda.synchronize()
    ra, rq, rqd = O.rollout(pos[rows].cpu().numpy(), vel[rows].cpu().numpy(), "motor", 1.0, 0.1, -1.0, 1.0, "double_integrator", dt, ip[rows].double().cpu().numpy(), iv[rows].double().cpu().numpy())
    assert np.array_equal(act[rows].cpu().numpy(), ra.astype(np.float32)) and np.array_equal(q[rows].cpu().numpy(), rq)
    seg = torch.full((B,), T, dtype=torch.int32, device="cuda")
    cp, cv = eng.condition_gather(pos, vel, seg)
    assert torch.equal(cp[rows], pos[rows, T - 1]) and torch.equal(cv[rows], vel[rows, T - 1])
    print("  per-episode phase, validity, pd_rollout, condition gather ok")


if __name__ == "__main__":
    main()
