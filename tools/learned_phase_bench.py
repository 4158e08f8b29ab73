#!/usr/bin/env python3
"""
The reference's learned-phase configurations at their REAL shapes, through every entry point of a step
(`profiles/r06_sweep.md`):
    TT-ProDMP          envs/mujoco/table_tennis/mp_wrapper.py:32-57   (learn tau + delay, nb = 3, alpha 25, T = 350, dt 0.008)
    TT-ProDMP Replan   :91-121                                       (nb = 2, t % 50 == 0, max_planning_times 3)
    BeerPong-ProMP     envs/mujoco/beerpong/mp_wrapper.py:9-25        (learn tau, nb = 2 + 2 zero-start, T = 300, dt 0.01)
    cfg5 TT-ProMP      table_tennis/mp_wrapper.py:11-30               (shared phase; here for the validity gate's flow)
rows: trajectory only | + actions (frozen state) | closed-loop step | verbose < 2 step | gated step (validity + penalty inside)
    python tools/learned_phase_bench.py [B ...] [key=value ...]
Entry points that the library declines (MPK_ENOTIMPL) are reported as such; the kernel column shows what ran last.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from closed_bench import graph_time  # noqa: E402

TT_P = 0.5 * np.array([1.0, 4.0, 2.0, 4.0, 1.0, 4.0, 1.0])
TT_D = 0.5 * np.array([0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1])
BP_P = np.array([1.5, 5, 2.55, 3, 2., 2, 1.25])
BP_D = np.array([0.02333333, 0.1, 0.0625, 0.08, 0.03, 0.03, 0.0125])
JNT_LOW = np.array([-2.6, -2.0, -2.8, -0.9, -4.8, -1.6, -2.2])      # table_tennis_utils.py:3-4
JNT_HIGH = np.array([2.6, 2.0, 2.8, 3.1, 1.3, 1.6, 2.2])

CASES = {
    "TT-ProDMP": dict(
        kw=dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=3, dt=0.008, duration=2.8,
                tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.8, 1.5), delay_bound=(0.05, 0.15),
                basis_alpha=25.0, basis_bandwidth_factor=3.0, weights_scale=0.7, auto_scale_basis=True, relative_goal=True,
                disable_goal=True),
        gains=(TT_P, TT_D), n_phase=2, every=None, mpt=1, gate=True),
    "TT-ProDMP-Replan": dict(
        kw=dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=2, dt=0.008, duration=2.8,
                tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.8, 1.5), delay_bound=(0.05, 0.15),
                basis_alpha=25.0, basis_bandwidth_factor=3.0, auto_scale_basis=True, goal_offset=1.0),
        gains=(TT_P, TT_D), n_phase=2, every=50, mpt=3, gate=True),
    "BeerPong-ProMP": dict(
        kw=dict(mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=7, num_basis=2, num_basis_zero_start=2,
                dt=0.01, duration=3.0, tau=3.0, learn_tau=True, tau_bound=(0.02, 3.0), basis_bandwidth_factor=3.0),
        gains=(BP_P, BP_D), n_phase=1, every=None, mpt=1, gate=False),
    "cfg5 TT-ProMP": dict(
        kw=dict(mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=7, num_basis=3, num_basis_zero_start=1,
                num_basis_zero_goal=1, dt=0.008, duration=2.8, tau=2.8),
        gains=(TT_P, TT_D), n_phase=0, every=None, mpt=1, gate=True),
}


def make_params(case, B, P, g):
    """weights small enough that most plans stay inside the joint limits (the gate's common case), phase inside its bounds"""
    params = 0.3 * torch.randn((B, P), generator=g)
    kw = case["kw"]
    if case["n_phase"] >= 1:
        lo, hi = kw["tau_bound"]
        lo = max(lo, 0.25 * kw["duration"])
        params[:, 0] = torch.rand(B, generator=g) * (hi - lo) + lo
    if case["n_phase"] >= 2:
        lo, hi = kw["delay_bound"]
        params[:, 1] = torch.rand(B, generator=g) * (hi - lo) + lo
    return params


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 8192, 65536]
    only = [a for a in sys.argv[1:] if a in CASES]
    for kv in sys.argv[1:]:
        if "=" in kv:
            k, v = kv.split("=")
            _lib.set_option(k, int(v))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    print(f"lib: {_lib.LIB_PATH}  stamp {_lib.load().mpk_source_hash().decode()[:12]}")
    print("| config | batch | entry point | kernel (last) | launches | us | episodes/s | alg. bytes / episode | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, case in CASES.items():
        if only and name not in only:
            continue
        kw = case["kw"]
        eng = TrajectoryEngine(device=0, **kw)
        T, D, P = eng.num_steps, eng.num_dof, eng.num_params
        pg, dg = case["gains"]
        for B in batches:
            params = make_params(case, B, P, g).to(dev)
            ip = (0.2 * (torch.rand((B, D), generator=g) * 2 - 1)).to(dev)
            iv = torch.zeros((B, D), device=dev)
            out = tuple(torch.empty((B, T, D), device=dev) for _ in range(3))
            reps = 20 if B * T <= 8192 * 350 else 4
            n_in = P * 4 + 2 * D * 4
            arr = T * D * 4
            static = RolloutSpec("motor", D, pg, dg, -1.0, 1.0, plant="static")
            closed = RolloutSpec("motor", D, pg, dg, -1.0, 1.0, plant="double_integrator", dt=kw["dt"])
            q0, qd0 = ip.double().contiguous(), iv.double().contiguous()
            q, qd = q0.clone(), qd0.clone()
            i32 = dict(dtype=torch.int32, device=dev)
            traj_steps, plan_steps = torch.zeros(B, **i32), torch.zeros(B, **i32)
            done = torch.zeros(B, dtype=torch.uint8, device=dev)
            # horizon / planning budget out of reach: every call executes `every` steps (the whole plan, or the Replan variant's 50)
            # without memsets of the integer state inside the timed graph (tools/closed_bench.py)
            every = case["every"] or T
            mpt = horizon = 2 ** 30

            def reset_state():
                pass

            def f_traj():
                eng.trajectory(params, ip, iv, 0.0, out=out[:2])

            def f_act():
                eng.trajectory_actions(params, ip, iv, static, q0, qd0, 0.0, out=out)

            def f_closed():
                reset_state()
                eng.replan_step(params, ip, iv, closed, q, qd, traj_steps, plan_steps, done, every, mpt, horizon, condition=True, out=out)

            def f_lean():
                reset_state()
                eng.episode_return(params, ip, iv, closed, q, qd, replan=(traj_steps, plan_steps, done, every, mpt, horizon), condition=True)

            rows = [("trajectory", f_traj, n_in + 2 * arr), ("+ actions (frozen state)", f_act, n_in + 3 * arr + 16 * D),
                    ("closed-loop step (replan_step)", f_closed, n_in + 3 * arr + 32 * D),
                    ("verbose < 2 step (episode_return)", f_lean, n_in + 32 * D + 16)]
            if case["gate"] and hasattr(eng, "_gate"):
                valid = torch.empty(B, dtype=torch.uint8, device=dev)
                pen = torch.empty(B, dtype=torch.float64, device=dev)

                def f_gated():
                    # (the done bytes are cleared inside the timed graph -- a 1 KB memset --: a plan that violates a limit FINISHES its
                    # episode, and replaying full-horizon steps on a batch that keeps such episodes measures the masked chain of their
                    # waves, not the gate: profiles/r06_finished_episodes.md)
                    done.zero_()
                    eng.replan_step(params, ip, iv, closed, q, qd, traj_steps, plan_steps, done, every, mpt, horizon, condition=True, out=out,
                                    gate=dict(pos_low=JNT_LOW, pos_high=JNT_HIGH, check_tau_delay=case["n_phase"] == 2,
                                              tau_bound=kw.get("tau_bound"), delay_bound=kw.get("delay_bound"), valid=valid, penalty=pen))
                rows.append(("gated closed-loop step (validity + penalty inside)", f_gated, n_in + 3 * arr + 32 * D))

                def f_gated_lean():
                    done.zero_()
                    eng.episode_return(params, ip, iv, closed, q, qd, replan=(traj_steps, plan_steps, done, every, mpt, horizon), condition=True,
                                       gate=dict(pos_low=JNT_LOW, pos_high=JNT_HIGH, check_tau_delay=case["n_phase"] == 2,
                                                 tau_bound=kw.get("tau_bound"), delay_bound=kw.get("delay_bound"), valid=valid, penalty=pen))
                rows.append(("gated verbose < 2 step", f_gated_lean, n_in + 32 * D + 16 + 9))
            if case["gate"]:
                valid_pen = {}

                def f_gated3():     # today's flow: trajectory -> k_validity (re-reads pos) -> advance -> rollout -> gather
                    reset_state()
                    pos, vel = eng.trajectory(params, ip, iv, 0.0, out=out[:2])
                    chk = case["n_phase"] == 2
                    v, pen = eng.traj_validity(pos, JNT_LOW, JNT_HIGH, params if chk else None, kw.get("tau_bound") if chk else None,
                                               kw.get("delay_bound") if chk else None, with_penalty=True)
                    done.bitwise_or_((~v).to(torch.uint8))
                    seg = eng.replan_advance(traj_steps, plan_steps, done, every, mpt, horizon)
                    eng.pd_rollout(closed, pos, vel, q, qd, n_steps=seg, out=out[2])
                    eng.condition_gather(pos, vel, seg)
                    valid_pen["v"] = v
                rows.append(("gated step, separate launches (trajectory, validity, advance, rollout, gather)", f_gated3, n_in + 3 * arr + 32 * D))
            for label, fn, nbytes in rows:
                try:
                    fn()
                    torch.cuda.synchronize()
                except NotImplementedError as e:
                    print(f"| {name} | {B} | {label} | MPK_ENOTIMPL ({str(e)[:60]}) | - | - | - | {nbytes} | - |")
                    continue
                kern = eng.last_kernel()
                if "gated" in label:
                    nv = int((done != 0).sum())
                    label += f" [{B - nv} of {B} valid]"
                    done.zero_()
                t = graph_time(fn, reps=reps)
                print(f"| {name} | {B} | {label} | `{kern}` | | {t * 1e6:.1f} | {B / t:.3e} | {nbytes} | {B * nbytes / t / 8e12 * 100:.1f} % |", flush=True)
            del out, params
        del eng


if __name__ == "__main__":
    main()
