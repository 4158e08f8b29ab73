#!/usr/bin/env python3
"""
Gated against ungated closed-loop step, eager launches under rocprofv3 (one kernel name per variant via the handle option trick is not
needed: the two variants alternate in blocks of N launches, the kernel trace orders them):
    rocprofv3 --kernel-trace --stats -d gpurun_out/r06/gate_probe -- python3 tools/gate_probe.py TT-ProDMP 8192 30
prints event-timed per-launch averages itself.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402
from learned_phase_bench import CASES, JNT_HIGH, JNT_LOW, make_params  # noqa: E402


def main():
    name, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    scale = 1.0
    live = 0
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        if k == "live":
            live = int(v)           # 1: the done bytes are cleared before EVERY launch (both variants): what a violating PLAN costs, without the
                                    # finished episodes it leaves behind in later launches
        elif k == "scale":
            scale = float(v)        # weights x scale: small enough and no plan leaves the joint limits (the gate's cost without a violating wave)
        else:
            _lib.set_option(k, int(v))
    case = CASES[name]
    kw = case["kw"]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    eng = TrajectoryEngine(device=0, **kw)
    T, D, P = eng.num_steps, eng.num_dof, eng.num_params
    params = make_params(case, B, P, g)
    params[:, case["n_phase"]:] *= scale
    params = params.to(dev)
    ip = (0.2 * (torch.rand((B, D), generator=g) * 2 - 1)).to(dev)
    iv = torch.zeros((B, D), device=dev)
    out = tuple(torch.empty((B, T, D), device=dev) for _ in range(3))
    pg, dg = case["gains"]
    closed = RolloutSpec("motor", D, pg, dg, -1.0, 1.0, plant="double_integrator", dt=kw["dt"])
    q, qd = ip.double().contiguous(), iv.double().contiguous()
    i32 = dict(dtype=torch.int32, device=dev)
    ts, ps = torch.zeros(B, **i32), torch.zeros(B, **i32)
    dn = torch.zeros(B, dtype=torch.uint8, device=dev)
    valid = torch.empty(B, dtype=torch.uint8, device=dev)
    pen = torch.empty(B, dtype=torch.float64, device=dev)
    every = case["every"] or T
    big = 2 ** 30
    gate = dict(pos_low=JNT_LOW, pos_high=JNT_HIGH, check_tau_delay=case["n_phase"] == 2, tau_bound=kw.get("tau_bound"),
                delay_bound=kw.get("delay_bound"), valid=valid, penalty=pen)

    def run(gt, lean):
        if live:
            dn.zero_()
        if lean:
            eng.episode_return(params, ip, iv, closed, q, qd, replan=(ts, ps, dn, every, big, big), condition=True, gate=gt)
        else:
            eng.replan_step(params, ip, iv, closed, q, qd, ts, ps, dn, every, big, big, condition=True, out=out, gate=gt)

    for lean in (False, True):
        for label, gt in (("ungated", None), ("gated", gate), ("ungated", None), ("gated", gate)):
            for _ in range(5):
                run(gt, lean)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(N):
                run(gt, lean)
            b.record()
            torch.cuda.synchronize()
            print(f"{name} B={B} {'lean' if lean else 'step'} {label}: {a.elapsed_time(b) / N * 1e3:.1f} us per launch ({eng.last_kernel()}), "
                  f"{int((dn != 0).sum())} done")
            dn.zero_()


if __name__ == "__main__":
    main()
