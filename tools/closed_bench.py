#!/usr/bin/env python3
"""
The fused closed-loop step (BlackBoxWrapper.step on the GPU-resident double integrator) per kernel variant:
  full   mpk_trajectory_rollout, cfg2 shape, all 100 steps executed
  plan   mpk_replan_step, cfg4 shape (P = 35, 25 of 100 steps executed, condition gather, integer state)
timed as a captured graph of 20 launches (HIP events), per batch size and per option set.
    python tools/closed_bench.py [--ring] [B ...]        (MPK_LIB=<other build> for A/B runs of two builds; --ring: the ring's geometries)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402

PG = 0.01 * np.array([120., 120., 120., 120., 50., 30., 10.])
DG = 0.01 * np.array([10., 10., 10., 10., 6., 5., 3.])
CFG2 = dict(mp_type="prodmp", phase_type="exp", basis_type="prodmp", num_dof=7, num_basis=5, dt=0.02, duration=2.0,
            tau=1.5, alpha_phase=3.0, basis_bandwidth_factor=2.0, basis_alpha=10.0)
CFG4 = dict(CFG2, basis_bandwidth_factor=3.0, weights_scale=0.3, goal_scale=0.3, auto_scale_basis=True, disable_goal=True)


def capture(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return g


def replay_ms(g):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def time_rows(graphs, reps=20, rounds=9, busy_ms=60.0):
    """Every row of a (batch, step) pair is warmed by the time the GPU was BUSY with it (event time, not wall time: the shader
    clock needs ~20 ms of load to settle, profiles/r02_clock_probe.md), and the timed rounds alternate over the rows, so that no
    row owns the first or the last slot (round 3 measured "auto" 32.6 us first and 29.8 us last at plan / 16 384)."""
    for gi, g in enumerate(graphs):
        if os.environ.get('CB_TRACE'): print(f'(warming row {gi})', file=sys.stderr, flush=True)
        busy = 0.0
        while busy < busy_ms:
            busy += replay_ms(g)
    ts = [[] for _ in graphs]
    for r in range(rounds):
        order = range(len(graphs)) if r % 2 == 0 else reversed(range(len(graphs)))
        for i in order:
            if os.environ.get('CB_TRACE'): print(f'(round {r} row {i})', file=sys.stderr, flush=True)
            replay_ms(graphs[i])                        # one untimed replay: the row before left other lines in the caches
            ts[i].append(replay_ms(graphs[i]) * 1e-3 / reps)
    return [float(np.median(t)) for t in ts]


def graph_time(fn, reps=20, rounds=7):
    """one row on its own (the other tools' entry point): captured graph of `reps` launches, warmed by GPU-busy time, median"""
    return time_rows([capture(fn, reps)], reps=reps, rounds=rounds)[0]


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 8192, 16384]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    variants = [("auto", {}), ("pipe=1", {"pipe": 1}), ("pipe=0", {"pipe": 0}), ("split", {"split": 1}), ("quad", {"quad": 2}), ("duo", {"quad": 3}), ("mono", {"quad": 4}), ("auto (again)", {})]
    if "--ring" in sys.argv:      # the closed loop on the ring (k_traj_ring<.., closed>) and its launch geometries against the lane-quarter kernels
        R = {"ring": 1}
        variants = [("auto", {}), ("quad", {"quad": 2}), ("duo", {"quad": 3}), ("ring", dict(R)), ("ring tickets", dict(R, ring_dbg=4)),
                    ("ring, consumers store", dict(R, ring_dbg=8, ring_nc=4)), ("ring ns2", dict(R, ring_ns=2, ring_np=7)),
                    ("ring no stores", dict(R, ring_dbg=2, ablations=1)), ("ring no compute", dict(R, ring_dbg=1, ablations=1)), ("auto (again)", {})]
    if "--ring-sweep" in sys.argv:
        variants = [("duo", {"quad": 3})]
        for dbg in (0, 4):
            for np_ in (6, 8, 10):
                for ns in (1, 2, 3):
                    for nc in (3, 4):
                        variants.append((f"np{np_} ns{ns} nc{nc}{' static' if dbg else ''}", dict(ring=1, ring_np=np_, ring_ns=ns, ring_nc=nc, ring_dbg=dbg)))
        if "--full-only" in sys.argv:
            pass
    if "--ring-ablate" in sys.argv:   # measurements only (outputs unwritten): ring_dbg 1 no production / chain, 2 no stores
        base = {"ring": 1, "ablations": 1}         # (the output-dropping bits of ring_dbg are masked out without it)
        variants = [("ring", dict(base)), ("ring static", dict(base, ring_dbg=4)), ("no compute", dict(base, ring_dbg=1)), ("no stores", dict(base, ring_dbg=2)),
                    ("neither", dict(base, ring_dbg=3)), ("static no compute", dict(base, ring_dbg=5)), ("static no stores", dict(base, ring_dbg=6)),
                    ("static neither", dict(base, ring_dbg=7)), ("np7 ns1 nc4", dict(base, ring_np=7, ring_ns=1, ring_nc=4)),
                    ("np6 ns1 nc4 static", dict(base, ring_np=6, ring_ns=1, ring_nc=4, ring_dbg=4)), ("np7 ns1 nc4 static", dict(base, ring_np=7, ring_ns=1, ring_nc=4, ring_dbg=4)),
                    ("np8 ns1 nc3 static", dict(base, ring_np=8, ring_ns=1, ring_nc=3, ring_dbg=4))]
    print(f"lib: {_lib.LIB_PATH}")
    print("| step | batch | variant | kernel | us | episodes-or-plans/s | GB/s (alg.) | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for B in batches:
        for name, kw, P, nbytes in (("full", CFG2, 42, 224 + 3 * 2800), ("plan", CFG4, 35, 196 + 3 * 2800)):
            if "--full-only" in sys.argv and name != "full":
                continue
            eng = TrajectoryEngine(device=0, **kw)
            params = torch.randn((B, P), generator=g).to(dev)
            ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
            iv = torch.zeros((B, 7), device=dev)
            spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
            q, qd = ip.double().contiguous(), iv.double().contiguous()
            out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
            ts = torch.zeros(B, dtype=torch.int32, device=dev)
            ps = torch.zeros(B, dtype=torch.int32, device=dev)
            dn = torch.zeros(B, dtype=torch.uint8, device=dev)
            if name == "full":
                fn = lambda: eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out)      # noqa: E731
            else:
                # horizon / planning budget out of reach: every call executes the 25 steps up to the next multiple of 25,
                # like each of cfg4's four plans, without memsets of the integer state inside the timed graph
                def fn():
                    eng.replan_step(params, ip, iv, spec, q, qd, ts, ps, dn, 25, 2 ** 30, 2 ** 30, condition=True, out=out)
            rows = []
            for vn, opts in variants:
                _lib.reset_options()
                for k, v in opts.items():
                    _lib.set_option(k, v)
                try:
                    rows.append((vn, capture(fn), eng.last_kernel()))     # the options are read at launch = at capture
                    print(f"(captured {name} {B} {vn})", file=sys.stderr, flush=True)
                except Exception as e:  # noqa: BLE001
                    print(f"| {name} | {B} | {vn} | failed: {e} |")
            times = time_rows([r[1] for r in rows])
            for (vn, _, kern), t in zip(rows, times):
                print(f"| {name} | {B} | {vn} | `{kern}` | {t * 1e6:.2f} | {B / t:.3e} | "
                      f"{B * nbytes / t / 1e9:.0f} | {B * nbytes / t / 8e12 * 100:.1f} % |")
            _lib.reset_options()
            del eng, out


if __name__ == "__main__":
    main()
