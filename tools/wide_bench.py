#!/usr/bin/env python3
"""
k_traj_wide (more than 16 contraction columns, shared phase): the reference's num_basis = 1000 example
(examples/examples_movement_primitives.py:67: fancy_ProMP/Reacher5d-v0, 5 DoF, 200 steps) and a few other widths.
This is the one place of the path where the matrix cores are the bound: 2 T K D flop against (K D + 2 T D) 4 bytes per
episode.  Reports time per launch, TFLOP/s of the contraction against the fp32 MFMA peak (157.3 TF) and GB/s.
    python tools/wide_bench.py [B ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fancy_gym_amd import TrajectoryEngine  # noqa: E402

PEAK_TF = 157.3


def timed(fn, n=10, warm=3):
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


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 4096, 16384]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    print("| config | K (columns) | B | kernel | us | episodes/s | TFLOP/s (2 T K D n_out) | of 157.3 TF | GB/s (alg.) |")
    print("|---|---|---|---|---|---|---|---|---|")
    cases = [("promp zero_rbf 5 DoF x 200 steps, num_basis 1000 (the reference's example)", "promp", "linear", "zero_rbf", 5, 1000, 0.02, 4.0, dict(tau=4.0, num_basis_zero_start=1, num_basis_zero_goal=0)),
             ("promp rbf 7 DoF x 100 steps, num_basis 64", "promp", "linear", "rbf", 7, 64, 0.02, 2.0, dict(tau=2.0)),
             ("prodmp 7 DoF x 100 steps, num_basis 125", "prodmp", "exp", "prodmp", 7, 125, 0.02, 2.0, dict(tau=1.5, alpha_phase=3.0, basis_alpha=10.0)),
             ("dmp 7 DoF x 200 steps, num_basis 256", "dmp", "exp", "rbf", 7, 256, 0.02, 4.0, dict(tau=4.0, alpha_phase=2.0))]
    for name, mp, ph, bs, D, nb, dt, dur, kw in cases:
        eng = TrajectoryEngine(mp, ph, bs, D, nb, dt=dt, duration=dur, device=0, **kw)
        T, P = eng.num_steps, eng.num_params
        K = nb + {"promp": 1 if bs == "zero_rbf" else 0, "prodmp": 3, "dmp": 0}[mp]
        nout = 2 if mp == "prodmp" else 1
        for B in batches:
            g = torch.Generator().manual_seed(0)
            params = torch.randn((B, P), generator=g).to(dev)
            ip = (torch.rand((B, D), generator=g) * 2 - 1).to(dev)
            iv = torch.zeros((B, D), device=dev)
            out = (torch.empty((B, T, D), device=dev), torch.empty((B, T, D), device=dev))
            t = timed(lambda: eng.trajectory(params, ip, iv, 0.0, out=out))
            flop = 2.0 * T * K * D * nout * B
            nbytes = B * (P * 4 + 2 * D * 4 + 2 * T * D * 4)
            print(f"| {name} | {K} | {B} | `{eng.last_kernel()}` | {t * 1e6:.1f} | {B / t:.3e} | {flop / t / 1e12:.1f} | "
                  f"{flop / t / 1e12 / PEAK_TF * 100:.1f} % | {nbytes / t / 1e9:.0f} |")
            del params, ip, iv, out
    # more than 16 DoF with a shared phase: column groups of k_traj_wide against the per-episode-phase kernels the same
    # shapes took before (timed by passing the init_time as a per-episode vector)
    print()
    print("| config (D > 16, shared phase) | B | kernel | us | GB/s (alg.) | of 8 TB/s | per-episode-phase kernel | us |")
    print("|---|---|---|---|---|---|---|---|")
    cases = [("prodmp 32 DoF x 100 steps, num_basis 10", "prodmp", "exp", "prodmp", 32, 10, 0.02, 2.0, dict(tau=1.5, alpha_phase=3.0, basis_alpha=10.0)),
             ("promp zero_rbf 24 DoF x 100 steps, num_basis 8", "promp", "linear", "zero_rbf", 24, 8, 0.02, 2.0, dict(tau=2.0, num_basis_zero_start=1, num_basis_zero_goal=0)),
             ("dmp 20 DoF x 200 steps, num_basis 12", "dmp", "exp", "rbf", 20, 12, 0.02, 4.0, dict(tau=4.0, alpha_phase=2.0)),
             ("prodmp 48 DoF x 100 steps, num_basis 40", "prodmp", "exp", "prodmp", 48, 40, 0.02, 2.0, dict(tau=1.5, alpha_phase=3.0, basis_alpha=10.0))]
    for name, mp, ph, bs, D, nb, dt, dur, kw in cases:
        eng = TrajectoryEngine(mp, ph, bs, D, nb, dt=dt, duration=dur, device=0, **kw)
        T, P = eng.num_steps, eng.num_params
        for B in batches:
            g = torch.Generator().manual_seed(0)
            params = torch.randn((B, P), generator=g).to(dev)
            ip = (torch.rand((B, D), generator=g) * 2 - 1).to(dev)
            iv = torch.zeros((B, D), device=dev)
            it = torch.zeros((B,), device=dev)
            out = (torch.empty((B, T, D), device=dev), torch.empty((B, T, D), device=dev))
            t = timed(lambda: eng.trajectory(params, ip, iv, 0.0, out=out))
            k0 = eng.last_kernel()
            t1 = timed(lambda: eng.trajectory(params, ip, iv, it, out=out))
            nbytes = B * (P * 4 + 2 * D * 4 + 2 * T * D * 4)
            print(f"| {name} | {B} | `{k0}` | {t * 1e6:.1f} | {nbytes / t / 1e9:.0f} | {nbytes / t / 8e12 * 100:.1f} % | "
                  f"`{eng.last_kernel()}` | {t1 * 1e6:.1f} |")
            del params, ip, iv, out


if __name__ == "__main__":
    main()
