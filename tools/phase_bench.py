#!/usr/bin/env python3
"""
The per-episode-phase kernels (learned tau / delay: the reference's TableTennis / BeerPong configurations, per-episode
init_time after replanning drift): cfg2 with learned tau (k_traj_phase<prodmp>), cfg5 with learned tau + delay
(k_traj_phase<promp>), cfg3 with learned tau (k_traj_phase<dmp>).   python tools/phase_bench.py [B ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import TrajectoryEngine, _lib  # noqa: E402
from closed_bench import graph_time  # noqa: E402
from run_cfg import KW  # noqa: E402

CASES = {
    "cfg2' prodmp learn_tau": ("cfg2tau", 1),
    "cfg5' promp learn_tau+delay": ("cfg5tau", 2),
    "cfg3' dmp learn_tau": (dict(KW["cfg3"], learn_tau=True, tau_bound=(2.0, 4.0)), 1),
}
# other horizons of the cfg2' shape (chunk-size model of the prodmp kernel):  python tools/phase_bench.py horizons ...
HORIZONS = {f"cfg2' prodmp learn_tau, T = {T}": (dict(KW["cfg2tau"], duration=T * 0.02), 1) for T in (64, 128, 150, 350)}


def main():
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 65536]
    for kv in sys.argv[1:]:
        if "=" in kv:
            k, v = kv.split("=")
            _lib.set_option(k, int(v))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    print("| config | batch | kernel | us | trajectories/s | GB/s (alg.) | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    cases = dict(CASES)
    if "horizons" in sys.argv:
        cases = HORIZONS
    if "prodmp" in sys.argv:
        cases = {k: v for k, v in CASES.items() if "prodmp" in k}
    for name, (kw, n_ph) in cases.items():
        kw = KW[kw] if isinstance(kw, str) else kw
        eng = TrajectoryEngine(device=0, **kw)
        T, D, P = eng.num_steps, eng.num_dof, eng.num_params
        lo, hi = kw["tau_bound"]
        for B in batches:
            params = torch.randn((B, P), generator=g)
            params[:, 0] = torch.rand(B, generator=g) * (hi - lo) * 0.6 + lo + 0.3 * (hi - lo)
            if n_ph > 1:
                params[:, 1] = torch.rand(B, generator=g) * 0.1 + 0.05
            params = params.to(dev)
            ip = (torch.rand((B, D), generator=g) * 2 - 1).to(dev)
            iv = torch.zeros((B, D), device=dev)
            out = (torch.empty((B, T, D), device=dev), torch.empty((B, T, D), device=dev))
            t = graph_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out), reps=20 if B <= 16384 else 4)
            nbytes = P * 4 + 2 * D * 4 + 2 * T * D * 4
            print(f"| {name} | {B} | `{eng.last_kernel()}` | {t * 1e6:.1f} | {B / t:.3e} | {B * nbytes / t / 1e9:.0f} | "
                  f"{B * nbytes / t / 8e12 * 100:.1f} % |")
            del out, params


if __name__ == "__main__":
    main()
