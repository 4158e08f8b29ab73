#!/usr/bin/env python3
"""
The fused learned-phase launch at 65 536 episodes runs in one of two modes per PROCESS / allocation (34 - 36 % or 43 - 50 % of 8 TB/s:
profiles/r06_phase_fused_large.md).  This probe times the same launch on several fresh output allocations inside one process -- separate
tensors, one slab, a slab with skewed array bases, after emptying the caching allocator -- and prints addresses beside times.
(Its `--tiles` mode, the A/B of the removed `phase_tiles` option, went with the option: profiles/r06_phase_fused_large.md has the table.)
    python tools/alloc_mode_probe.py [B]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine  # noqa: E402
from closed_bench import graph_time  # noqa: E402
from learned_phase_bench import CASES, make_params  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    case = CASES["TT-ProDMP"]
    kw = case["kw"]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    eng = TrajectoryEngine(device=0, **kw)
    T, D, P = eng.num_steps, eng.num_dof, eng.num_params
    params = make_params(case, B, P, g).to(dev)
    ip = (0.2 * (torch.rand((B, D), generator=g) * 2 - 1)).to(dev)
    iv = torch.zeros((B, D), device=dev)
    pg, dg = case["gains"]
    static = RolloutSpec("motor", D, pg, dg, -1.0, 1.0, plant="static")
    q0, qd0 = ip.double().contiguous(), iv.double().contiguous()
    n = B * T * D
    nbytes = B * (P * 4 + 2 * D * 4 + 3 * T * D * 4 + 16 * D)
    keep = []

    def run(label, out):
        t = graph_time(lambda: eng.trajectory_actions(params, ip, iv, static, q0, qd0, 0.0, out=out), reps=4)
        fill = graph_time(lambda: [o.fill_(1.0) for o in out], reps=4)
        print(f"{label:34s} {t * 1e6:7.1f} us = {nbytes / t / 8e12 * 100:4.1f} %   fill of the same arrays {3 * n * 4 / fill / 1e12:4.2f} TB/s   "
              f"bases {[hex(o.data_ptr()) for o in out]}", flush=True)

    for r in range(1 if "--skews" in sys.argv else 4):
        out = tuple(torch.empty((B, T, D), device=dev) for _ in range(3))
        keep.append(out)
        run(f"separate tensors #{r}", out)
    skews = (0, 4096, 65536, 1 << 20, (1 << 21) + 4096)
    if "--skews" in sys.argv:
        M, K = 1 << 20, 1 << 10
        skews = (2 * M, 2 * M + 4 * K, 2 * M + 8 * K, 2 * M + 64 * K, 4 * M + 4 * K, 4 * M + 8 * K, M + 4 * K, 512 * K + 4 * K, 8 * K, 12 * K,
                 6 * M + 4 * K, 6 * M + 12 * K, 2 * M + 2 * K, 2 * M + 512, 2 * M + 256, 8 * M + 4 * K, 16 * M + 4 * K, 3 * M, 3 * M + 4 * K)
    for skew in skews:
        slab = torch.empty(3 * n + 3 * (skew // 4) + 1024, device=dev)
        if "--skews" not in sys.argv:
            keep.append(slab)
        out = tuple(slab[i * (n + skew // 4):i * (n + skew // 4) + n].view(B, T, D) for i in range(3))
        run(f"one slab, skew {skew}", out)
    keep.clear()
    torch.cuda.empty_cache()
    for r in range(3):
        out = tuple(torch.empty((B, T, D), device=dev) for _ in range(3))
        run(f"after empty_cache #{r}", out)
        del out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
