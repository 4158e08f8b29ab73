"""k_traj_phase<prodmp> (cfg2 + learned tau) at a few thousand episodes: row table in LDS / gathered from L2, per-episode / flat rounds, chunk sizes.
    python tools/prodmp_phase_probe.py [B ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import TrajectoryEngine, _lib
from tools.closed_bench import CFG2, graph_time
eng = TrajectoryEngine(device=0, **dict(CFG2, learn_tau=True, tau_bound=(1.0, 2.0)))
g = torch.Generator().manual_seed(0)
print("| B | options | kernel | us | of 8 TB/s |")
print("|---|---|---|---|---|")
for B in [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]:
    params = torch.randn((B, eng.num_params), generator=g).cuda(); params[:, 0] = torch.rand(B, generator=g).cuda() + 1.0
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(2))
    for opts in ({}, {"phase_table": 0}, {"phase_flat": 1}, {"phase_flat": 0}, {"phase_table": 0, "phase_flat": 1}, {"phase_table": 0, "phase_flat": 0},
                 {"phase_flat": 1, "phase_chunk": 2}, {"phase_flat": 1, "phase_chunk": 3}, {"phase_table": 0, "phase_flat": 1, "phase_chunk": 2}, {"phase": 0}):
        _lib.reset_options()
        for k, v in opts.items(): _lib.set_option(k, v)
        t = graph_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out))
        print(f"| {B} | {opts or 'auto'} | `{eng.last_kernel()}` | {t*1e6:.1f} | {B*5828/t/8e12*100:.1f} % |", flush=True)
