"""k_traj_ring with plain / write-through stores beyond the memory-side cache (trajectory only and + actions).  python tools/ring_wt_probe.py [B ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from closed_bench import CFG2, PG, DG, graph_time
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(device=0, **CFG2)
spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
spec_c = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
print("| B | mode | stores | us | of 8 TB/s |")
print("|---|---|---|---|---|")
for B in [int(a) for a in sys.argv[1:]] or [49152, 65536, 98304, 131072, 262144]:
    params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(3))
    for mode, nb in (("traj", 5824), ("act", 8624), ("closed", 8624)):
        fn = {"traj": lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]), "act": lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out),
              "closed": lambda: eng.trajectory_rollout(params, ip, iv, spec_c, cp, cv, out=out)}[mode]
        for rnd in range(2):
            for wt in (0, 1):
                _lib.reset_options(); _lib.set_option("ring", 1); _lib.set_option("write_through", wt)
                t = graph_time(fn, reps=10, rounds=5)
                print(f"| {B} | {mode} | {'write-through' if wt else 'plain'} | {t*1e6:.1f} | {B*nb/t/8e12*100:.1f} % |", flush=True)
    del out
