import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from closed_bench import CFG2, PG, DG, graph_time
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(device=0, **CFG2)
spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
print("| B | mode | options | kernel | us | of 8 TB/s |")
for B in [int(a) for a in sys.argv[1:]]:
    params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(3))
    for mode, nb in (("traj", 5824), ("act", 8624)):
        fn = (lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2])) if mode == "traj" else (lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out))
        for opts in ({}, {"mapping": 1}, {"flat": 1}, {"ring": 1}):
            _lib.reset_options()
            for k, v in opts.items(): _lib.set_option(k, v)
            try:
                t = graph_time(fn, reps=10, rounds=5)
                print(f"| {B} | {mode} | {opts or 'auto'} | `{eng.last_kernel()}` | {t*1e6:.1f} | {B*nb/t/8e12*100:.1f} % |", flush=True)
            except Exception as e:
                print(f"| {B} | {mode} | {opts} | failed {e} |", flush=True)
    _lib.reset_options()
    del out
