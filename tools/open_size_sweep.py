import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch, numpy as np
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from closed_bench import CFG2, PG, DG, graph_time
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(device=0, **CFG2)
spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
print("| B | kernel | us | of 8 TB/s | us per 1024 episodes |")
for B in [int(a) for a in sys.argv[1:]]:
    params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(3))
    t = graph_time(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out))
    print(f"| {B} | `{eng.last_kernel()}` | {t*1e6:.2f} | {B*8624/t/8e12*100:.1f} % | {t*1e6/B*1024:.2f} |", flush=True)
