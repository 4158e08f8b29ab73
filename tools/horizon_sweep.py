import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch, numpy as np
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from closed_bench import CFG2, PG, DG, graph_time
g = torch.Generator().manual_seed(0)
for B in (4096, 8192):
  for T in (96, 97, 100, 104, 112):
    eng = TrajectoryEngine(device=0, **dict(CFG2, duration=T * 0.02))
    params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
    q, qd = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, T, 7), device="cuda") for _ in range(3))
    t = graph_time(lambda: eng.trajectory_rollout(params, ip, iv, spec, q, qd, out=out))
    print(B, T, eng.last_kernel(), f"{t*1e6:.2f} us", flush=True)
