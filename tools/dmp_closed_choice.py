import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from tools.closed_bench import graph_time
from tools.dmp_bench import CFG3
eng = TrajectoryEngine(device=0, **CFG3)
g = torch.Generator().manual_seed(0)
spec_d = RolloutSpec("motor", 7, 1.0, 0.1, -1.0, 1.0, plant="double_integrator", dt=0.02)
for B in (16384, 32768, 49152, 65536, 131072):
    params = torch.randn((B, eng.num_params), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    pos = torch.empty((B, 200, 7), device="cuda"); vel = torch.empty_like(pos); act = torch.empty_like(pos)
    q, qd = ip.double().contiguous(), iv.double().contiguous()
    fn = lambda: eng.trajectory_rollout(params, ip, iv, spec_d, q, qd, out=(pos, vel, act))
    for opts in ({}, {"ring": 0}, {"ring": 0, "quad": 2}, {"ring": 0, "quad": 3}, {"ring": 1}):
        _lib.reset_options()
        for k, v in opts.items(): _lib.set_option(k, v)
        fn(); torch.cuda.synchronize(); kern = eng.last_kernel()
        t = graph_time(fn)
        print(f"| {B} | {opts or 'auto'} | `{kern}` | {t*1e6:.1f} | {B*(224+3*5600)/t/8e12*100:.1f} % |", flush=True)
