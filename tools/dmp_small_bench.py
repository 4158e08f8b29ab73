import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import TrajectoryEngine, _lib
from closed_bench import graph_time
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=4.0, tau=4.0, alpha_phase=2.0, dmp_alpha=25.0, device=0)
print("| B | options | kernel | us | of 8 TB/s |")
for B in [int(a) for a in sys.argv[1:]]:
    params = torch.randn((B, eng.num_params), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    out = tuple(torch.empty((B, 200, 7), device="cuda") for _ in range(2))
    for opts in ({}, {"quad": 0}, {"quad": 4}, {"quad": 3}, {"quad": 2}):
        _lib.reset_options()
        for k, v in opts.items(): _lib.set_option(k, v)
        t = graph_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out))
        print(f"| {B} | {opts or 'auto'} | `{eng.last_kernel()}` | {t*1e6:.2f} | {B*11424/t/8e12*100:.1f} % |", flush=True)
