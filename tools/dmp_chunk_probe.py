import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import TrajectoryEngine, _lib
from tools.closed_bench import graph_time
eng = TrajectoryEngine(device=0, mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=4.0, tau=4.0,
                       alpha_phase=2.0, dmp_alpha=25.0, learn_tau=True, tau_bound=(2.0, 4.0))
g = torch.Generator().manual_seed(0)
for B in (2048, 4096, 8192, 16384, 65536):
    params = torch.randn((B, eng.num_params), generator=g).cuda(); params[:, 0] = torch.rand(B, generator=g).cuda() * 2 + 2
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    out = tuple(torch.empty((B, 200, 7), device="cuda") for _ in range(2))
    for E in (-1,):
        for pf in ((0,) if E != -1 else (-1, 0)):
            _lib.reset_options(); _lib.set_option("phase_chunk", E); _lib.set_option("phase_flat", pf)
            t = graph_time(lambda: eng.trajectory(params, ip, iv, 0.0, out=out))
            print(f"| {B} | chunk {E} flat {pf} | `{eng.last_kernel()}` | {t*1e6:.1f} | {B*11424/t/8e12*100:.1f} % |", flush=True)
