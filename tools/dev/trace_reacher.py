"""Per-tile timeline of ONE wave of the rollout kernels on LongSimpleReacher (5 x 200), with and without the reward
(MPK_TRACE build: MPK_BUILD_AMALGAMATED=1 MPK_EXTRA_FLAGS=-DMPK_TRACE MPK_BUILD_OUT=ab/lib_trace.so python __graft_entry__.py --force;
 MPK_LIB=ab/lib_trace.so python tools/dev/trace_reacher.py [B] [key=value ...]).  Stamps: 1 unit start, 10 + 3 rt after the staging of
 tile rt, 11 + 3 rt after its chain, 12 + 3 rt after its reward pass, 90 unit end."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for kv in sys.argv[2:]:
    k, v = kv.split("="); _lib.set_option(k, int(v))
lib = C.CDLL(_lib.LIB_PATH); lib.mpk_debug_trace.argtypes = [C.c_void_p, C.c_int]
D, T = 5, 200
eng = TrajectoryEngine(device=0, mp_type="promp", phase_type="linear", basis_type="zero_rbf", num_dof=5, num_basis=5,
                       num_basis_zero_start=1, dt=0.01, duration=2.0, tau=2.0)
g = torch.Generator().manual_seed(0)
params = torch.randn((B, eng.num_params), generator=g).cuda(); ip = (torch.rand((B, D), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, D), device="cuda")
pos, vel = eng.trajectory(params, ip, iv, 0.0)
spec = RolloutSpec("motor", D, 0.6, 0.075, -1000.0, 1000.0, plant="double_integrator", dt=0.01)
q, qd = ip.double().contiguous(), iv.double().contiguous()
act = torch.empty((B, T, D), device="cuda"); rew = torch.empty((B, T), dtype=torch.float64, device="cuda")
goal = (torch.rand((B, 2), generator=g, dtype=torch.float64) * 4 - 2).cuda()
buf = np.zeros(512, np.int64)
for name, fn in (("no reward", lambda: eng.pd_rollout(spec, pos, vel, q, qd, out=act)),
                 ("reward", lambda: eng.reacher_rollout(spec, pos, vel, q, qd, goal, out=(act, rew)))):
    for _ in range(300): fn()
    torch.cuda.synchronize(); lib.mpk_debug_trace(buf.ctypes.data, 256)
    for rep in range(3):
        fn(); torch.cuda.synchronize()
        n = lib.mpk_debug_trace(buf.ctypes.data, 256); st = buf[:2 * n].reshape(n, 2)
        print(f"--- {name} B={B}: {n} stamps, total {st[-1, 1] - st[0, 1]} ticks")
        if rep == 2:
            prev = st[0, 1]
            for tag, c in st:
                print(f"  tag {tag:3d}  +{c - prev:7d}  (t = {c - st[0, 1]:7d})"); prev = c
