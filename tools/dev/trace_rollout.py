import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from closed_bench import CFG2, PG, DG
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for kv in sys.argv[2:]:
    k, v = kv.split("="); _lib.set_option(k, int(v))
lib = C.CDLL(_lib.LIB_PATH); lib.mpk_debug_trace.argtypes = [C.c_void_p, C.c_int]
eng = TrajectoryEngine(device=0, **CFG2)
g = torch.Generator().manual_seed(0)
params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
pos, vel = eng.trajectory(params, ip, iv, 0.0)
spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
q, qd = ip.double().contiguous(), iv.double().contiguous()
act = torch.empty((B, 100, 7), device="cuda")
buf = np.zeros(512, np.int64)
for _ in range(300): eng.pd_rollout(spec, pos, vel, q, qd, out=act)
torch.cuda.synchronize(); lib.mpk_debug_trace(buf.ctypes.data, 256)
for rep in range(3):
    eng.pd_rollout(spec, pos, vel, q, qd, out=act); torch.cuda.synchronize()
    n = lib.mpk_debug_trace(buf.ctypes.data, 256); st = buf[:2 * n].reshape(n, 2)
    print(f"--- B={B}: {n} stamps, total {st[-1, 1] - st[0, 1]} cycles")
    if rep == 2:
        prev = st[0, 1]
        for tag, c in st:
            print(f"  tag {tag:3d}  +{c - prev:7d}  (t = {c - st[0, 1]:7d})"); prev = c
