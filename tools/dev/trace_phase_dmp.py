"""Per-tile timeline of ONE wave of k_traj_phase<dmp> (cfg3 + learned tau), forcing rows interpolated / exact ("phase_table" 1 / 0).
MPK_TRACE build: MPK_BUILD_AMALGAMATED=1 MPK_EXTRA_FLAGS=-DMPK_TRACE MPK_BUILD_OUT=ab/lib_trace.so python __graft_entry__.py --force
    MPK_LIB=ab/lib_trace.so python tools/dev/trace_phase_dmp.py [B]
stamps: 10 + 5 tile: tile start, + 1 rows built, + 2 forcing contracted, + 3 Euler steps done, + 4 stored
    ... trace_phase_dmp.py B wg: k_traj_phase_dmp_wg -- 0 entry, 1 tables staged, 2 chunk inputs in, 10 + 4 block: rows + forcing built, + 1 Euler
    starts, + 2 Euler done, + 3 stored
one-unit trace build: MPK_TRACE_UNIT=mpk_traj_phase.hip MPK_BUILD_OUT=ab/lib_trace_phase.so python __graft_entry__.py --force"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from fancy_gym_amd import TrajectoryEngine, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lib = C.CDLL(_lib.LIB_PATH); lib.mpk_debug_trace.argtypes = [C.c_void_p, C.c_int]
eng = TrajectoryEngine(device=0, mp_type="dmp", phase_type="exp", basis_type="rbf", num_dof=7, num_basis=5, dt=0.02, duration=4.0, tau=4.0,
                       alpha_phase=2.0, dmp_alpha=25.0, learn_tau=True, tau_bound=(2.0, 4.0))
g = torch.Generator().manual_seed(0)
params = torch.randn((B, eng.num_params), generator=g).cuda()
params[:, 0] = torch.rand(B, generator=g).cuda() * 2 + 2
ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
buf = np.zeros(512, np.int64)
WG = "wg" in sys.argv
for mode in (1, 0):
    _lib.set_option("phase_table", mode); _lib.set_option("phase_flat", 1 if WG else 0)
    for _ in range(50): eng.trajectory(params, ip, iv, 0.0)
    torch.cuda.synchronize(); lib.mpk_debug_trace(buf.ctypes.data, 256)
    eng.trajectory(params, ip, iv, 0.0); torch.cuda.synchronize()
    n = lib.mpk_debug_trace(buf.ctypes.data, 256); st = buf[:2 * n].reshape(n, 2)
    print(f"--- {eng.last_kernel()} phase_table={mode} B={B}: {n} stamps")
    prev = st[0, 1]
    for tag, c in st:
        print(f"  tag {tag:3d}  +{c - prev:7d}  (t = {c - st[0, 1]:7d})"); prev = c
