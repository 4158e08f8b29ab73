import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch, bench
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
def timed(fn, n=20, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
B = 262144
params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
cp, cv = ip.double().contiguous(), iv.double().contiguous()
out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out), n=40)
V = [dict(flat=1), dict(ring=1), dict(ring=1, ring_dbg=64), dict(ring=1, ring_dbg=8), dict(ring=1, ring_dbg=72), dict(ring=1, ring_np=9), dict(ring=1, ring_ns=3)]
print("| options | +actions us | frac | traj only us | frac |")
for rep in range(2):
    for v in V:
        _lib.reset_options()
        for k, x in v.items(): _lib.set_option(k, x)
        t3 = timed(lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)); k3 = eng.last_kernel()
        t2 = timed(lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2]))
        print(f"| {v} {k3} | {t3*1e6:.1f} | {B*8624/t3/8e12*100:.1f} % | {t2*1e6:.1f} | {B*5824/t2/8e12*100:.1f} % |", flush=True)
