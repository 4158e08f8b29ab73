import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import TrajectoryEngine, _lib
from closed_bench import CFG2, graph_time
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(device=0, **CFG2)
A = {"ring": 1, "ablations": 1}
for B in (65536, 262144):
    params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(2))
    fn = lambda: eng.trajectory(params, ip, iv, 0.0, out=out)
    for name, opts in (("ring", dict(A)), ("neither", dict(A, ring_dbg=3)), ("neither, no loads", dict(A, ring_dbg=11)), ("no stores, no loads", dict(A, ring_dbg=10)),
                       ("no loads", dict(A, ring_dbg=8)), ("no production, no loads", dict(A, ring_dbg=9)),
                       ("static neither no loads", dict(A, ring_dbg=15)), ("ring np4", dict(A, ring_np=4)), ("ring np12", dict(A, ring_np=12)), ("np12 neither", dict(A, ring_np=12, ring_dbg=3)),
                       ("ns1", dict(A, ring_ns=1)), ("ns1 neither", dict(A, ring_ns=1, ring_dbg=3)), ("ns4 neither", dict(A, ring_ns=4, ring_dbg=3))):
        _lib.reset_options()
        for k, v in opts.items(): _lib.set_option(k, v)
        t = graph_time(fn, reps=10, rounds=5)
        print(f"| {B} | {name} | {t * 1e6:.1f} |", flush=True)
    del out
