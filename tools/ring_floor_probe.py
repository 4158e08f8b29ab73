"""Where the fixed cost of k_traj_ring goes (round 5): the open-loop launch forced onto the ring at sizes below its crossover, with the
ablation bits of "ring_dbg" (1 no production, 2 no stores; "ablations" 1) and the launch-geometry options.
    python tools/ring_floor_probe.py [B ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import TrajectoryEngine, _lib
from closed_bench import CFG2, graph_time
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(device=0, **CFG2)
print("| B | variant | kernel | us |")
print("|---|---|---|---|")
A = {"ring": 1, "ablations": 1}
for B in [int(a) for a in sys.argv[1:]] or [3072, 12288, 24576, 65536]:
    params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(2))
    fn = lambda: eng.trajectory(params, ip, iv, 0.0, out=out)
    for name, opts in (("tiles", {"mapping": 1}), ("flat", {"flat": 1}), ("ring", dict(A)), ("ring, no production", dict(A, ring_dbg=1)),
                       ("ring, no stores", dict(A, ring_dbg=2)), ("ring, neither", dict(A, ring_dbg=3)),
                       ("ring, static batches", dict(A, ring_dbg=4)), ("ring, static, neither", dict(A, ring_dbg=7)),
                       ("ring, tickets of 1 batch", dict(A, ring_tb=1)), ("ring, tickets of 2", dict(A, ring_tb=2)), ("ring, tickets of 5 (round 4)", dict(A, ring_tb=5)),
                       ("ring, 1 group per batch", dict(A, ring_m=1)), ("ring, 2 groups per batch", dict(A, ring_m=2)),
                       ("ring, 4 producers", dict(A, ring_np=4)), ("ring, write-through", dict(A, write_through=1)),
                       ("ring, plain stores", dict(A, write_through=0))):
        _lib.reset_options()
        for k, v in opts.items(): _lib.set_option(k, v)
        t = graph_time(fn, reps=10, rounds=5)
        print(f"| {B} | {name} | `{eng.last_kernel()}` | {t * 1e6:.1f} |", flush=True)
    _lib.reset_options()
    del out
