"""k_traj_flat at one / two / three workgroups (4 / 8 / 12 waves) per CU against the ring and the tile-major kernel, cfg2's shape, trajectory only
and + actions, alternating rounds (round 5: profiles/r05_flat_workgroups.md).   python tools/flat_wg_sweep.py traj|act B ..."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from closed_bench import CFG2, PG, DG, capture, time_rows
g = torch.Generator().manual_seed(0)
eng = TrajectoryEngine(device=0, **CFG2)
spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
mode = sys.argv[1]
print("| mode | episodes | variant | kernel | us | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for B in [int(a) for a in sys.argv[2:]]:
    params = torch.randn((B, 42), generator=g).cuda(); ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda(); iv = torch.zeros((B, 7), device="cuda")
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(3))
    fn = (lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2])) if mode == "traj" else (lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out))
    nb = 5824 if mode == "traj" else 8624
    rows = []
    for name, opts in (("auto", {}), ("flat, 3 workgroups per CU", {"flat": 1, "phase_waves": 12}), ("flat, 2", {"flat": 1}), ("flat, 1", {"flat": 1, "phase_waves": 4}),
                       ("ring", {"ring": 1}), ("tiles", {"mapping": 1})):
        _lib.reset_options()
        for k, v in opts.items(): _lib.set_option(k, v)
        rows.append((name, capture(fn, 10), eng.last_kernel()))
    ts = time_rows([r[1] for r in rows], reps=10, rounds=7)
    for (name, _, k), t in zip(rows, ts):
        print(f"| {mode} | {B} | {name} | `{k}` | {t*1e6:.1f} | {B*nb/t/8e12*100:.1f} % |", flush=True)
    del out
