#!/usr/bin/env python3
"""Where the wall clock of bench.py's timed region goes at K = 20 (a 0.19 ms graph): graph launch call, completion wait.
Variants of the wait: torch.cuda.synchronize() alone; spin on event.query() first; stream.synchronize()."""
import os
import sys
import time

import ctypes as C
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 4096
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 0))
cp, cv = ip.double().contiguous(), iv.double().contiguous()
out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
lib = _lib.load()
h, rcfg = eng._h, C.byref(spec.c)


def step(sp):
    lib.mpk_trajectory_actions(h, params.data_ptr(), ip.data_ptr(), iv.data_ptr(), 0.0, rcfg, cp.data_ptr(), cv.data_ptr(),
                               out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), B, sp)


stream = torch.cuda.current_stream()
for _ in range(5):
    step(stream.cuda_stream)
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(stream)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for _ in range(K):
            step(side.cuda_stream)
stream.wait_stream(side)
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.05:
    g.replay(); torch.cuda.synchronize()
ev = torch.cuda.Event()


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    if mode == "spin":
        ev.record(stream)
        while not ev.query():
            pass
    elif mode == "stream":
        stream.synchronize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t0) * 1e6


for mode in ("sync", "spin", "stream", "sync", "spin", "stream"):
    rs = [run(mode) for _ in range(200)]
    rs.sort(key=lambda r: r[1])
    med = rs[len(rs) // 2]
    print(f"K={K} wait={mode:6s}: replay() call {med[0]:6.1f} us, region {med[1]:7.1f} us = {med[1] / K:6.3f} us per step; best region {rs[0][1]:7.1f}")
