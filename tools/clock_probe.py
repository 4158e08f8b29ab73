#!/usr/bin/env python3
"""
Why do the HBM-streaming rows move by +-15 % from box to box?  Streams the headline kernel family at B = 262144 (2.2 GB
written per launch) for a few seconds while a thread samples rocm-smi (shader / memory / fabric clocks, package power,
temperatures), and prints the achieved GB/s per half-second window next to the samples.
    python tools/clock_probe.py [seconds]
"""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fancy_gym_amd import RolloutSpec, TrajectoryEngine  # noqa: E402
from closed_bench import CFG2, DG, PG  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        t = time.perf_counter()
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True,
                               text=True, timeout=5)
            d = json.loads(r.stdout).get("card0", {})
            out.append((t, d))
        except Exception as e:  # noqa: BLE001
            out.append((t, {"error": str(e)}))
        time.sleep(0.15)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    B = 262144
    eng = TrajectoryEngine(device=0, **CFG2)
    g = torch.Generator().manual_seed(0)
    params = torch.randn((B, 42), generator=g).to(dev)
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).to(dev)
    iv = torch.zeros((B, 7), device=dev)
    spec = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    out = tuple(torch.empty((B, 100, 7), device=dev) for _ in range(3))
    fn = lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, 0.0, out=out)   # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, smi = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, smi))
    th.start()
    t0 = time.perf_counter()
    windows = []
    while time.perf_counter() - t0 < seconds:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        a.record()
        n = 40
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        windows.append((w0, a.elapsed_time(b) * 1e-3 / n))
    stop.set(); th.join()
    nbytes = 8624 * B
    print(f"kernel {eng.last_kernel()}, B = {B}, {nbytes / 1e9:.2f} GB algorithmic per launch")
    print("| t (s) | us / launch | GB/s | of 8 TB/s |")
    print("|---|---|---|---|")
    for w0, t in windows:
        print(f"| {w0 - t0:.2f} | {t * 1e6:.1f} | {nbytes / t / 1e9:.0f} | {nbytes / t / 8e12 * 100:.1f} % |")
    ts = [t for _, t in windows]
    print(f"\nlaunch time min / median / max: {min(ts) * 1e6:.1f} / {np.median(ts) * 1e6:.1f} / {max(ts) * 1e6:.1f} us "
          f"(spread {(max(ts) / min(ts) - 1) * 100:.1f} %)")
    keys = sorted({k for _, d in smi for k in d})
    print("\nrocm-smi samples during the run:")
    for k in keys:
        vals = [d[k] for _, d in smi if k in d]
        uniq = sorted(set(vals))
        print(f"  {k}: {uniq[:6]}{' ...' if len(uniq) > 6 else ''}  ({len(vals)} samples)")


if __name__ == "__main__":
    main()
