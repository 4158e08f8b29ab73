"""What a plain fill reaches at the output sizes of the launcher's crossovers (torch.fill_ of two / three arrays of B x 100 x 7 floats,
captured graph of 10, GPU-busy warm-up): the yardstick for the 0.3 - 1 GB "dip" of the open-loop kernels.  python tools/fill_size_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from closed_bench import graph_time
torch.cuda.set_device(0)
print("| episodes (cfg2 shape) | arrays | MB written | fill us | TB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for B in (12288, 24576, 32768, 49152, 65536, 98304, 131072, 262144):
    for narr in (2, 3):
        xs = [torch.empty((B, 100, 7), device="cuda") for _ in range(narr)]
        def fn():
            for x in xs: x.fill_(1.0)
        t = graph_time(fn, reps=10, rounds=5)
        nbytes = narr * B * 2800
        print(f"| {B} | {narr} | {nbytes / 1e6:.0f} | {t * 1e6:.1f} | {nbytes / t / 1e12:.2f} | {nbytes / t / 8e12 * 100:.1f} % |", flush=True)
        del xs
