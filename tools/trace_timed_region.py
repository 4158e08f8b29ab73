#!/usr/bin/env python3
"""
Split a rocprofv3 kernel trace of `bench.py` into its phases.  The bench launches the headline kernel W times eagerly
(warm-up), replays the captured graph of K launches untimed until the GPU has been busy ~30 ms, replays it once between
barrier + synchronize (the wall-clock-timed region: `value`, `ms_per_step`) and once more between two HIP events
(`roofline.kernel_avg_us`); the streaming row and everything else come afterwards under other kernel names.  `--stats`
averages all launches of a kernel together.
    python tools/trace_timed_region.py <..._kernel_trace.csv> [K=2000] [W=200] [kernel-name substring = k_traj_tiles]
"""
import csv
import sys

import numpy as np


def main():
    path = sys.argv[1]
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    name = sys.argv[4] if len(sys.argv) > 4 else "k_traj_tiles"
    rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
    st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
    o = np.argsort(st); st, en = st[o], en[o]
    dur = en - st
    n = len(dur)
    n_untimed = (n - W) // K - 2
    print(f"{n} launches of {rows[0]['Kernel_Name'].split('(')[0]} = {W} eager + {n_untimed} untimed + 1 wall-clock-timed + "
          f"1 event-timed replay(s) of {K}")
    print(f"  eager warm-up ({W}):                  mean duration {dur[:W].mean() / 1e3:.2f} us")
    for i in range(n_untimed):
        s = slice(W + i * K, W + (i + 1) * K)
        print(f"  untimed graph replay {i + 1} ({K}):        mean duration {dur[s].mean() / 1e3:.2f} us")
    for label, i in (("wall-clock-timed replay", n_untimed), ("event-timed replay     ", n_untimed + 1)):
        s = slice(W + i * K, W + (i + 1) * K)
        gap = st[s][1:] - en[s][:-1]
        print(f"  {label} ({K}): mean duration {dur[s].mean() / 1e3:.2f} us, mean gap to the next launch "
              f"{gap.mean() / 1e3:.2f} us, span / launch {(en[s][-1] - st[s][0]) / K / 1e3:.2f} us")
    print(f"  all launches (what --stats reports):  mean duration {dur.mean() / 1e3:.2f} us")


if __name__ == "__main__":
    main()
