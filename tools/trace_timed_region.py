#!/usr/bin/env python3
"""
Split a rocprofv3 kernel trace of `bench.py` into its phases: eager warm-up launches, the untimed first graph replay and
the timed replay (the last K launches).  `--stats` averages all of them together; the bench's `kernel_avg_us` is the
timed replay only.     python tools/trace_timed_region.py <..._kernel_trace.csv> [K=2000] [W=200]
"""
import csv
import sys

import numpy as np


def main():
    path = sys.argv[1]
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    rows = [r for r in csv.DictReader(open(path)) if "k_traj" in r["Kernel_Name"]]
    st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
    o = np.argsort(st); st, en = st[o], en[o]
    dur = en - st
    print(f"{len(rows)} launches of {rows[0]['Kernel_Name'].split('(')[0]}")
    print(f"  eager warm-up ({W}):            mean duration {dur[:W].mean() / 1e3:.2f} us")
    print(f"  first graph replay ({K}):     mean duration {dur[W:W + K].mean() / 1e3:.2f} us")
    t = slice(len(dur) - K, len(dur))
    gap = st[t][1:] - en[t][:-1]
    print(f"  timed graph replay ({K}):     mean duration {dur[t].mean() / 1e3:.2f} us, mean gap to the next launch "
          f"{gap.mean() / 1e3:.2f} us, span / launch {(en[-1] - st[len(dur) - K]) / K / 1e3:.2f} us")
    print(f"  all launches (what --stats reports): mean duration {dur.mean() / 1e3:.2f} us")


if __name__ == "__main__":
    main()
