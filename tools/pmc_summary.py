#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (per kernel: mean counter value per dispatch + mean duration)."""
import collections
import csv
import glob
import os
import sys


def main(dirs):
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
            rows = list(csv.DictReader(open(f)))
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            dur = collections.defaultdict(dict)
            for r in rows:
                k = r["Kernel_Name"].split("(")[0][-60:]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print(f"## {f}")
            for k, cs in agg.items():
                if "k_traj" not in k and "k_pd" not in k and "k_phase" not in k and "k_episode" not in k:
                    continue
                ds = list(dur[k].values())
                print(f"  {k}: {len(ds)} dispatches, mean {sum(ds) / len(ds) / 1e3:.1f} us under PMC")
                for c, v in sorted(cs.items()):
                    print(f"    {c:30s} {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1:] or sorted(glob.glob("gpurun_out/pmc_*")))
