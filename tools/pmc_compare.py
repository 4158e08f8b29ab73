#!/usr/bin/env python3
"""
Counters of several rocprofv3 --pmc passes side by side, one column per kernel: every counter summed over its instances
and averaged over the dispatches of a kernel (the first `skip` dispatches dropped as warm-up).
    python tools/pmc_compare.py [--skip N] [--cycle SUBSTR=lab1,lab2,...] <dir> [<dir> ...]   (directories holding *_results.json)
--cycle: the dispatches of the kernel whose name contains SUBSTR take the labels lab1, lab2, ... in turn, in dispatch order (one
kernel launched under several option settings: tools/ring_pmc_driver.py).
"""
import glob
import json
import os
import sys


def short(name):
    for key, lab in (("k_traj_stream<2, 0", "stream +actions"), ("k_traj_stream<2, -1", "stream traj only"),
                     ("FillFunctor", "torch fill"), ("k_traj_tiles", "tiles"), ("k_traj_wide<0, 13>", "wide promp K=1001 T=200"),
                     ("k_traj_wide<0, 7>", "wide promp K=64 T=100"), ("k_traj_wide<2, 7>", "wide prodmp K=128 T=100"),
                     ("k_traj_wide<1, 13>", "wide dmp K=256 T=200"), ("k_pd_rollout_tiles<1", "pd_rollout_tiles<1>"), ("k_pd_rollout_tiles<4", "pd_rollout_tiles<4>"), ("k_traj_", None)):
        if key in name:
            return lab or name.split("(")[0].replace("void mpk::", "")[:40]
    return None


def main():
    args = sys.argv[1:]
    skip = 0
    cycle = {}
    while args and args[0] in ("--skip", "--cycle"):
        if args[0] == "--skip":
            skip = int(args[1])
        else:
            key, labs = args[1].split("=", 1)
            cycle[key] = labs.split(",")
        args = args[2:]
    table, cols, dur = {}, [], {}
    for d in args:
        for path in sorted(glob.glob(os.path.join(d, "*_results.json"))):
            root = json.load(open(path))["rocprofiler-sdk-tool"][0]
            names = {c["id"]["handle"]: c["name"] for c in root["counters"]}
            ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("kernel_name") for k in root["kernel_symbols"]}
            per_kernel = {}
            seen = {}
            recs = sorted(root["callback_records"]["counter_collection"], key=lambda r: r["dispatch_data"]["start_timestamp"])
            for rec in recs:
                kname = ksym.get(rec["dispatch_data"]["dispatch_info"]["kernel_id"], "")
                lab = short(kname)
                for key, labs in cycle.items():
                    if key in kname:
                        i = seen.get(key, 0)
                        seen[key] = i + 1
                        lab = labs[i % len(labs)]
                if lab is None:
                    continue
                sums = {}
                for r in rec["records"]:
                    cn = names.get(r["counter_id"]["handle"], "?")
                    sums[cn] = sums.get(cn, 0.0) + r["value"]
                dd = rec["dispatch_data"]
                per_kernel.setdefault(lab, []).append((sums, (dd["end_timestamp"] - dd["start_timestamp"]) / 1e3))
            for lab, runs in per_kernel.items():
                runs = runs[skip:] or runs
                if lab not in cols:
                    cols.append(lab)
                dur.setdefault(lab, []).extend(r[1] for r in runs)
                for cn in runs[0][0]:
                    table.setdefault(cn, {})[lab] = sum(r[0].get(cn, 0.0) for r in runs) / len(runs)
    print("| counter (sum over instances, per launch) | " + " | ".join(cols) + " |")
    print("|---|" + "---|" * len(cols))
    print("| kernel duration under the counter passes, us | " + " | ".join(f"{sum(dur[c]) / len(dur[c]):.1f}" for c in cols) + " |")
    for cn in sorted(table):
        print(f"| {cn} | " + " | ".join(f"{table[cn].get(c, float('nan')):.4g}" for c in cols) + " |")


if __name__ == "__main__":
    main()
