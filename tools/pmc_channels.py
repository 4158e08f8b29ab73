#!/usr/bin/env python3
"""
Per-instance (L2 channel: 8 XCDs x 16 channels) view of a rocprofv3 --pmc pass: rocprofv3's CSV sums the instances of a
counter, its JSON keeps one record per instance.  For every dispatch of the kernels matching `pattern` prints min / mean /
max over the instances and the max/mean imbalance.     python tools/pmc_channels.py <pattern> <dir-with-*_results.json> ...
"""
import glob
import json
import os
import sys


def main():
    pat = sys.argv[1]
    print("| counter | kernel | dispatches | instances | mean per instance | min | max | max / mean | sum |")
    print("|---|---|---|---|---|---|---|---|---|")
    for d in sys.argv[2:]:
        for path in sorted(glob.glob(os.path.join(d, "*_results.json"))):
            root = json.load(open(path))["rocprofiler-sdk-tool"][0]
            names = {c["id"]["handle"]: c["name"] for c in root["counters"]}
            ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("kernel_name") for k in root["kernel_symbols"]}
            agg = {}
            for rec in root["callback_records"]["counter_collection"]:
                kname = ksym.get(rec["dispatch_data"]["dispatch_info"]["kernel_id"], "?")
                if pat not in kname:
                    continue
                per = {}
                for r in rec["records"]:
                    per.setdefault(names.get(r["counter_id"]["handle"], "?"), []).append(r["value"])
                for cn, vals in per.items():
                    agg.setdefault((cn, kname.split("(")[0][-60:]), []).append(vals)
            for (cn, kn), runs in agg.items():
                vals = runs[-1]                      # the last dispatch (warm)
                mean = sum(vals) / len(vals)
                print(f"| {cn} | `{kn}` | {len(runs)} | {len(vals)} | {mean:.0f} | {min(vals):.0f} | {max(vals):.0f} | "
                      f"{max(vals) / mean if mean else 0:.3f} | {sum(vals):.0f} |")


if __name__ == "__main__":
    main()
