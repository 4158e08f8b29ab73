#!/usr/bin/env python3
"""
Summary of tools/gpu/episode_pmc.sh: per (kernel, batch) the counters of the two rocprofv3 --pmc passes (summed over their instances,
averaged over the dispatches behind the first 3) and what they say about the bound:
    issue share   = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES per SIMD x SIMDs)       -- cycles a SIMD spends issuing vector instructions
    wait share    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                                  -- wave-cycles spent waiting for an instruction's operands
    valu / wave   = SQ_INSTS_VALU / SQ_WAVES
    python tools/episode_pmc_summary.py gpurun_out/r06_episode_pmc
"""
import glob
import json
import os
import sys


def load(d, skip=3):
    out = {}
    for path in sorted(glob.glob(os.path.join(d, "*_results.json"))):
        root = json.load(open(path))["rocprofiler-sdk-tool"][0]
        names = {c["id"]["handle"]: c["name"] for c in root["counters"]}
        ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("kernel_name") for k in root["kernel_symbols"]}
        recs = sorted(root["callback_records"]["counter_collection"], key=lambda r: r["dispatch_data"]["start_timestamp"])
        per = {}
        for rec in recs:
            kname = ksym.get(rec["dispatch_data"]["dispatch_info"]["kernel_id"], "")
            if "mpk::" not in kname or "k_build_shared" in kname or "k_traj_phase" in kname or "k_traj_tiles" in kname or "k_traj_stream" in kname:
                continue
            sums = {}
            for r in rec["records"]:
                cn = names.get(r["counter_id"]["handle"], "?")
                sums[cn] = sums.get(cn, 0.0) + r["value"]
            dd = rec["dispatch_data"]
            per.setdefault(kname.split("(")[0].replace("void mpk::", ""), []).append((sums, (dd["end_timestamp"] - dd["start_timestamp"]) / 1e3))
        for k, runs in per.items():
            runs = runs[skip:] or runs
            o = out.setdefault(k, {"us": []})
            o["us"].extend(r[1] for r in runs)
            for cn in runs[0][0]:
                o[cn] = sum(r[0].get(cn, 0.0) for r in runs) / len(runs)
    return out


def main():
    root = sys.argv[1]
    rows = []
    for which in ("lean", "lean_rw", "roll", "roll_rw", "tt_lean", "tt_step"):
        for B in (4096, 65536):
            merged = {}
            for p in (1, 2):
                for k, v in load(os.path.join(root, f"{which}_{B}_p{p}")).items():
                    m = merged.setdefault(k, {"us": []})
                    m["us"].extend(v.pop("us"))
                    m.update(v)
            for k, v in merged.items():
                rows.append((which, B, k, v))
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
            "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]
    print("| launch | B | kernel | us under PMC | " + " | ".join(cols) + " | VALU / wave | wait share of wave-cycles | VALU-issue share of busy SIMD-cycles |")
    print("|---|---|---|---|" + "---|" * (len(cols) + 3))
    for which, B, k, v in rows:
        us = sum(v["us"]) / max(len(v["us"]), 1)
        waves = v.get("SQ_WAVES", float("nan"))
        g = lambda c: v.get(c, float("nan"))      # noqa: E731
        per_wave = g("SQ_INSTS_VALU") / waves if waves else float("nan")
        wait = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else float("nan")
        # SQ_BUSY_CYCLES is summed over the 32 shader engines x ... instances; the share is formed against SQ_ACTIVE_INST_ANY instead where it is 0
        issue = g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES") if g("SQ_BUSY_CYCLES") else float("nan")
        print(f"| {which} | {B} | `{k[:60]}` | {us:.1f} | " + " | ".join(f"{g(c):.4g}" for c in cols) + f" | {per_wave:.0f} | {wait:.3f} | {issue:.3f} |")


if __name__ == "__main__":
    main()
