#!/usr/bin/env python3
"""A/B of library builds / options on the streaming step, alternating in ONE gpurun call (boxes differ by up to 35 %):
    python tools/ab_libs.py [rounds] <lib.so or ->[:key=value[,key=value]] ...       ('-' = the shipped library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 2
print("| library / options | kernel | B = 262144 +actions us | of 8 TB/s | trajectory only us | of 8 TB/s | B = 65536 +actions us | of 8 TB/s | "
      "trajectory only us | of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in range(rounds):
    for spec in args:
        lib, _, opts = spec.partition(":")
        env = dict(os.environ)
        env.pop("MPK_LIB", None)
        if lib != "-":
            env["MPK_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_case.py")] + [o for o in opts.split(",") if o],
                             env=env, capture_output=True, text=True)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("|")]
        print(lines[-1] if lines else f"| {spec} | failed: {out.stderr[-200:]} |", flush=True)
