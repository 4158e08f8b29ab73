#!/usr/bin/env python3
"""
Basic blocks of one kernel instantiation with their instruction mix (build container; no GPU): how many instructions a loop body
issues is what bounds the per-episode-phase kernels, and it is cheaper to read than to measure.
    python tools/dev/isa_blocks.py fancy_gym_amd/csrc/mpk_phase_fused.hip 'k_phase_fused<2, 2, true, 7, 3>' [min_instructions]
Prints: registers (VGPR / SGPR, spills, scratch) and every basic block with at least `min_instructions` (default 20) instructions:
label, count, {valu, valu64 (float64: half rate), salu, ds, vmem, lane (v_readlane / v_writelane: spilled scalars), mfma}.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    src, want = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    out = "/tmp/isa_blocks.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", '-DMPK_SOURCE_HASH="x"',
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "fancy_gym_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", out] + \
        [a for a in sys.argv[4:] if a.startswith("-D")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    syms = re.findall(r"^(_Z\S+):\s*; @", s, re.M)
    names = {sym: subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip() for sym in syms}
    hit = [sym for sym, n in names.items() if want in n]
    if not hit:
        print("no kernel matches; have:\n  " + "\n  ".join(sorted(set(n.split("(")[0] for n in names.values()))))
        return
    for sym in hit:
        meta = re.search(re.escape(sym) + r"\n.*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
                         r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s[s.index(".amdgpu_metadata"):], re.S)
        print(names[sym].split("(")[0], "-- scratch %s B, SGPR %s (spilled %s), VGPR %s (spilled %s)" % meta.groups() if meta else "")
        i = s.index("\n" + sym + ":")
        body = s[i:s.index(".Lfunc_end", i)].split("\n")
        cur, cnt, mix, total = "entry", 0, collections.Counter(), 0
        rows = []
        for line in body:
            t = line.strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                rows.append((cur, cnt, mix)); cur, cnt, mix = t.split(":")[0], 0, collections.Counter()
            elif t and not t.startswith((".", ";", "//")) and not t.endswith(":"):
                op = t.split()[0]
                key = ("mfma" if "mfma" in op else "valu64" if "_f64" in op else "ds" if op.startswith("ds_") else
                       "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                       "lane" if "readlane" in op or "writelane" in op else "salu" if op.startswith("s_") else "valu")
                mix[key] += 1; cnt += 1; total += 1
        rows.append((cur, cnt, mix))
        for label, n, m in rows:
            if n >= min_n:
                print(f"  {label:12s} {n:5d}  " + "  ".join(f"{k} {v}" for k, v in sorted(m.items())))
        print(f"  total {total} instructions in {len(rows)} blocks")


if __name__ == "__main__":
    main()
