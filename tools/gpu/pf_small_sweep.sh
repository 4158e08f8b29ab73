#!/bin/bash
# k_phase_fused at a few thousand episodes: chunk size x row table in LDS / L2 (profiles/r06_phase_fused_chunks.md)
for t in -1 0 1; do for c in 1 2 4 8; do
  opts="phase_chunk=$c"; [ $t -ge 0 ] && opts="$opts phase_table=$t"
  echo "== $opts"; python tools/learned_phase_bench.py 1024 2048 4096 TT-ProDMP $opts 2>/dev/null | grep "actions\|closed-loop step\|verbose < 2" | grep -v gated | cut -d'|' -f3,4,5,7
done; done
