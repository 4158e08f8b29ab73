#!/bin/bash
# k_phase_fused<..,act> at a few thousand episodes: tiles of a chunk on several waves (phase_split) x chunk size (profiles/r06_phase_fused_chunks.md)
for c in -1 2 4; do for sp in -1 1 2 4 8 22; do
  opts=""; [ $c -ge 0 ] && opts="$opts phase_chunk=$c"; [ $sp -ge 0 ] && opts="$opts phase_split=$sp"
  echo "== chunk $c split $sp"; python tools/learned_phase_bench.py 1024 2048 4096 8192 TT-ProDMP BeerPong-ProMP $opts 2>/dev/null | grep "actions" | cut -d'|' -f2,3,4,5,7
done; done
