#!/bin/bash
# k_phase_fused: register-allocation target / workgroup size A/B (MPK_LIB=<other build>), chunk sizes alternated in one call
mkdir -p gpurun_out/r06
for lib in "" ab/lib_pf3.so; do
  for c in 4 8; do
    echo "== lib=${lib:-shipped} phase_chunk=$c"
    if [ -n "$lib" ]; then export MPK_LIB=$lib; else unset MPK_LIB; fi; python tools/learned_phase_bench.py 8192 65536 TT-ProDMP BeerPong-ProMP phase_chunk=$c 2>/dev/null | grep -v "separate launches\|gated\|trajectory |\|^lib\|^| config\|^|---"
  done
done
