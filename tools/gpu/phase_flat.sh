#!/bin/bash
# per-episode-phase prodmp kernel: per-episode rounds vs flat rounds, chunk sizes
O=gpurun_out/r03_phase_flat; mkdir -p $O
python -m pytest tests/test_gpu_edge_cases.py -q -x -n 4 -p no:cacheprovider -k "flat_rounds or row_table or chunking" 2>&1 | tail -2
{
for opt in "" "phase_flat=0" "phase_flat=1"; do
  echo; echo "### T = 100, options: ${opt:-auto}"; python tools/phase_bench.py prodmp 2048 4096 8192 16384 32768 65536 262144 $opt 2>/dev/null | grep "cfg2'\|^| config\|^|---"
done
for opt in "" "phase_flat=0 phase_chunk=1" "phase_flat=0 phase_chunk=2" "phase_flat=0 phase_chunk=4" "phase_flat=1 phase_chunk=2" "phase_flat=1 phase_chunk=4" "phase_flat=1 phase_chunk=7"; do
  echo; echo "### horizons, options: ${opt:-auto}"; python tools/phase_bench.py horizons 4096 16384 65536 $opt 2>/dev/null | grep "cfg2'\|^| config\|^|---"
done
} > $O/phase_flat.md 2>&1
cat $O/phase_flat.md
