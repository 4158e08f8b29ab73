#!/bin/bash
O=gpurun_out/r03_misc1; mkdir -p $O
{
for opt in "phase_chunk=4" "phase_chunk=8" "phase_chunk=5" "phase_chunk=4" "phase_chunk=8"; do
  echo "### dmp $opt"; python tools/phase_bench.py 4096 16384 65536 $opt 2>/dev/null | grep "cfg3'"
done
} > $O/dmp_chunk.md 2>&1
cat $O/dmp_chunk.md
python tools/flat_small.py > $O/flat_small.md 2>&1; cat $O/flat_small.md
