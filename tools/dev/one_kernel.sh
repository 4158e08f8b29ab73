#!/bin/bash
# Compile ONE kernel instantiation of mpk_kernels.hip in seconds and print its resource usage (the full library takes
# ~90 s: ~300 instantiations).   tools/dev/one_kernel.sh 'k_traj_split<2,3,2,true>(TrajArgs, ActArgs)' [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/mpk_one
mkdir -p "$OUT"
cat > "$OUT/one.hip" <<SRC
#define MPK_DEVICE_ONLY 1
#include "$ROOT/fancy_gym_amd/csrc/mpk_kernels.hip"
namespace mpk { template __global__ void $1; }
SRC
shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I"$ROOT/include" -I"$ROOT/fancy_gym_amd/csrc" \
  -c "$OUT/one.hip" -o "$OUT/one.o" -Rpass-analysis=kernel-resource-usage --save-temps=obj "$@" 2>&1 | \
  grep -E "VGPRs:|AGPRs|SGPRs:|ScratchSize|Occupancy|LDS Size|error" | sed "s/.*remark: [^ ]* //" | tail -6 | tr "\n" " "; echo
ls "$OUT"/*.s 2>/dev/null | head -2
