cd $GRAFT_REPO_ROOT
for e in 1 2 3 4 6 9; do echo "E=$e"; python tools/phase_bench.py 4096 16384 65536 phase_chunk=$e 2>&1 | grep -v amdgpu | grep "cfg3"; done
