cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for l in head dpp; do echo "lib $l"; MPK_LIB=$PWD/fancy_gym_amd/libmpk_$l.so python tools/phase_bench.py 65536 2>&1 | grep -v amdgpu | grep "cfg5"; done; done
