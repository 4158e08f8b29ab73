cd $GRAFT_REPO_ROOT
for l in libmpk_p1 libmpk_p2; do MPK_LIB=$PWD/fancy_gym_amd/$l.so python tools/closed_bench.py 2048 4096 2>&1 | grep -v amdgpu | grep "lib\|full.*pipe=1"; done
