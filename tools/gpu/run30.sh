cd $GRAFT_REPO_ROOT
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_phase.py 65536 cfg5tau 2>&1 | grep -v amdgpu | tail -16
