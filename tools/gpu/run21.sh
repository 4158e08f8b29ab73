cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_phase.py 65536 phase_mm=0 2>&1 | grep -v amdgpu | tail -50 > $O/trace_phase_chain.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_phase.py 4096 phase_mm=0 2>&1 | grep -v amdgpu | tail -50 > $O/trace_phase_chain_4096.txt
paste $O/trace_phase_chain.txt $O/trace_phase_chain_4096.txt
