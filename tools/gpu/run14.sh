cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
rocm-smi --showmemorypartition --showcomputepartition --showmeminfo vram 2>&1 | grep -v "^$" > $O/smi_partition.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -50 > $O/trace_pipe_4096_full.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 8192 full pipe=1 2>&1 | grep -v amdgpu | tail -50 > $O/trace_pipe_8192_full.txt
python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "lib\|auto\|pipe=1" > $O/closed_base.md
MPK_LIB=$PWD/fancy_gym_amd/libmpk_prio.so python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "lib\|auto\|pipe=1" > $O/closed_prio.md
timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5 > $O/fuzz.log
timeout 600 python -m pytest tests/test_gpu_blackbox.py -m gpu -x -q -k "replan_step_equals" 2>&1 | tail -3 >> $O/fuzz.log
( cd /tmp && export TMPDIR=/tmp
for pass in "tlb1:TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "tlb2:TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$name -o b262144 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu --eager --no-streaming --batch 262144 > /dev/null 2>&1
done )
python tools/pmc_summary.py $O/pmc_tlb1 $O/pmc_tlb2 > $O/pmc_tlb.txt 2>&1
python bench.py --no-cpu --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench.json
cat $O/trace_pipe_4096_full.txt $O/closed_base.md $O/closed_prio.md $O/fuzz.log $O/pmc_tlb.txt $O/smi_partition.txt
