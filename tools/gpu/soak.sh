cd $GRAFT_REPO_ROOT
(time MPK_FUZZ_CASES_RC=20000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "closed_loop_step_through_the_ring" 2>&1 | tail -3) > gpurun_out/soak_rc.log 2>&1
(time MPK_FUZZ_CASES_R4=10000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "ring_kernel" 2>&1 | tail -3) > gpurun_out/soak_r4.log 2>&1
(time MPK_FUZZ_CASES=10000 MPK_FUZZ_CASES_R3=8000 MPK_FUZZ_CASES_RC=1 MPK_FUZZ_CASES_R4=1 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3) > gpurun_out/soak_rest.log 2>&1
cat gpurun_out/soak_rc.log gpurun_out/soak_r4.log gpurun_out/soak_rest.log
