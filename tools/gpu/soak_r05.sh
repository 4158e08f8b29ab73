# round 5: the whole fuzz file on the final library, fresh seeds (round 4's soaks ended at 700 000 / 40 000 / 20 000)
cd $GRAFT_REPO_ROOT
(time MPK_FUZZ_START=900000 MPK_FUZZ_CASES=15000 MPK_FUZZ_START_R3=900000 MPK_FUZZ_CASES_R3=8000 MPK_FUZZ_START_R3B=900000 MPK_FUZZ_CASES_R3B=6000 \
      MPK_FUZZ_START_R4=60000 MPK_FUZZ_CASES_R4=6000 MPK_FUZZ_START_RC=60000 MPK_FUZZ_CASES_RC=10000 MPK_FUZZ_START_BB=10000 MPK_FUZZ_CASES_BB=3000 \
      MPK_FUZZ_START_GATE=10000 MPK_FUZZ_CASES_GATE=2000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -n 4 2>&1 | grep -v "NCCL WARN" | tail -4) > gpurun_out/soak_r05.log 2>&1
cat gpurun_out/soak_r05.log
