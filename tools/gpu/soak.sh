# the whole fuzz file on the current library, fresh seeds per call (round 5 ended at 100 000 000 + 200 000; round 6 starts at 200 000 000)
#   SOAK_N=15000 (cases of the first test; the others in proportion) SOAK_START=<first seed>
cd $GRAFT_REPO_ROOT
(time MPK_FUZZ_START=${SOAK_START:-900000} MPK_FUZZ_CASES=${SOAK_N:-15000} MPK_FUZZ_START_R3=${SOAK_START:-900000} MPK_FUZZ_CASES_R3=$((${SOAK_N:-15000} / 2)) MPK_FUZZ_START_R3B=${SOAK_START:-900000} MPK_FUZZ_CASES_R3B=$((${SOAK_N:-15000} * 2 / 5)) \
      MPK_FUZZ_START_R4=${SOAK_START:-60000} MPK_FUZZ_CASES_R4=$((${SOAK_N:-15000} * 2 / 5)) MPK_FUZZ_START_RC=${SOAK_START:-60000} MPK_FUZZ_CASES_RC=$((${SOAK_N:-15000} * 2 / 3)) MPK_FUZZ_START_BB=${SOAK_START:-10000} MPK_FUZZ_CASES_BB=$((${SOAK_N:-15000} / 5)) \
      MPK_FUZZ_START_GATE=${SOAK_START:-10000} MPK_FUZZ_CASES_GATE=$((${SOAK_N:-15000} * 2 / 15)) timeout 3300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -n 4 2>&1 | grep -v "NCCL WARN" | tail -4) > gpurun_out/soak.log 2>&1
cat gpurun_out/soak.log
