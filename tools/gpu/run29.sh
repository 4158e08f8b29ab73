cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02soak
MPK_FUZZ_CASES=200000 timeout 2600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8 | tee gpurun_out/r02soak/fuzz.log
