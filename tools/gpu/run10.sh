cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert|FAILED" | tail -15 | tee gpurun_out/r02j/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|Power|Temp" | head -8 > gpurun_out/r02j/smi_before.txt
python tools/sweep.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02j/sweep.md
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|Power|Temp" | head -8 > gpurun_out/r02j/smi_after.txt
cat gpurun_out/r02j/smi_before.txt gpurun_out/r02j/smi_after.txt
