cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -6
for r in 1 2; do for l in libmpk_head libmpk; do echo "lib $l"; MPK_LIB=$PWD/fancy_gym_amd/$l.so python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "auto\|pipe=1" | grep -v "8192 | auto"; done; done
