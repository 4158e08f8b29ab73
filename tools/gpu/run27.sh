cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02y
python tools/stream_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02y/stream_probe.md
