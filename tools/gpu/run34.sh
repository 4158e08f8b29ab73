cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02sp
hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe 2>/dev/null
/tmp/store_probe 262144 2>&1 | tail -20 | tee gpurun_out/r02sp/store_probe_262144.txt
/tmp/store_probe 1048576 2>&1 | tail -20 | tee gpurun_out/r02sp/store_probe_1m.txt
python bench.py --no-cpu --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline_streaming']; print('streaming row', s['kernel_avg_us'], s['frac'], s['box_fill_GBps'])"
