cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z
rocm-smi --showtemp --json 2>/dev/null | head -c 1500; echo
python bench.py --no-cpu --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['roofline_streaming']))"
rocm-smi --showtemp 2>/dev/null | grep -i "temp" | head
