cd $GRAFT_REPO_ROOT
O=gpurun_out/r02fin; mkdir -p $O
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 300 $O/bench_k20.err
MPK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench_g2.json 2> $O/bench_g2.err
python - <<'PY'
import json
for f in ("bench_k20.json","bench_g2.json"):
    l=[x for x in open("gpurun_out/r02fin/"+f) if x.startswith("{")][-1]; d=json.loads(l)
    print(f, d["value"], d["ms_per_step"], d["n_gpus"], d["roofline"]["frac"], (d.get("roofline_streaming") or {}).get("frac"), (d.get("allgather") or {}).get("gathered_equals_shards"), (d.get("cpu_baseline") or {}).get("value"))
PY
python tools/phase_bench.py 4096 65536 2>&1 | grep -v amdgpu | grep cfg
