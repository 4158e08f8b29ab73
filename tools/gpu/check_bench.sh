#!/bin/bash
# quick GPU check of bench.py's code paths: plain, forced one-rank RCCL, two gloo ranks; plus the bench tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_bench; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/k20.json 2> $O/k20.err
MPK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --no-streaming > $O/k20_rccl1.json 2> $O/k20_rccl1.err
MPK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/g2.json 2> $O/g2.err
timeout 1500 python -m pytest tests/test_gpu_blackbox.py -q -k "bench or rccl" 2>&1 | tail -15 > $O/pytest.log
tail -c 1500 $O/k20.json; echo; tail -c 1200 $O/k20_rccl1.json; echo; tail -c 600 $O/g2.json; echo; cat $O/pytest.log
