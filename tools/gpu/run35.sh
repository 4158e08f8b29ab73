cd $GRAFT_REPO_ROOT
( time python bench.py > /tmp/b.json 2>/tmp/b.err ) 2>&1 | tail -3
tail -c 400 /tmp/b.json; echo
( time python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/b2.json 2>/tmp/b2.err ) 2>&1 | tail -3
