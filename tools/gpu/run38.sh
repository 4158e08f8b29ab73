cd $GRAFT_REPO_ROOT
python tools/gpu/nccl_world1.py 2>&1 | grep -i "world-1 ok\|error\|Traceback" | head -3
echo "rc=$?"
