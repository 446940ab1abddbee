cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/latency_b1.py 1 SP_LG
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lgb1 -o p -- python3 $GRAFT_REPO_ROOT/tools/latency_b1.py 1 SP_LG > /dev/null 2>&1
