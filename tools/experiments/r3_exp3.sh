#!/bin/bash
# round 3, experiment 3: the two-kernel dense upsample (den + store): parity, timing, rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT
set -e
python -m pytest tests/test_gpu_parity.py tests/test_r2_gpu.py -x -q -m gpu -k "upsample or dense or e2e or extract" > gpurun_out/r3e3_tests.log 2>&1 || { tail -30 gpurun_out/r3e3_tests.log; exit 1; }
tail -2 gpurun_out/r3e3_tests.log
python tools/up_bench.py | tee gpurun_out/r3e3_up.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3e3_prof -o up -- python3 $R/tools/up_bench.py > $R/gpurun_out/r3e3_prof.log 2>&1
cd $R
python bench.py --dense --log-assignment --no-cpu-baseline --no-extras --no-scale-legs --steps 10 | tee gpurun_out/r3e3_bench_dense.json
