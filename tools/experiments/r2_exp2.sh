#!/bin/bash
# round-2 check 2: new bench.py (default line, one-rank RCCL launch through the launcher), gpu tests
set -o pipefail
O=gpurun_out/r2e2; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 3000 $O/bench_default.err
timeout -k 10 300 python bench.py --gpus 1 --spawn --no-cpu-baseline --no-extras > $O/bench_spawn1.json 2>$O/bench_spawn1.err; echo "spawn rc=$?" | tee -a $O/summary.txt
tail -c 1500 $O/bench_spawn1.err
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
