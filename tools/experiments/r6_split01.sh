#!/bin/bash
# round 6: first two backbone layers in half-batches, the thin first layer of half 1 beside the compute-bound second layer of half 0
O=gpurun_out/r6_split01; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline --no-cpu-torch --no-scale-legs --steps 30 --warmup 5 > $O/$n.json 2> $O/$n.err
  python - "$O/$n.json" "$n" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"])
PY
}
run base EINX_SPLIT01=0
run split EINX_SPLIT01=1
run split_three7 EINX_SPLIT01=1 EINX_THREE_MIN=7
run base2 EINX_SPLIT01=0
run split2 EINX_SPLIT01=1 EINX_THREE_MIN=7
python -m pytest tests/test_e2e_gpu.py -q -x -k "baseline_batch or bench_batch" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
