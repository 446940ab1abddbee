#!/bin/bash
# round 6: conv1a recomputed inside conv1b's launch (conv1ab_kernel) against the two launches; same box, alternating
O=gpurun_out/r6_fuse01; mkdir -p $O
python -m pytest tests/test_conv_gpu.py -q -x -k "fused" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
grep -q "passed" $O/pytest.txt || exit 1
run() { n=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline --no-cpu-torch --no-scale-legs --steps 30 --warmup 5 > $O/$n.json 2> $O/$n.err
  python - "$O/$n.json" "$n" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["launch_ms"])
PY
}
run two EINX_FUSE01=0
run fused EINX_FUSE01=1
run two2 EINX_FUSE01=0
run fused2 EINX_FUSE01=1
