#!/bin/bash
# round 6: first layer recomputed inside the second layer's launch (conv1ab_kernel) against the two launches; same box, alternating
# EINX_FUSE01_CIN: comma list of first-layer channel counts the dispatcher may fuse ("" = none), experiment switch of this script's build
O=gpurun_out/r6_fuse01; mkdir -p $O
python -m pytest tests/test_conv_gpu.py -q -x -k "fused" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
grep -q "passed" $O/pytest.txt || exit 1
run() { n=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline --no-cpu-torch --no-scale-legs --steps 30 --warmup 5 > $O/$n.json 2> $O/$n.err
  python - "$O/$n.json" "$n" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"])
PY
}
run image_only EINX_FUSE01_CIN=1
run both EINX_FUSE01_CIN=1,5
run image_only2 EINX_FUSE01_CIN=1
run both2 EINX_FUSE01_CIN=1,5
run none EINX_FUSE01_CIN=0
