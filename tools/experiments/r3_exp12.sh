#!/bin/bash
# round 3, experiment 12: 16x16x4 small-grid conv kernel: parity, per-layer time at B=1, forward latency at B=1/2/4
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_random_shapes_gpu.py tests/test_r2_gpu.py -q -m gpu -x -k "conv or random or extract or e2e or properties or handle" 2>&1 | tail -2
for cfg in ${CFGS:-"4096 256" "8192 256" "8192 512" "16384 384" "0 256"}; do
  set -- $cfg
  export EINX_CONV16_MAX_TILES=$1 EINX_CONV16_MIN_WG=$2
  echo "EINX_CONV16_MAX_TILES=$1 (0: kernel off) EINX_CONV16_MIN_WG=$2"; python tools/profile_b.py 1 2>/dev/null | grep -E "^(event|image)\.(bb|det0|desc0)" | awk '{printf "%s %s | ", $1, $(NF-3)} END {print ""}'
  for b in 1 2 4; do echo -n "  "; python tools/latency_b1.py $b 2>/dev/null | tail -1 | sed 's/einx_extract (handle-level ABI): //'; done
done
