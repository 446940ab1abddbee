#!/bin/bash
# round 5: conv16d_kernel (LDS-DMA staging, AHEAD chunks in flight) vs conv16_kernel at single pairs, one box
set -o pipefail
for a in 3 2 4; do
  EINX_CONV16_AHEAD=$a timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "conv16" 2>&1 | tail -1 || exit 1
done
for a in 0 2 3 4 0 3; do
  echo "== EINX_CONV16_AHEAD=$a"
  EINX_CONV16_AHEAD=$a python tools/profile_b.py 1 2>&1 | grep -E "conv16_kernel|sum of|image.bb[2-7]|image.det0" | awk '{printf "%s %s | ", $1, $(NF-3)}'; echo
  EINX_CONV16_AHEAD=$a python tools/latency_graph.py 2>&1 | grep sp_mnn
done
