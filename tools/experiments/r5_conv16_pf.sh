#!/bin/bash
# round 5: fragment prefetch distance of the small-grid kernels (dependent 40-cycle MFMA chain vs LDS read latency), one box
for L in "" ab_libs/libeinx_pf6.so ab_libs/libeinx_pf9.so ab_libs/libeinx_pf17.so; do for a in 0 2; do
  echo "== lib=${L:-tree(PF3)} EINX_CONV16_AHEAD=$a"
  EINX_LIB=$L EINX_CONV16_AHEAD=$a python tools/profile_b.py 1 2>&1 | grep -E "conv16_kernel|sum of|image.bb[2-7]|image.det0" | awk '{printf "%s %s | ", $1, $(NF-3)}'; echo
  EINX_LIB=$L EINX_CONV16_AHEAD=$a python tools/latency_graph.py 2>&1 | grep sp_mnn
done; done
EINX_LIB=ab_libs/libeinx_pf17.so EINX_CONV16_AHEAD=0 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "conv16" 2>&1 | tail -1
EINX_LIB=ab_libs/libeinx_pf17.so EINX_CONV16_AHEAD=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "conv16" 2>&1 | tail -1
