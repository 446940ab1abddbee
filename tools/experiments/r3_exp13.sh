#!/bin/bash
# round 3, experiment 13: GEMM tile engine with two K-steps per ds_read_b64 (K-permuted LDS image, same MFMA order) vs ds_read_b32
cd $GRAFT_REPO_ROOT
EINX_LIB=ab_libs/libeinx_gb64.so python -m pytest tests/test_gpu_parity.py tests/test_r2_gpu.py -q -m gpu -x -k "mnn or lightglue or lg or linear or train or matcher" 2>&1 | tail -2
for v in cur gb64 cur gb64; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  echo -n "$v: "; EINX_LIB=$L python tools/lg_bench.py --reps 10 2>/dev/null | tail -12 | tr '\n' '|'; echo
done
