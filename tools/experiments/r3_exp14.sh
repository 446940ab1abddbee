#!/bin/bash
# round 3, experiment 14: attention without the rescale-by-one and without the key mask on whole blocks (bit-identical)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_r2_gpu.py -q -m gpu -x -k "lightglue or lg or train or matcher" 2>&1 | tail -2
for v in cur attnold cur attnold; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  echo -n "$v: "; EINX_LIB=$L python tools/lg_bench.py --skip-linear --reps 10 2>/dev/null | tail -3 | head -1
done
