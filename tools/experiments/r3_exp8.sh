#!/bin/bash
# round 3, experiment 8: ffn.0 + LayerNorm + GELU in one launch (EINX_LG_FUSE_LN=1) vs the shipped two-launch form
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_r2_gpu.py -q -m gpu -x -k "lightglue or lg or train or matcher" 2>&1 | tail -3
for v in fused split fused split; do
  if [ $v = fused ]; then export EINX_LG_FUSE_LN=1; else unset EINX_LG_FUSE_LN; fi
  echo -n "$v: "; python tools/lg_bench.py --skip-linear --reps 10 2>/dev/null | tail -3 | tr '\n' ' '; echo
done
