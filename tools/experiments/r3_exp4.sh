#!/bin/bash
# round 3, experiment 4: dense upsample A/B builds (EINX_LIB), per-kernel times from the library's HIP-event scopes
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-cur}; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  echo -n "$v: "; EINX_LIB=$L python tools/up_bench.py 2>/dev/null
done
