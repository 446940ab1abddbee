#!/bin/bash
# round 5: upsample_den_kernel unroll / staging variants (tools/up_bench.py), one box
for L in "" ab_libs/libeinx_upd_u2.so ab_libs/libeinx_upd_u4.so ab_libs/libeinx_upd_p16.so ab_libs/libeinx_upd_p16u4.so ""; do
  echo "== ${L:-tree}: $(EINX_LIB=$L python tools/up_bench.py 2>&1 | tail -1)"
done
