#!/bin/bash
# round 3, experiment 11: does the ablation without per-chunk global loads (abl5) run faster because the matrix cores are
# busier, or because the chip clocks higher on its (chunk-invariant) operands?  MFMA-busy + effective clock of both builds
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in cur abl5; do
  if [ $v = cur ]; then export EINX_LIB=""; else export EINX_LIB=$R/ab_libs/libeinx_$v.so; fi
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_abl_$v/p1 -o p -- python3 $R/bench.py --layer-table --no-cpu-baseline > $R/gpurun_out/pmc_abl_$v.log 2>&1
  echo "$v done"
done
