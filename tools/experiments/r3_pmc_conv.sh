#!/bin/bash
# rocprofv3 --pmc passes over the conv layers (bench.py --layer-table).  Run on the GPU box from the repo root:
#   OUT=pmc_conv3 [EINX_LIB=ab_libs/libeinx_X.so] bash tools/experiments/r3_pmc_conv.sh
set -e
R=$GRAFT_REPO_ROOT
OUT=${OUT:-pmc_conv3}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$OUT/p$i -o p -- python3 $R/bench.py --layer-table --no-cpu-baseline > $R/gpurun_out/${OUT}_$i.log 2>&1
  echo "pass $i done"
done
