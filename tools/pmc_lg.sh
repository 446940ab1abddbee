#!/bin/bash
# PMC passes over one LightGlue forward (run on the GPU box via gpurun from the repo root)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  mkdir -p $R/gpurun_out/pmc_lg; rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_lg/p$i -o p -- python3 $R/tools/lg_bench.py --skip-linear --reps 1 > $R/gpurun_out/pmc_lg_$i.log 2>&1
done
