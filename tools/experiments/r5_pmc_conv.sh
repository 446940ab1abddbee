#!/bin/bash
# round 5: rocprofv3 --pmc passes over the conv layer table (one pass per counter set), summarised by tools/pmc_tiles.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_conv5; mkdir -p $R/gpurun_out/pmc_conv5
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_conv5/p$i -o p -- python3 $R/bench.py --layer-table --no-cpu-baseline > $R/gpurun_out/pmc_conv5_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_conv5_$i.log; exit 1; }
done
cd $R && python3 tools/pmc_tiles.py gpurun_out/pmc_conv5 > gpurun_out/pmc_conv5/summary.json && python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_conv5/summary.json'))
for k,v in d.items(): print(k[18:60], v.get('mean_us_per_launch_profiled'), 'busy',v.get('mfma_busy_frac'),'MHz',v.get('effective_mhz'),'valu/mfma',v.get('valu_per_mfma'),'wait_any',v.get('wait_inst_any_per_wave_cycle'))
PY
