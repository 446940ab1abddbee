#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2e8; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in default xcd default xcd; do
  if [ $v = default ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python bench.py --layer-table 2>/dev/null | grep -E "image.bb1|image.bb2|image.bb5|event.bb1|total" | sed "s/^/$v: /"
  EINX_LIB=$L python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['achieved'])"
done
cd /tmp && export TMPDIR=/tmp
for v in default xcd; do
  if [ $v = default ]; then L=""; else L="$GRAFT_REPO_ROOT/ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --kernel-only > $O/pmc_$v.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/pmc_$v/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:80]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "conv_block" in k: print("$v", k[40:80], len(v), sum(v)/len(v)/1024, "MiB FETCH")
PY
done
python -m pytest tests/test_r2_gpu.py -x -q -k "launcher" 2>&1 | tail -3
