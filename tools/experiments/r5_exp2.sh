#!/bin/bash
# round 5: occupancy variants and forced tiles of the new conv kernel, one box; then the headline of the working tree
set -o pipefail
O=gpurun_out/r5_exp2; mkdir -p $O
run() {  # tag, lib, tile
  EINX_LIB=$2 EINX_CONV_TILE=$3 timeout -k 10 300 python bench.py --layer-table > $O/layers_$1.txt 2>&1 || { tail -20 $O/layers_$1.txt; exit 1; }
  echo "== $1: $(tail -1 $O/layers_$1.txt)"
}
run tree "" ""
run wps6 ab_libs/libeinx_wps6.so ""
run wps4_4 ab_libs/libeinx_wps4_4.so ""
for t in 0 1 2 3 4; do run tile$t "" $t; done
run tree2 "" ""
timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --steps 40 > $O/bench_tree.json 2> $O/bench_tree.err || { tail -20 $O/bench_tree.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_exp2/bench_tree.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
PY
cd $O && for f in layers_*.txt; do echo "$f $(grep -E '^(event|image)\.' $f | awk '{printf "%s ", $(NF-3)}')"; done
