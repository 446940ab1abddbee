#!/bin/bash
# round-2 experiment 1: baseline tests + conv wave-layout variants (layer tables)
set -o pipefail
O=gpurun_out/r2e1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
for e in 0 3 4 8 12; do
  EINX_CONV_EXP=$e timeout -k 10 300 python bench.py --layer-table > $O/layers_exp$e.txt 2>&1 || exit 1
  echo "exp $e: $(tail -1 $O/layers_exp$e.txt)" | tee -a $O/summary.txt
done
timeout -k 10 300 python bench.py > $O/bench_default.json 2>$O/bench_default.err || exit 1
tail -c 600 $O/bench_default.json
