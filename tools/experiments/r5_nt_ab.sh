#!/bin/bash
# round 5: non-temporal stores for >= 512 MiB un-pooled conv outputs: parity, then SP / SiLK benches of the previous build and the tree on one box
set -o pipefail
O=gpurun_out/r5_nt; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv" > $O/pytest_conv.txt 2>&1 || { tail -30 $O/pytest_conv.txt; exit 1; }
tail -2 $O/pytest_conv.txt
for rep in 1 2; do
  for tag in vf nt; do
    for cfg in sp_mnn silk_mnn; do
      EINX_LIB=ab_libs/libeinx_$tag.so timeout -k 10 400 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-scale-legs --steps 12 > $O/${cfg}_${tag}_$rep.json 2> $O/${cfg}_${tag}_$rep.err || { tail -20 $O/${cfg}_${tag}_$rep.err; exit 1; }
      echo "$tag $cfg run $rep: $(python -c "import json,sys; d=json.loads(open('$O/${cfg}_${tag}_$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
    done
  done
done
