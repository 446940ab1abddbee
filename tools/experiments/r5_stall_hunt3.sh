#!/bin/bash
O=gpurun_out/r5_stall; mkdir -p $O
for rep in 1 2 3 4 5; do for v in base gcdisable gc threads1 sleep; do
  timeout -k 10 200 python tools/experiments/r5_stall_hunt.py $v 2>/dev/null | grep -v "allocator before" | tee -a $O/summary3.txt
done; done
