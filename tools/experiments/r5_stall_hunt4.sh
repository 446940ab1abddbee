#!/bin/bash
O=gpurun_out/r5_stall; mkdir -p $O
for rep in 1 2 3; do for v in base threads1; do
  timeout -k 10 200 python tools/experiments/r5_stall_hunt.py $v 2>/dev/null | grep -v "allocator before" | tee -a $O/summary4.txt
done; done
cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat 2>/dev/null | head -12; nproc
