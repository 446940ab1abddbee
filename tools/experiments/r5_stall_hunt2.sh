#!/bin/bash
O=gpurun_out/r5_stall; mkdir -p $O
for v in base threads1 passive plainreload blasonly base threads1 plainreload blasonly; do
  timeout -k 10 200 python tools/experiments/r5_stall_hunt.py $v 2>$O/err2_$v.txt | grep -v "allocator before" | tee -a $O/summary2.txt
done
python -c "import numpy; numpy.show_config()" 2>&1 | grep -i -A3 "blas\|lapack" | head -30
python -c "import torch; print(torch.__config__.parallel_info())" | head -20
