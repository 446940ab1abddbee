#!/bin/bash
O=gpurun_out/r5_stall; mkdir -p $O
for v in base noreload emptycache gc nowatch sleep base; do
  timeout -k 10 200 python tools/experiments/r5_stall_hunt.py $v 2>$O/err_$v.txt | tee -a $O/summary.txt
done
# one run with the HIP API log around the loop (level 3: API calls with timestamps)
EINX_STALL_MARK=1 AMD_LOG_LEVEL=3 timeout -k 10 300 python tools/experiments/r5_stall_hunt.py base > $O/log_base.out 2> $O/log_base.err
python - <<'PY'
import re
# find the largest timestamp gaps between consecutive log lines after "MARK loop start"
lines = open('gpurun_out/r5_stall/log_base.err', errors='replace').read().split('\n')
start = next((i for i, l in enumerate(lines) if 'MARK loop start' in l), 0)
ts = []
for i in range(start, len(lines)):
    m = re.search(r'\[pid:\s*\d+\s+tid:\s*(0x[0-9a-f]+)\]', lines[i])
    t = re.match(r':\d+:[^:]*:\s*\d+\s*:\s*(\d+)\s*us', lines[i])
    if t: ts.append((int(t.group(1)), i))
gaps = sorted(((ts[k+1][0]-ts[k][0], ts[k][1]) for k in range(len(ts)-1)), reverse=True)[:6]
out = open('gpurun_out/r5_stall/gaps.txt', 'w')
for g, i in gaps:
    out.write(f"==== gap {g} us after line {i}\n" + '\n'.join(l[:260] for l in lines[max(start, i-6):i+8]) + '\n')
out.close()
print(open('gpurun_out/r5_stall/log_base.out').read())
print(open('gpurun_out/r5_stall/gaps.txt').read()[:6000])
PY
# keep the merge small
head -c 3000000 $O/log_base.err > $O/log_base_head.err; rm -f $O/log_base.err
