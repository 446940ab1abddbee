#!/bin/bash
# the driver's N=1 command on a fresh box, timed, with the contract keys of its JSON line
cd $GRAFT_REPO_ROOT
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_like.json 2> gpurun_out/driver_like.err
t1=$(date +%s.%N)
echo "wall $(echo "$t1 - $t0" | bc) s"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/driver_like.json") if x.startswith("{")]
d=json.loads(l[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data")})
print(d["config"]["workload"][:80]); print(d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d.get("verified_pairs"), d.get("streams"))
PY
