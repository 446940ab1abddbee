#!/bin/bash
# Headline step with and without a one-rank process group, alternating on one box (is the process-group run slower, and does the
# hardware-queue count matter?).  Prints one line per run: label value ms_per_step.
cd $GRAFT_REPO_ROOT
O=gpurun_out/pg_ab.txt
: > $O
F="--no-cpu-baseline --no-extras --no-scale-legs --steps 60 --warmup 10"
one() {  # label, env assignments..., then the command after --
  lbl=$1; shift
  out=$("$@" 2>/dev/null | grep '^{' | tail -1)
  python - "$lbl" "$out" >> $O <<'PY'
import json,sys
d=json.loads(sys.argv[2]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("streams"))
PY
}
for rep in 1 2; do
one plain python bench.py $F
one torchrun python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 $F
one spawn python bench.py --gpus 1 --spawn $F
GPU_MAX_HW_QUEUES=16 one torchrun_q16 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$rep bench.py --gpus 1 $F
GPU_MAX_HW_QUEUES=4 one torchrun_q4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2955$rep bench.py --gpus 1 $F
GPU_MAX_HW_QUEUES=16 one plain_q16 python bench.py $F
done
cat $O
