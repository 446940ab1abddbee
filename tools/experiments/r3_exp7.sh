#!/bin/bash
# round 3, experiment 7: head fork/join at small batch
cd $GRAFT_REPO_ROOT
for b in 1 2 4 1 2 4; do
  echo -n "fork    "; python tools/latency_b1.py $b 2>/dev/null | tail -1
  echo -n "no fork "; EINX_NO_FORK=1 python tools/latency_b1.py $b 2>/dev/null | tail -1
done
