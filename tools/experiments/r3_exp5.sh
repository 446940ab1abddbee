#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 4 8 16 32 64; do python tools/up_bench.py $b 2>/dev/null; done
