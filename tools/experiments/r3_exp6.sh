#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "--no-scale-legs" "" "--no-scale-legs"; do
  python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', d['value'], [ (s['stage'][:20], s.get('ms'), s.get('frac')) for s in d['roofline_stages'][:2]])"
done
