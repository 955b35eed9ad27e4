#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
for ts in 0 1; do for mb in 24 48 96 192; do
  r=$(RUNLMC_TWO_STREAMS=$ts RUNLMC_CHUNK_MB=$mb python3 bench.py --config c2 --steps 50 --warmup 3 --no-cpu --no-nll --no-extra --no-full 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: round(v['roofline_frac'],4) for k,v in d.get('batch_sweep',{}).items()})")
  echo "two_streams=$ts chunk=$mb: $r"
done; done
