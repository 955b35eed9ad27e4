#!/bin/bash
# A/B of environment settings on the C5 / C2 products: tools/r02_ab.sh "<env1>" "<env2>" ...
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
for e in "$@"; do
  for c in c5 c2; do
    steps=10; [ $c = c2 ] && steps=200
    r=$(env $e python3 bench.py --config $c --steps $steps --warmup 3 --no-cpu --no-nll --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms %.4f frac %.4f full %.4f sweep %s' % (d['ms_per_step'], d['roofline']['frac'], d['full_mvm']['ms_per_step'], {k: round(v['roofline_frac'],4) for k,v in d.get('batch_sweep',{}).items()}))")
    echo "[$e] $c: $r"
  done
done
