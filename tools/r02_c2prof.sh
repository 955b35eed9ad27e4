#!/bin/bash
# GPU box: C2 NLL + gradient step with the opt-in polynomial rounds against the default
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {
  python3 $root/bench.py --config c2 --steps 50 --warmup 5 --no-cpu --no-extra --no-sweep --no-full 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'nll ms', round(d['nll_grad']['seconds']*1e3,3), d['nll_grad']['iterations_mean'], 'eps1', round(d['nll_grad_eps1']['seconds']*1e3,3))"
}
run default
export RUNLMC_POLY_ROUND=1
run polynomial_rounds
out=/tmp/c2p; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --config c2 --steps 20 --warmup 5 --no-cpu --no-extra --no-sweep --no-full > $out/b.json 2> $out/err.txt
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob('$out/*/*kernel_stats.csv')[0])))
for r in rows[:4]:
    print('  %-40s calls %6s avg us %8.2f' % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
