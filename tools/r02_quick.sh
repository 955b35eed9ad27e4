#!/bin/bash
# quick GPU check of a kernel change: tag = $1; GPU suite (optional: SKIP_TESTS=1), C5 / C2 product timings, kernel stats
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
out=gpurun_out/$tag
mkdir -p $out
if [ -z "${SKIP_TESTS:-}" ]; then
  (time python -m pytest tests -q -m gpu ${PYTEST_ARGS:-}) > $out/tests.log 2>&1
  tail -5 $out/tests.log
fi
python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu --no-nll --no-extra > $out/c5.json 2> $out/c5.err
python3 bench.py --config c2 --steps 200 --warmup 20 --no-cpu --no-nll --no-extra > $out/c2.json 2> $out/c2.err
python3 - <<PY
import json
for c in ('c5','c2'):
    try:
        d=json.loads(open('$out/%s.json'%c).read().strip().splitlines()[-1])
        print(c, 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'full ms', round(d['full_mvm']['ms_per_step'],4), 'sweep', {k: round(v['roofline_frac'],4) for k,v in d.get('batch_sweep',{}).items()})
    except Exception as e:
        print(c, 'failed', e, open('$out/%s.err'%c).read()[-800:])
PY
tools/profile.sh ${tag}_c5 --config c5 --steps 5 --warmup 2 --no-extra > $out/profile_c5.txt 2>&1
head -8 $out/profile_c5.txt | cut -c1-200
