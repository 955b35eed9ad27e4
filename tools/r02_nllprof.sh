#!/bin/bash
# GPU box: kernel stats of ONE C5 NLL+gradient evaluation (eps = 0.1 and eps = 1): tag = $1
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/nllprof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- \
    python3 "$root/bench.py" --config c5 --steps 3 --warmup 1 --no-cpu --no-sweep --no-full --no-extra \
    > "$out/bench.json" 2> "$out/stderr.txt"
find "$out" -name '*kernel_stats.csv' | head -1 | xargs -r head -25 | cut -c1-220
python3 - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('ms_per_step','nll_grad','nll_grad_eps1') if k in d})
PY
rm -f $out/*/*kernel_trace.csv
