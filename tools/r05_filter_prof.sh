#!/bin/bash
# GPU box: per-kernel durations of the filter family's products (rocprofv3 --kernel-trace --stats)
#   tools/r05_filter_prof.sh <tag> "<families>" [ENV=1 ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O
tag=$1; fams=${2:-matern mix}; shift; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
for k in $fams; do
  rm -rf /tmp/prof_$k
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$k -- python3 $R/tools/one_family.py c5 $k 129 20 > /dev/null 2>&1
  f=$(find /tmp/prof_$k -name "*kernel_stats.csv" | head -1)
  echo "== $k $tag $*"
  if [ -n "$f" ]; then
    cp $f $O/kernel_stats_${k}_$tag.csv
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(s in n for s in ('k_sf', 'k_lr', 'k2_', 'k3_')):
        print('%-46s calls %4s  avg %9.1f us  total %6.2f %%' % (n.split('(')[0][-46:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
  fi
done
