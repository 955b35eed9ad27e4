#!/bin/bash
# GPU box: per-kernel durations of one python command (rocprofv3 --kernel-trace --stats)
#   tools/r06_kernel_stats.sh <tag> <script> [args ...]       -> gpurun_out/r06/kernel_stats_<tag>.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 "$@" > $O/run_$tag.log 2>&1 < /dev/null
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" 2>/dev/null | head -1)
echo "== $tag: $*"
tail -n 3 $O/run_$tag.log | grep -v "rocprofv3\]" | cut -c1-400
if [ -n "$f" ] && [ -f "$f" ]; then
  cp "$f" $O/kernel_stats_$tag.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-60s calls %6s  avg %9.2f us  total %6.2f %%' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
else
  echo "no kernel_stats.csv under /tmp/prof_$tag"
fi
