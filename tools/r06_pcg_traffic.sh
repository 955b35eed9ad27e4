#!/bin/bash
# GPU box: HBM-side bytes per launch of the kernels of a preconditioned C5 solve (rocprofv3 --pmc, FETCH_SIZE and
# WRITE_SIZE in passes of their own; counter handling as tools/traffic_families.py: kilobytes at the L2's fabric
# side, FETCH_SIZE doubled on gfx950)      tools/r06_pcg_traffic.sh matern -> gpurun_out/r06/pcg_traffic_<kern>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O
kern=${1:-matern}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/r06_pcg_profile.py c5 $kern > $O/run_pmc_$c.log 2>&1 < /dev/null
done
python3 - "$kern" > $O/pcg_traffic_$kern.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name', '').split('(')[0].replace('void ', '')
            if row['Counter_Name'] != c:
                continue
            tot[name][c] += float(row['Counter_Value'])
            if c == 'FETCH_SIZE':
                cnt[name] += 1
print('C5 %s, two preconditioned solves of 129 systems: HBM-side bytes per launch' % sys.argv[1])
for name in sorted(tot, key=lambda k: -(tot[k]['FETCH_SIZE'] * 2048 + tot[k]['WRITE_SIZE'] * 1024)):
    n = max(cnt[name], 1)
    rd, wr = tot[name]['FETCH_SIZE'] * 1024 * 2 / n, tot[name]['WRITE_SIZE'] * 1024 / n
    if rd + wr < 1e6:
        continue
    print('  %-44s launches %5d   read %8.1f MB   write %8.1f MB' % (name[:44], n, rd / 1e6, wr / 1e6))
PY
cat $O/pcg_traffic_$kern.txt
