#!/bin/bash
# GPU box: per-kernel times of the grid product at the four kernel families
#   tools/prof_families.sh [c5|c2] [batch]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cfg=${1:-c5}; batch=${2:-}
cd /tmp && export TMPDIR=/tmp
out=/tmp/fam_$cfg; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/families.py $cfg $batch > $out/run.txt 2> $out/err.txt
cat $out/run.txt
python3 - <<PY
import csv,glob
f=glob.glob('$out/*/*kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print('  %-60s calls %6s avg us %9.2f total ms %9.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
mkdir -p $root/gpurun_out/prof_families_$cfg
cp $out/*/*kernel_stats.csv $root/gpurun_out/prof_families_$cfg/ 2>/dev/null
cp $out/run.txt $root/gpurun_out/prof_families_$cfg/
