#!/bin/bash
# GPU box: per-launch durations of the headline product's kernels (the part of tools/r06_record.sh)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r06; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5; mkdir -p /tmp/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu --no-nll --no-sweep --no-full --no-extra --no-families > /tmp/prof_c5/bench.json 2> /tmp/prof_c5/err.txt < /dev/null
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" $out/c5_products_kernel_stats.csv; fi
python3 - <<PY > $out/c5_k129_poly_launch_durations.txt
import csv,glob,collections
fs=glob.glob('/tmp/prof_c5/**/*kernel_trace.csv',recursive=True)
d=collections.defaultdict(list)
if fs:
    for r in csv.DictReader(open(fs[0])):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        if n.startswith(('k_lr_project','k_lr_mix','k_lr_expand')):
            d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('# headline product (C5, 129 vectors, polynomial form, rank 24): per-launch durations in us of the launches of')
print('# the timed region (bench.py --steps 50 --warmup 5 under rocprofv3 --kernel-trace); set-time launches (a few')
print('# dozen rows, < 60 us for the projection / expansion) listed apart')
for n,v in sorted(d.items()):
    big=sorted(x for x in v if x >= (60 if 'mix' not in n else 0))
    small=[x for x in v if x < (60 if 'mix' not in n else 0)]
    if big:
        print('%-24s full-size launches %3d  min %.1f  median %.1f  mean %.1f  max %.1f'%(n,len(big),big[0],big[len(big)//2],sum(big)/len(big),big[-1]))
    if small:
        print('%-24s set-time launches  %3d  mean %.1f'%(n,len(small),sum(small)/len(small)))
PY
cat $out/c5_k129_poly_launch_durations.txt
tail -c 400 /tmp/prof_c5/bench.json
