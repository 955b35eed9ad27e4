#!/bin/bash
# GPU box: durations of the headline product's kernels in LAUNCH ORDER (is the spread of
# k_lr_project a drift over time or an alternation?)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/seq; rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -- python3 $R/tools/one_family.py c5 rbf 129 60 > /dev/null 2>&1
python3 - <<PY | tee $O/c5_k129_launch_sequence.txt
import csv,glob
f=glob.glob('/tmp/seq/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
seq={'k_lr_project<24>':[], 'k_lr_expand<24>':[]}
gaps=[]
prev=None
for r in rows:
    n=r['Kernel_Name'].split('(')[0].replace('void ','')
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if n in seq and d>60: seq[n].append(d)
print('C5, 129 vectors, rank 24: kernel durations (us) in launch order, 60 products back to back')
for n,v in seq.items():
    print(n, ' '.join('%.0f'%x for x in v))
PY
