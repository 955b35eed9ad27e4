#!/bin/bash
# GPU box: mid-solve timeline of the C5 solver round at 129 and 17 systems
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for nr in ${NRHS:-129 17}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/c5r_$nr -- python3 $R/tools/solve_rounds.py c5 $nr 41 ${KERN:-rbf} > $O/c5r_$nr.log 2>&1
  grep -v "^[EWI]2026" $O/c5r_$nr.log | tail -2
  t=$(find $O/c5r_$nr -name "*kernel_trace.csv" | head -1)
  python3 - <<PY > $O/c5_round_k${nr}_${TAG:-x}_timeline.txt
import csv
rows=[r for r in csv.DictReader(open("$t"))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ps=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_minres2_p')]
a,b=ps[-26],ps[-25]
t0=int(rows[a]['Start_Timestamp'])
print('C5 solver round (${KERN:-rbf}), $nr systems, mid-solve: wall %.1f us'%((int(rows[b]['Start_Timestamp'])-t0)/1e3))
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-60s start %8.1f  dur %8.1f us'%(r['Kernel_Name'].split('(')[0].replace('void ','')[:60],(s-t0)/1e3,(e-s)/1e3))
w=[(int(rows[ps[-26+i+1]]['Start_Timestamp'])-int(rows[ps[-26+i]]['Start_Timestamp']))/1e3 for i in range(10)]
print('ten consecutive rounds (us):',' '.join('%.0f'%x for x in w))
PY
  cat $O/c5_round_k${nr}_${TAG:-x}_timeline.txt
  rm -rf $O/c5r_$nr
done
