#!/bin/bash
# GPU box, round 4, second record: sweep with short grids in the structured forms,
# mid-solve timelines of the C5 solver round (129 systems and one rank's 17), where the
# NLL + gradient step spends its time at the full batch and at the 8-way share
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_suite.py -x -q -m gpu -k "polynomial or reference_rule or probe_dtypes or filter" 2>&1 | tail -3
python tools/sweep.py rbf > $O/sweep_dqm_rbf.txt 2>&1; cat $O/sweep_dqm_rbf.txt
for n in 128 16; do python tools/nll_breakdown.py c5 $n rbf 2>&1 | grep -v amdgpu.ids | tee -a $O/nll_breakdown_c5.txt; done
cd /tmp; export TMPDIR=/tmp
for nr in 17 129; do
  rocprofv3 --kernel-trace --output-format csv -d $O/c5r_$nr -- python3 $R/tools/solve_rounds.py c5 $nr 41 > $O/c5r_$nr.log 2>&1
  grep -v "^[EWI]2026" $O/c5r_$nr.log | tail -3
  t=$(find $O/c5r_$nr -name "*kernel_trace.csv" | head -1)
  python3 - <<PY > $O/c5_round_k${nr}_timeline.txt
import csv
rows=[r for r in csv.DictReader(open("$t"))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ps=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_minres2_p')]
# a round in the middle of the LAST solve (41 rounds + one idle replay of 10): 25 rounds before the end
a,b=ps[-26],ps[-25]
t0=int(rows[a]['Start_Timestamp'])
print('C5 solver round, $nr systems, mid-solve: wall %.1f us'%((int(rows[b]['Start_Timestamp'])-t0)/1e3))
busy=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    busy+=e-s
    print('%-60s start %8.1f  dur %8.1f us'%(r['Kernel_Name'].split('(')[0].replace('void ','')[:60],(s-t0)/1e3,(e-s)/1e3))
print('sum of kernel durations %.1f us'%(busy/1e3))
# ten consecutive rounds
w=[(int(rows[ps[-26+i+1]]['Start_Timestamp'])-int(rows[ps[-26+i]]['Start_Timestamp']))/1e3 for i in range(10)]
print('ten consecutive rounds (us):',' '.join('%.0f'%x for x in w))
PY
  cat $O/c5_round_k${nr}_timeline.txt
  rm -rf $O/c5r_$nr
done
