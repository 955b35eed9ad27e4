#!/bin/bash
# GPU box, round 4, first record: tests, published micro-benchmarks in both stopping
# modes, time to the reference's tolerance, per-kernel split of the C5 solver round at
# the full batch (129) and at one rank's share of an 8-way probe split (17)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/gputest.txt; cat $O/gputest.txt
python examples/published_microbench.py > $O/published_microbench.txt 2>&1; cat $O/published_microbench.txt
python tools/time_to_tolerance.py c2 rbf 0 2 > $O/time_to_tolerance_c2.txt 2>&1; cat $O/time_to_tolerance_c2.txt
timeout 600 python tools/time_to_tolerance.py c5 rbf 3000 0 > $O/time_to_tolerance_c5.txt 2>&1; cat $O/time_to_tolerance_c5.txt
cd /tmp; export TMPDIR=/tmp
for nr in 17 129; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5r_$nr -- python3 $R/tools/solve_rounds.py c5 $nr 21 > $O/c5r_$nr.log 2>&1
  grep -v "^[EWI]2026" $O/c5r_$nr.log | tail -3
  f=$(find $O/c5r_$nr -name "*kernel_stats.csv" | head -1); cp $f $O/c5_rounds_k${nr}_kernel_stats.csv
  t=$(find $O/c5r_$nr -name "*kernel_trace.csv" | head -1)
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print("  ", r["Name"][:60].ljust(62), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3), r["Percentage"])
PY
  # per-launch durations of the last full round (between the last two k_minres2_p)
  python3 - <<PY > $O/c5_round_k${nr}_timeline.txt
import csv
rows=[r for r in csv.DictReader(open("$t"))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ps=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_minres2_p')]
a,b=ps[-3],ps[-2]
t0=int(rows[a]['Start_Timestamp'])
print('C5 solver round, $nr systems: wall %.1f us'%((int(rows[b]['Start_Timestamp'])-t0)/1e3))
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-60s start %8.1f  dur %8.1f us'%(r['Kernel_Name'].split('(')[0].replace('void ','')[:60],(s-t0)/1e3,(e-s)/1e3))
PY
  cat $O/c5_round_k${nr}_timeline.txt
  rm -rf $O/c5r_$nr
done
