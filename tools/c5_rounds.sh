#!/bin/bash
# per-kernel split of the C5 solver rounds for a few batch sizes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nr in ${NRHS:-43 129}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5r_$nr -- python3 $R/tools/solve_rounds.py c5 $nr 11 > $R/gpurun_out/c5r_$nr.log 2>&1
  grep -v "^[EWI]2026" $R/gpurun_out/c5r_$nr.log | tail -3
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/c5r_$nr/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print("  ", r["Name"][:50].ljust(52), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3), r["Percentage"])
PY
done
