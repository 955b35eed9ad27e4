#!/bin/bash
# GPU box: per-kernel times of three C5 NLL + gradient steps (tools/nll_breakdown.py), one file per
# kernel family: which kernels a step of the final tree spends its time in
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for fam in ${FAMS:-rbf matern}; do
  rm -rf /tmp/nllp; mkdir -p /tmp/nllp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nllp -- python3 $R/tools/nll_breakdown.py c5 128 $fam > /tmp/nllp/out.txt 2> /tmp/nllp/err.txt
  f=$(find /tmp/nllp -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -16 "$f" > $O/c5_nll_${fam}_kernel_stats.csv
  grep "rep" /tmp/nllp/out.txt | tail -3 > $O/c5_nll_${fam}_breakdown.txt
  cat $O/c5_nll_${fam}_breakdown.txt; head -3 $O/c5_nll_${fam}_kernel_stats.csv | cut -c1-120
done
