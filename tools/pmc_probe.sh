#!/bin/bash
# Run on the GPU box: rocprofv3 --pmc passes over tools/v4_probe.py, or $PROBE (counters only).
#   tools/pmc_probe.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- [probe args]
set -u
tag=$1; shift
passes=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do passes+=("$1"); shift; done
[ $# -gt 0 ] && shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for p in "${passes[@]}"; do
  out=$root/gpurun_out/pmc_${tag}/pass$i
  mkdir -p "$out"
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$out" -- \
      python3 "$root/${PROBE:-tools/v4_probe.py}" "$@" > "$out/probe.txt" 2> "$out/stderr.txt"
  i=$((i+1))
done
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/pmc_${tag}"
