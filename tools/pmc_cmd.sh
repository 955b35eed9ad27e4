#!/bin/bash
# GPU box: rocprofv3 counter passes over an arbitrary python tool.
#   tools/pmc_cmd.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- tools/x.py [args]
# (separate runs per pass; never combined with sys/hip/hsa tracing)
set -u
tag=$1; shift
passes=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do passes+=("$1"); shift; done
[ $# -gt 0 ] && shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for p in "${passes[@]}"; do
  out=$root/gpurun_out/pmc_${tag}/pass$i
  mkdir -p "$out"
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$out" -- \
      python3 "$script" "$@" > "$out/stdout.txt" 2> "$out/stderr.txt"
  i=$((i+1))
done
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/pmc_${tag}" | grep -v "at::native"
