#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py.
#   tools/profile.sh <tag> [bench.py args...]
# Writes gpurun_out/prof_<tag>/ (CSV stats + the bench JSON line).
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- \
    python3 "$root/bench.py" --no-cpu --no-nll --no-sweep "$@" > "$out/bench.json" 2> "$out/stderr.txt"
find "$out" -name '*kernel_stats.csv' | head -1 | xargs -r head -12
cat "$out/bench.json"
