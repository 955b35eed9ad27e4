#!/bin/bash
# rocprofv3 kernel stats of the NLL+gradient step (bench.py with the MVM loop cut short)
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- \
    python3 "$root/bench.py" --no-cpu --steps 2 --warmup 1 "$@" > "$out/bench.json" 2> "$out/stderr.txt"
find "$out" -name '*kernel_stats.csv' | head -1 | xargs -r head -20
