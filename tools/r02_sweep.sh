#!/bin/bash
# C5 product: streams x chunk size sweep
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
out=gpurun_out/sweep_$1; mkdir -p $out
for st in 1 2 3 4; do for mb in 64 96 128 192 288; do
  r=$(RUNLMC_STREAMS=$st RUNLMC_CHUNK_MB=$mb python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu --no-nll --no-extra --no-sweep 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4))")
  echo "streams=$st chunk=$mb : $r" | tee -a $out/sweep.txt
done; done
