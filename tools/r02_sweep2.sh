#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
out=gpurun_out/sweep_$1; mkdir -p $out
run() { r=$(env "$@" python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu --no-nll --no-extra --no-sweep 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"); echo "$* : $r" | tee -a $out/sweep.txt; }
run RUNLMC_CHUNK_MB=64 RUNLMC_TILE_C=16
run RUNLMC_CHUNK_MB=64 RUNLMC_TILE_C=4
run RUNLMC_CHUNK_MB=64 RUNLMC_TILE_C=8 RUNLMC_THR_C=256
run RUNLMC_CHUNK_MB=64 RUNLMC_TILE_C=16 RUNLMC_THR_C=256
run RUNLMC_CHUNK_MB=64 RUNLMC_NO_MIXTAB=1
run RUNLMC_CHUNK_MB=64 RUNLMC_NO_K3=1
run RUNLMC_CHUNK_MB=32
run RUNLMC_CHUNK_MB=48
for c in "RUNLMC_TILE_C=8" "RUNLMC_TILE_C=16"; do
RUNLMC_STREAMS=1 RUNLMC_CHUNK_MB=96 env $c tools/profile.sh sw2 --config c5 --steps 5 --warmup 2 --no-extra | head -4 | cut -c1-110,200-280
done
