#!/bin/bash
# round-2 baseline on the GPU box: C5 129-vector product, kernel stats, chunk sweep, PMC
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
mkdir -p gpurun_out/r02base
for mb in 48 96 192; do for ts in 0 1; do
  echo "chunk=$mb two_streams=$ts" >> gpurun_out/r02base/chunk_sweep.txt
  RUNLMC_CHUNK_MB=$mb RUNLMC_TWO_STREAMS=$ts python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu --no-nll --no-sweep >> gpurun_out/r02base/chunk_sweep.txt 2>&1
done; done
tools/profile.sh r02base_c5 --config c5 --steps 10 --warmup 2 > gpurun_out/r02base/profile.txt 2>&1
RUNLMC_CHUNK_MB=96 RUNLMC_TWO_STREAMS=0 tools/profile.sh r02base_c5_96 --config c5 --steps 10 --warmup 2 > gpurun_out/r02base/profile96.txt 2>&1
tools/pmc.sh r02base_c5 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" -- --config c5 --steps 3 --warmup 1 > gpurun_out/r02base/pmc.txt 2>&1
