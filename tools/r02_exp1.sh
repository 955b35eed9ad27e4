#!/bin/bash
# experiment: per-kernel durations of the C5 product, one stream, and PMC of the k3 row kernel
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
out=gpurun_out/exp1; mkdir -p $out
for mb in 96 192; do
  RUNLMC_CHUNK_MB=$mb RUNLMC_TWO_STREAMS=0 tools/profile.sh exp1_one_$mb --config c5 --steps 5 --warmup 2 --no-extra > $out/one_$mb.txt 2>&1
done
RUNLMC_CHUNK_MB=96 RUNLMC_TWO_STREAMS=0 RUNLMC_NO_K3=1 tools/profile.sh exp1_k2_96 --config c5 --steps 5 --warmup 2 --no-extra > $out/k2_96.txt 2>&1
RUNLMC_CHUNK_MB=96 RUNLMC_TWO_STREAMS=0 tools/pmc.sh exp1 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" -- --config c5 --steps 3 --warmup 1 --no-extra > $out/pmc.txt 2>&1
for f in one_96 one_192 k2_96; do echo == $f; head -4 $out/$f.txt | cut -c1-160; done
cat $out/pmc.txt | grep -v rocclr
