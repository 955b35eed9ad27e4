#!/bin/bash
# GPU box: HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and issue counters of the kernels of
# the round-5 solver rounds: the row-polynomial round with MINRES's P and B inside its kernels
# (rbf) and the filter-form round with P inside the staged W product (matern), 129 systems.  Rounds
# launched one by one (RUNLMC_NO_GRAPH): a captured replay runs on after the last round
# asked for, on frozen systems that move nothing, and would dilute the per-launch means.
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r05; mkdir -p $out
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ3="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
for fam in rbf matern; do
  RUNLMC_DEBUG=1 RUNLMC_NO_GRAPH=1 bash $root/tools/pmc_cmd.sh r05_round_$fam "$SQ1" "$SQ3" "FETCH_SIZE" "WRITE_SIZE" -- tools/solve_rounds.py c5 129 11 $fam \
    | grep -E "k_minres2_|k_rp_|k_lr_mix|k_spmv_|k_sf_" > $out/pmc_c5_round_k129_${fam}_summary.txt
  rm -rf $root/gpurun_out/pmc_r05_round_$fam
  cat $out/pmc_c5_round_k129_${fam}_summary.txt
done
