#!/bin/bash
# GPU box: SQ / MFMA counters and HBM bytes of the C5 solver round's kernels in the
# row-polynomial form (k_minres2_p, k_minres2_b, k_rp_project, k_lr_mix, k_rp_expand),
# 129 and 17 systems.  Separate counter passes, kernel trace only.
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r04; mkdir -p $out
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ4="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
for k in 129 17; do
  bash $root/tools/pmc_cmd.sh r04_round_$k "$SQ1" "$SQ2" "$SQ3" "$SQ4" "FETCH_SIZE" "WRITE_SIZE" -- tools/solve_rounds.py c5 $k 11 \
    | grep -E "k_minres2_p|k_minres2_b|k_rp_|k_lr_mix" > $out/pmc_c5_rp_round_k${k}_summary.txt
  rm -rf $root/gpurun_out/pmc_r04_round_$k
  cat $out/pmc_c5_rp_round_k${k}_summary.txt
done
