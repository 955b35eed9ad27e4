#!/bin/bash
export RUNLMC_DEBUG=1   # the switches below are debug hooks (csrc/rl_gridop.hip: read_knobs)
# GPU box: MINRES's B inside the row-polynomial projection against B as its own kernel
# (RUNLMC_NO_RP_FUSE=1), same box: round timelines (tools/r04_rounds.sh) and the NLL + gradient step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O   # (run as: bash tools/rp_fuse_ab.sh 2>&1 | tee gpurun_out/rp_fuse_ab.txt)
for kern in rbf periodic; do
  for mode in fused unfused; do
    [ $mode = unfused ] && export RUNLMC_NO_RP_FUSE=1 || unset RUNLMC_NO_RP_FUSE
    echo "== $kern, $mode"
    KERN=$kern bash $R/tools/r04_rounds.sh 2>&1 | grep -E "per round|k_minres2|k_rp_|k_lr_mix|ten"
  done
done
for mode in fused unfused; do
  [ $mode = unfused ] && export RUNLMC_NO_RP_FUSE=1 || unset RUNLMC_NO_RP_FUSE
  echo "== NLL + gradient, C5 rbf, $mode: 128 probes, then 16 (one rank's share)"
  python3 $R/tools/nll_breakdown.py c5 128 rbf 2>&1 | grep -v "runlmc\]" | tail -4
  python3 $R/tools/nll_breakdown.py c5 16 rbf 2>&1 | grep -v "runlmc\]" | tail -4
done
