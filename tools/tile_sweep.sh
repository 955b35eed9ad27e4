#!/bin/bash
# tile-size sweep of the three-kernel grid product at one batch size (GPU box)
b=${1:-17}
export RUNLMC_NO_V4=1
for R in 1 2 4; do for TR in 64 128 256; do
  RUNLMC_TILE_R=$R RUNLMC_THR_R=$TR python tools/v4_probe.py $b | sed "s/^/R=$R thrR=$TR  /"
done; done
for C in 4 8 16 32; do for TC in 64 128 256; do
  RUNLMC_TILE_C=$C RUNLMC_THR_C=$TC python tools/v4_probe.py $b | sed "s/^/C=$C thrC=$TC  /"
done; done
