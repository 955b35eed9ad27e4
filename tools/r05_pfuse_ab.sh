#!/bin/bash
# GPU box: MINRES's P inside the row-polynomial expansion (k_minres2_ph + k_rp_expand<.., true>)
# against P as its own kernel (RUNLMC_NO_RP_PFUSE=1), same box: parity tests, C5 round timelines
# at 129 / 17 systems, the NLL + gradient step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_suite.py tests/test_gpu_full_size.py -x -q -k "row_polynomial or minres or solver or gradient" > $O/pfuse_tests.txt 2>&1
tail -3 $O/pfuse_tests.txt
export RUNLMC_DEBUG=1
for mode in fused unfused; do
  if [ $mode = unfused ]; then export RUNLMC_NO_RP_PFUSE=1; else unset RUNLMC_NO_RP_PFUSE; fi
  echo "=== P $mode"
  TAG=$mode KERN=${KERN:-rbf} bash tools/r05_rounds.sh
  cd $R
  python tools/nll_breakdown.py c5 2>&1 | grep -v "^[EWI]2026" | tail -${NLL_LINES:-6}
  python tools/nll_breakdown.py c5 16 2>&1 | grep -v "^[EWI]2026" | tail -${NLL_LINES:-6}
done
