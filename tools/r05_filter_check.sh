#!/bin/bash
# GPU box: the filter family after a kernel change -- parity at full size, then the families' times
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_full_size.py -x -q -k "family or matern" > $O/filter_tests.txt 2>&1
tail -3 $O/filter_tests.txt
python -m pytest tests/test_gpu_suite.py -x -q -k "filter or synth" >> $O/filter_tests.txt 2>&1
tail -2 $O/filter_tests.txt
python tools/families.py c5 > $O/families_c5_$1.txt 2>&1
cat $O/families_c5_$1.txt
bash $R/tools/r05_filter_prof.sh $1 "matern mix"
