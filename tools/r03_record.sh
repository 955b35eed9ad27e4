#!/bin/bash
# GPU box: the round's record run -> gpurun_out/r03/ (copy what is kept to profiles/r03/)
#   tools/r03_record.sh [quick]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r03; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "$1" != "quick" ]; then
  (cd $root && python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5) > $out/gputest.txt
  cat $out/gputest.txt
fi
# per-kernel times of the default bench command (products only)
rm -rf /tmp/prof_c5; mkdir -p /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu --no-nll --no-sweep --no-full --no-extra > /tmp/prof_c5/bench.json 2> /tmp/prof_c5/err.txt
cp $(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1) $out/c5_products_kernel_stats.csv
head -14 $out/c5_products_kernel_stats.csv | cut -c1-160
# HBM traffic of the product per kernel family (two PMC passes each)
for fam in rbf periodic matern mix fft; do
  rm -rf $root/gpurun_out/pmc_fam_$fam
  kern=$fam; extra=""; [ $fam = fft ] && kern=rbf && extra=fft
  for pass in 0 1; do
    ctr=FETCH_SIZE; [ $pass = 1 ] && ctr=WRITE_SIZE
    o=$root/gpurun_out/pmc_fam_$fam/pass$pass; mkdir -p $o
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $o -- python3 $root/tools/one_family.py c5 $kern 129 10 $extra > $o/stdout.txt 2> $o/stderr.txt
  done
  python3 $root/tools/traffic_families.py $root/gpurun_out/pmc_fam_$fam c5 129 $fam 10 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/one_family.py c5 $fam 129 10 (round 3)" | tee $out/traffic_$fam.txt
done
cp $root/profiles/r03/traffic.json $out/traffic.json
# the default bench command, as the driver runs it (reads the traffic table written above)
python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 600 $out/bench_default.json
python3 $root/tools/families.py c5 > $out/families_c5.txt 2>/dev/null; cat $out/families_c5.txt
python3 $root/tools/families.py c2 > $out/families_c2.txt 2>/dev/null
python3 $root/tools/families.py c2 1024 > $out/families_c2_1024.txt 2>/dev/null
python3 $root/tools/setup_time.py 2>/dev/null > $out/setup_time.txt; cat $out/setup_time.txt
python3 $root/examples/published_microbench.py 2>/dev/null > $out/published_microbench.txt; cat $out/published_microbench.txt
if [ "$1" != "quick" ]; then
  python3 $root/tools/sweep.py rbf 2>/dev/null > $out/sweep_dqm_rbf.txt
  python3 $root/tools/sweep.py matern 2>/dev/null > $out/sweep_dqm_matern.txt
  tail -9 $out/sweep_dqm_matern.txt
  for fam in periodic matern mix; do $root/tools/bench_family.sh $fam; done
  (cd $root && python3 tests/report_iteration_parity.py c5) > $out/iteration_parity_c5.txt 2> $out/iteration_parity_c5.err
  cat $out/iteration_parity_c5.txt
fi
