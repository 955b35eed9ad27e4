#!/bin/bash
# GPU box: the round-6 records (profiles/r06).   tools/r06_record.sh [quick]
#   1. PMC traffic of the products the bench line quotes (FETCH_SIZE / WRITE_SIZE, separate passes)
#   2. the default bench command as the driver runs it, and its rocprofv3 --kernel-trace --stats summary
#   3. the three other kernel families' bench lines, family products, small-batch product, direct-solve probes
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r06; mkdir -p $out
export RL_PROFILE_ROUND=r06
cd /tmp && export TMPDIR=/tmp
rm -f $root/profiles/r06/traffic.json
for spec in "c5 rbf 129 rbf" "c5 rbf 129 fft" "c5 periodic 129 periodic" "c5 matern 129 matern" "c5 mix 129 mix" "c2 rbf 17 rbf" "c2 rbf 17 fft"; do
  set -- $spec; cfg=$1; kern=$2; batch=$3; fam=$4
  extra=""; [ $fam = fft ] && extra=fft
  calls=10; [ $cfg = c2 ] && calls=50
  rm -rf $root/gpurun_out/pmc_fam
  for pass in 0 1; do
    ctr=FETCH_SIZE; [ $pass = 1 ] && ctr=WRITE_SIZE
    o=$root/gpurun_out/pmc_fam/pass$pass; mkdir -p $o
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $o -- python3 $root/tools/one_family_default.py $cfg $kern $batch $calls $extra > $o/stdout.txt 2> $o/stderr.txt < /dev/null
  done
  key=$fam; [ $fam = rbf ] && key=poly
  python3 $root/tools/traffic_families.py $root/gpurun_out/pmc_fam $cfg $batch $key $calls "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/one_family_default.py $cfg $kern $batch $calls $extra (round 6)" < /dev/null | tee $out/traffic_${cfg}_${batch}_${fam}.txt
done
rm -rf $root/gpurun_out/pmc_fam
cp $root/profiles/r06/traffic.json $out/traffic.json 2>/dev/null
# the default bench command, as the driver runs it (reads the traffic table written above)
cd $root
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err < /dev/null
tail -c 600 $out/bench_default.err
cd /tmp
rm -rf /tmp/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $root/bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err < /dev/null
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" $out/bench_default_kernel_stats.csv; fi
if [ "${1:-}" != quick ]; then
  cd $root
  for k in periodic matern mix; do
    python3 bench.py --kern $k --no-cpu --no-extra > $out/bench_$k.json 2> $out/bench_$k.err < /dev/null
  done
  python3 tools/families.py c5 2>/dev/null > $out/families_c5.txt < /dev/null
  python3 tools/families.py c2 2>/dev/null > $out/families_c2.txt < /dev/null
  python3 tools/r06_small_batch.py 2>/dev/null > $out/small_batch_c2.txt < /dev/null
  python3 tools/r06_direct_probe.py c5 rbf 2>/dev/null | tail -1 > $out/direct_probe_c5_rbf.txt < /dev/null
  python3 tools/r06_direct_probe.py c5 periodic 2>/dev/null | tail -1 > $out/direct_probe_c5_periodic.txt < /dev/null
  python3 tools/r06_direct_probe.py c2 rbf 2>/dev/null | tail -1 > $out/direct_probe_c2_rbf.txt < /dev/null
  python3 tools/r06_generate_stages.py 2>/dev/null > $out/generate_stages_c5.txt < /dev/null
  python3 tools/nll_breakdown.py c5 128 rbf 2>/dev/null > $out/nll_breakdown_c5_rbf.txt < /dev/null
  python3 tools/nll_breakdown.py c5 16 rbf 2>/dev/null >> $out/nll_breakdown_c5_rbf.txt < /dev/null
fi
ls $out
