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
# per-launch durations of the headline's three kernels (full-size launches only: the set-time verification
# launches the same kernels on a few dozen rows)
rm -rf /tmp/prof_c5; mkdir -p /tmp/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu --no-nll --no-sweep --no-full --no-extra --no-families > /tmp/prof_c5/bench.json 2> /tmp/prof_c5/err.txt < /dev/null
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" $out/c5_products_kernel_stats.csv; fi
python3 - <<PY > $out/c5_k129_poly_launch_durations.txt
import csv,glob,collections
fs=glob.glob('/tmp/prof_c5/**/*kernel_trace.csv',recursive=True)
d=collections.defaultdict(list)
if fs:
    for r in csv.DictReader(open(fs[0])):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        if n.startswith(('k_lr_project','k_lr_mix','k_lr_expand')):
            d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('# headline product (C5, 129 vectors, polynomial form, rank 24): per-launch durations in us of the launches of')
print('# the timed region (bench.py --steps 50 --warmup 5 under rocprofv3 --kernel-trace); set-time launches (a few')
print('# dozen rows, < 60 us for the projection / expansion) listed apart')
for n,v in sorted(d.items()):
    big=sorted(x for x in v if x >= (60 if 'mix' not in n else 0))
    small=[x for x in v if x < (60 if 'mix' not in n else 0)]
    if big:
        print('%-24s full-size launches %3d  min %.1f  median %.1f  mean %.1f  max %.1f'%(n,len(big),big[0],big[len(big)//2],sum(big)/len(big),big[-1]))
    if small:
        print('%-24s set-time launches  %3d  mean %.1f'%(n,len(small),sum(small)/len(small)))
PY
cat $out/c5_k129_poly_launch_durations.txt
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
  for a in "c5 mix" "c5 matern" "c2 mix" "c2 matern"; do
    python3 tools/r06_pcg_probe.py $a 2>/dev/null | tail -1 > $out/pcg_probe_$(echo $a | tr " " _).txt < /dev/null
  done
  python3 tools/r06_generate_stages.py 2>/dev/null > $out/generate_stages_c5.txt < /dev/null
  python3 tools/nll_breakdown.py c5 128 rbf 2>/dev/null > $out/nll_breakdown_c5_rbf.txt < /dev/null
  python3 tools/nll_breakdown.py c5 16 rbf 2>/dev/null >> $out/nll_breakdown_c5_rbf.txt < /dev/null
fi
ls $out
