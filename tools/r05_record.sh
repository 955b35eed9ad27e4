#!/bin/bash
# GPU box: the round's record run -> gpurun_out/r05/ (what is kept is copied to profiles/r05/)
#   tools/r05_record.sh [quick]
mode=$1      # (kept apart: `set --` below reuses the positional parameters)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r05; mkdir -p $out
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
cd /tmp && export TMPDIR=/tmp
if [ "$mode" != "quick" ]; then
  (cd $root && timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5) > $out/gputest.txt
  cat $out/gputest.txt
fi
# per-kernel times of the default bench command (products only) + per-launch durations of the
# headline's three kernels (full-size launches only: the set-time verification launches the same
# kernels on a few dozen rows)
rm -rf /tmp/prof_c5; mkdir -p /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu --no-nll --no-sweep --no-full --no-extra --no-families > /tmp/prof_c5/bench.json 2> /tmp/prof_c5/err.txt
cp $(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1) $out/c5_products_kernel_stats.csv
python3 - <<PY > $out/c5_k129_poly_launch_durations.txt
import csv,glob,collections
f=glob.glob('/tmp/prof_c5/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('void ','')
    if n.startswith(('k_lr_project','k_lr_mix','k_lr_expand')):
        d[n].append(((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, int(r.get('Grid_Size_X',r.get('Grid_Size',0)) or 0)))
print('# headline product (C5, 129 vectors, polynomial form, rank 24): per-launch durations in us of the launches of')
print('# the timed region (bench.py --steps 50 --warmup 5 under rocprofv3 --kernel-trace); set-time launches (a few')
print('# dozen rows, < 60 us for the projection / expansion) listed apart')
for n,v in sorted(d.items()):
    big=[x for x,_ in v if x>= (60 if 'mix' not in n else 0)]
    small=[x for x,_ in v if x < (60 if 'mix' not in n else 0)]
    big.sort()
    if big:
        print('%-20s full-size launches %3d  min %.1f  median %.1f  mean %.1f  max %.1f'%(n,len(big),big[0],big[len(big)//2],sum(big)/len(big),big[-1]))
    if small:
        print('%-20s set-time launches  %3d  mean %.1f'%(n,len(small),sum(small)/len(small)))
PY
cat $out/c5_k129_poly_launch_durations.txt
# HBM traffic of the product per kernel family (two PMC passes each), C5 129 vectors; C2 on the
# transform kernels at 17 and 1024 vectors
rm -f $root/profiles/r05/traffic.json
for spec in "c5 rbf 129 rbf" "c5 periodic 129 periodic" "c5 matern 129 matern" "c5 mix 129 mix" "c5 rbf 129 fft" "c2 rbf 17 fft" "c2 rbf 1024 fft" "c2 rbf 1024 rbf"; do
  set -- $spec; cfg=$1; kern=$2; batch=$3; fam=$4
  extra=""; [ $fam = fft ] && extra=fft
  calls=10; [ $cfg = c2 ] && calls=50
  rm -rf $root/gpurun_out/pmc_fam
  for pass in 0 1; do
    ctr=FETCH_SIZE; [ $pass = 1 ] && ctr=WRITE_SIZE
    o=$root/gpurun_out/pmc_fam/pass$pass; mkdir -p $o
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $o -- python3 $root/tools/one_family.py $cfg $kern $batch $calls $extra > $o/stdout.txt 2> $o/stderr.txt
  done
  python3 $root/tools/traffic_families.py $root/gpurun_out/pmc_fam $cfg $batch $fam $calls "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/one_family.py $cfg $kern $batch $calls $extra (round 5)" | tee $out/traffic_${cfg}_${batch}_${fam}.txt
done
rm -rf $root/gpurun_out/pmc_fam
cp $root/profiles/r05/traffic.json $out/traffic.json
# SQ counters of the final kernels, per family (C5, 129 vectors) and of the single-tile kernel
for fam in rbf periodic matern mix; do
  bash $root/tools/pmc_cmd.sh r05_sq_$fam "$SQ1" "$SQ2" "$SQ3" -- tools/one_family.py c5 $fam 129 10 | grep -E "k_lr_project|k_lr_expand|k_lr_mix|k_sf_" > $out/pmc_c5_${fam}_summary.txt
  rm -rf $root/gpurun_out/pmc_r05_sq_$fam
done
bash $root/tools/pmc_cmd.sh r05_sq_k1 "$SQ1" "$SQ2" "$SQ3" -- tools/k1_product_run.py 13 238 256 20 | grep -E "k1_product" > $out/pmc_k1_product_fx2007_shape_summary.txt
python3 $root/tools/k1_product_run.py 13 238 256 50 2>/dev/null >> $out/pmc_k1_product_fx2007_shape_summary.txt
python3 $root/tools/k1_product_run.py 13 238 2048 20 2>/dev/null >> $out/pmc_k1_product_fx2007_shape_summary.txt
python3 $root/tools/k1_product_run.py 4 504 2048 20 2>/dev/null >> $out/pmc_k1_product_fx2007_shape_summary.txt
rm -rf $root/gpurun_out/pmc_r05_sq_k1
cat $out/pmc_c5_matern_summary.txt $out/pmc_k1_product_fx2007_shape_summary.txt
# the default bench command, as the driver runs it (reads the traffic table written above)
python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 1500 $out/bench_default.json
python3 $root/tools/families.py c5 2>/dev/null > $out/families_c5.txt; cat $out/families_c5.txt
python3 $root/tools/families.py c2 2>/dev/null > $out/families_c2.txt
python3 $root/tools/families.py c2 1024 2>/dev/null > $out/families_c2_1024.txt
python3 $root/tools/setup_time.py 2>/dev/null > $out/setup_time.txt; cat $out/setup_time.txt
(cd $root && python3 tests/report_published_errors.py) > $out/published_errors.txt 2> $out/published_errors.err; tail -20 $out/published_errors.txt
(cd $root && python3 examples/published_microbench.py) 2>/dev/null > $out/published_microbench.txt; cat $out/published_microbench.txt
if [ "$mode" != "quick" ]; then
  python3 $root/tools/sweep.py rbf 2>/dev/null > $out/sweep_dqm_rbf.txt
  python3 $root/tools/sweep.py matern 2>/dev/null > $out/sweep_dqm_matern.txt
  tail -9 $out/sweep_dqm_matern.txt
  for fam in periodic matern mix; do
    python3 $root/bench.py --kern $fam --steps 20 --warmup 3 --no-families --no-sweep > $out/bench_$fam.json 2> $out/bench_$fam.err
    python3 -c "
import json; d=json.load(open('$out/bench_$fam.json')); n=d['nll_grad']
print('$fam product ms %.3f frac %.3f | nll s %.3f it %.0f | share ceiling %.2f' % (d['ms_per_step'], d['roofline']['frac'], n['seconds'], n['iterations_mean'], n.get('projected_strong_scaling_8gpu',{}).get('ceiling',0)))"
  done
  (cd $root && python3 tests/report_iteration_parity.py c5) > $out/iteration_parity_c5.txt 2> $out/iteration_parity_c5.err
  cat $out/iteration_parity_c5.txt
fi
