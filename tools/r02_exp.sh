#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {
  out=/tmp/e_$1; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --config c5 --steps 10 --warmup 2 --no-cpu --no-nll --no-sweep --no-extra --no-full > $out/bench.json 2> $out/err.txt
  python3 - <<PY
import csv,glob,collections,json
f=glob.glob('$out/*/*kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0]
    d[n].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for n,v in d.items():
    if 'k_lr_project' in n or 'k_lr_expand' in n or 'k_lr_mix' in n:
        big=sorted(x for x in v if x>0.5*max(v))
        print('$1', n[:24], len(big), 'min', big[0], 'median', big[len(big)//2])
try:
    print('$1', 'bench ms', json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['ms_per_step'])
except Exception as e: print(open('$out/err.txt').read()[-500:])
PY
  rm -rf $out
}
run side1
cp $root/runlmc_amd/csrc/librunlmc_hip.so /tmp/keep.so
cp $root/runlmc_amd/csrc/librunlmc_side0.so $root/runlmc_amd/csrc/librunlmc_hip.so
run side0
cp /tmp/keep.so $root/runlmc_amd/csrc/librunlmc_hip.so
run side1_again
