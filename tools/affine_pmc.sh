#!/bin/bash
export RUNLMC_DEBUG=1   # the switches below are debug hooks (csrc/rl_gridop.hip: read_knobs)
# GPU box: HBM-side bytes of the transform kernels at C2, 1024 vectors, old order vs pair-affine
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
AB_CONTROLS=1 AB_KB=3072 python tools/affine_ab.py c2 17 1024 2>&1 | grep -v amdgpu.ids | tee $O/affine_ab_c2_controls.txt
cd /tmp; export TMPDIR=/tmp
for mode in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_aff${mode}_$c; rm -rf $d; mkdir -p $d
    RUNLMC_AFFINE=$mode RUNLMC_AFFINE_KB=3072 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/one_family.py c2 rbf 1024 20 fft > $d/stdout.txt 2> $d/stderr.txt
  done
  python3 - <<PY | tee -a $O/affine_traffic_c2_k1024.txt
import csv,glob,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
for c in ('FETCH_SIZE','WRITE_SIZE'):
    f=glob.glob('$O/pmc_aff${mode}_%s/**/*counter_collection.csv'%c, recursive=True)
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        if n.startswith(('k2_','k3_')):
            tot[n][r['Counter_Name']]+=float(r['Counter_Value']); 
            if c=='FETCH_SIZE': cnt[n]+=1
print('RUNLMC_AFFINE=$mode  (C2, 1024 vectors; FETCH_SIZE x 2 x 1024 B, WRITE_SIZE x 1024 B, per product)')
for n,c in tot.items():
    print('  %-24s launches %5d  read %8.1f MB  write %8.1f MB'%(n,cnt[n],c['FETCH_SIZE']*2048/20/1e6,c['WRITE_SIZE']*1024/20/1e6))
PY
  rm -rf $O/pmc_aff${mode}_FETCH_SIZE $O/pmc_aff${mode}_WRITE_SIZE
done
