#!/bin/bash
# round-2 record run on the GPU box: kernel stats + PMC (traffic, waits) of the
# C5 and C2 grid products, the default bench line, the fits.  Outputs under
# gpurun_out/final/ (copied into profiles/r02/ afterwards).
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
out=gpurun_out/final; mkdir -p $out
common="--no-cpu --no-nll --no-sweep --no-extra --no-full"
tools/profile.sh final_c5 --config c5 --steps 10 --warmup 2 --no-extra --no-full > $out/profile_c5.txt 2>&1
tools/profile.sh final_c2 --config c2 --steps 100 --warmup 10 --no-extra --no-full > $out/profile_c2.txt 2>&1
tools/pmc.sh final_c5 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- --config c5 --steps 4 --warmup 1 --no-extra --no-full > $out/pmc_c5.txt 2>&1
tools/pmc.sh final_c2 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- --config c2 --steps 45 --warmup 5 --no-extra --no-full > $out/pmc_c2.txt 2>&1
RUNLMC_CHUNK_MB=1000000 RUNLMC_STREAMS=1 tools/pmc.sh final_c2_k1024 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- --config c2 --batch 1024 --steps 5 --warmup 1 --no-extra --no-full > $out/pmc_c2_k1024.txt 2>&1
cp -r gpurun_out/pmc_final_c5 gpurun_out/pmc_final_c2 $out/ 2>/dev/null
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
(time python3 bench.py) > $out/bench_default.txt 2>&1
tools/r02_nllprof.sh final > $out/nllprof.txt 2>&1
python3 tools/grad_accuracy.py > $out/grad_accuracy.txt 2>&1
python3 tools/mall_probe.py > $out/mall_probe.txt 2>&1
python3 tools/form_crossover.py > $out/form_crossover.txt 2>&1
for w in fx2007 weather weather1000 synth; do python3 examples/fit_real_data.py $w 10 > $out/fit_$w.txt 2>&1; tail -1 $out/fit_$w.txt; done
tail -c 1500 $out/bench_default.txt
