root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r06; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $root/bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err < /dev/null
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" $out/bench_default_kernel_stats.csv; head -6 "$f" | cut -c1-60,150-260; fi
rm -rf /tmp/prof_c5; mkdir -p /tmp/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu --no-nll --no-sweep --no-full --no-extra --no-families > /tmp/prof_c5/bench.json 2> /tmp/prof_c5/err.txt < /dev/null
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" $out/c5_products_kernel_stats.csv; grep -E "k_lr_project|k_lr_mix|k_lr_expand" "$f" | cut -c1-40,150-260; fi
