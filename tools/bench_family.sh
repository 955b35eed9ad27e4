#!/bin/bash
# GPU box: the whole bench line (product, full operator, NLL + gradient, C2) for one kernel family
#   tools/bench_family.sh matern
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
k=$1; out=$root/gpurun_out/r03; mkdir -p $out
python3 $root/bench.py --kern $k --steps 20 --warmup 3 --no-families --no-sweep > $out/bench_$k.json 2> $out/bench_$k.err
tail -c 300 $out/bench_$k.err
python3 - <<PY
import json
d = json.load(open("$out/bench_$k.json"))
n, c2 = d["nll_grad"], d["c2"]
print("$k", "product ms %.3f frac %.3f form %s" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["form"]),
      "| full operator ms %.3f" % d["full_mvm"]["ms_per_step"],
      "| nll s %.3f it %.0f res %.3g (eps1 %.3f)" % (n["seconds"], n["iterations_mean"], n["residual_max"], d["nll_grad_eps1"]["seconds"]),
      "| c2 product us %.1f nll ms %.2f" % (c2["ms_per_step"] * 1e3, c2["nll_grad"]["seconds"] * 1e3),
      "| cpu nll s %.1f (%s) speedup %.0f" % (d["cpu_baseline"]["nll_grad"]["seconds"], d["cpu_baseline"]["nll_grad"]["kind"], n.get("speedup_vs_cpu", 0)))
PY
