"""Round 6: nll_grad.seconds of a bench line on stdin (python bench.py ... | python tools/r06_nll_seconds.py label)."""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], round(d['nll_grad']['seconds'] * 1e3, 2), 'ms; on device', round(d['nll_grad']['probes_on_device']['seconds'] * 1e3, 2))
