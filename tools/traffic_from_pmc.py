"""HBM-side bytes per bench step of the grid product from rocprofv3 PMC passes
(tools/pmc.sh: FETCH_SIZE and WRITE_SIZE in separate runs) -> profiles/r02/traffic.json.

    python tools/traffic_from_pmc.py <pmc dir> <config> <batch> <steps incl. warm-up> <source label>

Counter handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are kilobytes at the L2's fabric side (Infinity-Cache hits included);
on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads,
so it is doubled for the kernels whose reads are the complex intermediates (row
kernel, adjoint column kernel); the forward column kernel reads 8-byte reals and
is left as reported.  WRITE_SIZE is used as reported (uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, config, batch, steps, label = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
tot = defaultdict(lambda: defaultdict(float))
# one file per counter pass (pass0, pass1, ...): the newest, should a directory
# hold the output of more than one run
paths = []
for pdir in sorted(glob.glob(os.path.join(root, 'pass*'))):
    cand = glob.glob(os.path.join(pdir, '**', '*counter_collection.csv'), recursive=True)
    if cand:
        paths.append(max(cand, key=os.path.getmtime))
for path in paths:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name', '').split('(')[0].replace('void ', '')
            tot[name][row['Counter_Name']] += float(row['Counter_Value'])
def entry(prefixes, doubled):
    """Sum of the kernels whose names start with one of `prefixes`."""
    bytes_total, detail = 0.0, {}
    for name, c in tot.items():
        if not name.startswith(prefixes):
            continue
        double = doubled(name)
        rd = c.get('FETCH_SIZE', 0.0) * 1024 * (2 if double else 1)
        wr = c.get('WRITE_SIZE', 0.0) * 1024
        detail[name] = {'read_bytes_per_step': rd / steps, 'write_bytes_per_step': wr / steps,
                        'fetch_doubled': double}
        bytes_total += rd + wr
    return bytes_total, detail


out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'profiles', 'r02', 'traffic.json')
try:
    table = json.load(open(out_path))
except (OSError, ValueError):
    table = {}
# transform kernels (bench.py times them beside the polynomial form, same step count)
b_fft, d_fft = entry(('k2_', 'k3_', 'k1_product'), lambda n: not n.startswith('k2_cols_fwd'))
if d_fft:
    table['%s:%d' % (config, batch)] = {'bytes_per_step': b_fft / steps, 'source': label,
                                        'kernels': d_fft}
    print(config, batch, 'transform kernels: bytes per step %.4g' % (b_fft / steps))
# polynomial form.  Calibrated on byte counts that are known exactly: the
# projection reads every element of x once (C5: 1.032 GB; FETCH_SIZE reports
# 519.6 MB per launch -- its 512-byte-per-wave loads are 128-byte requests tallied
# at 64, the guide's factor 2) and writes 49 x 1290 x 24 partial sums (12.14 MB;
# WRITE_SIZE reports 12.14 MB: as reported), the expansion writes y once (1.032 GB;
# WRITE_SIZE 1.0324 GB).  So FETCH_SIZE doubled, WRITE_SIZE as reported.  (The
# handful of set-time launches of these kernels -- r rows each -- are in the
# sums: < 1 %.)
b_lr, d_lr = entry(('k_lr_',), lambda n: True)
if d_lr:
    table['%s:%d:poly' % (config, batch)] = {'bytes_per_step': b_lr / steps, 'source': label,
                                             'kernels': d_lr}
    print(config, batch, 'polynomial form: bytes per step %.4g' % (b_lr / steps))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(table, open(out_path, 'w'), indent=1, sort_keys=True)
