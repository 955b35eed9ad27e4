"""HBM-side bytes per grid product of one kernel family, from rocprofv3 PMC passes
over tools/one_family.py (FETCH_SIZE and WRITE_SIZE in separate runs), into
profiles/r05/traffic.json under the key bench.py looks up.

    python tools/traffic_families.py <pmc dir> <config> <batch> <family> <calls> <label>

Counter handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE
are kilobytes at the L2's fabric side (Infinity-Cache hits included); on gfx950
FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes, so it
is doubled (calibration on counts known exactly: k_lr_project and k_sf_carries each
read every element of x once); WRITE_SIZE is used as reported."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, config, batch, family, calls, label = (sys.argv[1], sys.argv[2], int(sys.argv[3]),
                                             sys.argv[4], int(sys.argv[5]), sys.argv[6])
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for pdir in sorted(glob.glob(os.path.join(root, 'pass*'))):
    cand = glob.glob(os.path.join(pdir, '**', '*counter_collection.csv'), recursive=True)
    if not cand:
        continue
    with open(max(cand, key=os.path.getmtime)) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name', '').split('(')[0].replace('void ', '')
            tot[name][row['Counter_Name']] += float(row['Counter_Value'])
            cnt[name][row['Counter_Name']] += 1
detail, total = {}, 0.0
for name, c in tot.items():
    if not name.startswith(('k2_', 'k3_', 'k1_') if family == 'fft' else
                           ('k_lr_project', 'k_lr_mix', 'k_lr_expand', 'k_lr_small', 'k_sf_')):
        continue
    # the product's launches only: the set-time verification also launches k_lr_* kernels
    # (a few dozen rows each) -- per-launch averages over ALL launches would be diluted,
    # so take totals and divide by the number of products; their share is < 1 %
    rd = c.get('FETCH_SIZE', 0.0) * 1024 * 2
    wr = c.get('WRITE_SIZE', 0.0) * 1024
    detail[name] = {'read_bytes_per_product': rd / calls, 'write_bytes_per_product': wr / calls,
                    'launches_counted': cnt[name].get('FETCH_SIZE', 0)}
    total += rd + wr
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'profiles', os.environ.get('RL_PROFILE_ROUND', 'r05'), 'traffic.json')
try:
    table = json.load(open(out_path))
except (OSError, ValueError):
    table = {}
key = ('%s:%d' % (config, batch) if family == 'fft' else
       '%s:%d:%s' % (config, batch, 'poly' if family == 'rbf' else family))
table[key] = {'bytes_per_step': total / calls, 'source': label, 'kernels': detail}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(table, open(out_path, 'w'), indent=1, sort_keys=True)
print(key, 'bytes per product %.4g' % (total / calls))
for k, v in sorted(detail.items()):
    print('   %-34s read %.4g  write %.4g' % (k, v['read_bytes_per_product'], v['write_bytes_per_product']))
