"""Summarise rocprofv3 counter_collection CSVs: mean counter value per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name', '')
            short = name.split('(')[0].replace('void ', '')
            acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
lines = []
for k in sorted(acc):
    parts = ['%s=%.4g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(acc[k].items())]
    lines.append('%-28s %s' % (k[:28], '  '.join(parts)))
out = '\n'.join(lines)
print(out)
open(os.path.join(root, 'summary.txt'), 'w').write(out + '\n')
