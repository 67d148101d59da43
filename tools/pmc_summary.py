"""Average per-dispatch PMC counter values per kernel from rocprofv3 counter_collection CSVs.
usage: python tools/pmc_summary.py [--all] <dir-or-csv> [...]  (tools aid; not part of the product)
Without --all only the inference path's fused kernels are listed; --all lists every kernel (the training step's passes)."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.match(r'(?:void )?(?:\(anonymous namespace\)::|fear::)?([A-Za-z0-9_]+)(<[^>]*>)?', name)
    s = m.group(1) + (m.group(2) or '') if m else name
    return s[:70]


def main():
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    every = '--all' in sys.argv
    for arg in [a for a in sys.argv[1:] if a != '--all']:
        files = [arg] if arg.endswith('.csv') else glob.glob(os.path.join(arg, '**', '*counter_collection.csv'), recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                vals[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    counters = sorted({c for k in vals.values() for c in k})
    for k, d in vals.items():
        if not every and not ('fused' in k or 'chain' in k or 'tile' in k or 'pw_mfma' in k or 'sep16' in k or 'e1pair' in k):
            continue
        print(k)
        for c in counters:
            if c in d:
                v = d[c]
                print(f'    {c:34s} {sum(v) / len(v):16.0f}  (n={len(v)})')


if __name__ == '__main__':
    main()
