#!/usr/bin/env python3
"""Per-kernel means of the counters collected by collect_pmc.sh: one row per (pass, kernel, counter) with the mean counter
value per launch, the launch count and the mean duration.  Usage: summarize_pmc.py gpurun_out/pmc_<tag> > summary.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'pfa::([A-Za-z0-9_]+)', name)
    return m.group(1) if m else name.split('(')[0][:60]


def main(root):
    w = csv.writer(sys.stdout)
    w.writerow(['pass', 'kernel', 'Counter_Name', 'mean', 'count', 'dur_us'])
    for d in sorted(glob.glob(os.path.join(root, '*'))):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: [0.0, 0, 0.0])
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                k = (short(row['Kernel_Name']), row['Counter_Name'])
                a = acc[k]
                a[0] += float(row['Counter_Value'])
                a[1] += 1
                if row.get('End_Timestamp') and row.get('Start_Timestamp'):
                    a[2] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
        for (kernel, counter), (tot, n, dur) in sorted(acc.items()):
            if 'pfa' in kernel or not kernel.startswith(('Cijk', 'void at', '__amd')):
                w.writerow([os.path.basename(d), kernel, counter, round(tot / n, 1), n, round(dur / n, 1)])


if __name__ == '__main__':
    main(sys.argv[1])
