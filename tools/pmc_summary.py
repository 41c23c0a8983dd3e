#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv (pmc_counter_collection.csv): mean counter value per (kernel, grid size)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'][:70] + ' grid=' + r['Grid_Size'] + ' wg=' + r['Workgroup_Size'] + ' lds=' + r['LDS_Block_Size']
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, d in agg.items():
    if flt not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:32s} {sum(v) / len(v):16.0f}   (n={len(v)})')
