#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (counter_collection.csv)."""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = re.sub(r'\(.*', '', r['Kernel_Name'])[:70]
    key = (name, r['Grid_Size'], r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'))
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    n = len(next(iter(c.values())))
    print(k, 'dispatches', n)
    for cn, v in c.items():
        print(f'    {cn:32s} avg {sum(v)/len(v):16.1f}')
