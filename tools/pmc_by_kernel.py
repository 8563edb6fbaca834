#!/usr/bin/env python3
"""Sum the rows of a rocprofv3 counter_collection.csv per (kernel, counter): counter totals and per-dispatch averages."""
import collections
import csv
import sys

tot = collections.defaultdict(float)
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get('Kernel_Name', r.get('Kernel Name', '?'))
    name = name.replace('(anonymous namespace)::', '').split('(')[0][:40]
    tot[(name, r['Counter_Name'])] += float(r['Counter_Value'])
    cnt[name].add(r.get('Dispatch_Id', r.get('Dispatch_ID')))
names = sorted({k[0] for k in tot}, key=lambda n: -tot.get((n, 'SQ_BUSY_CYCLES'), 0))
ctrs = sorted({k[1] for k in tot})
print('%-40s %8s ' % ('kernel', 'launches') + ' '.join('%22s' % c[-22:] for c in ctrs))
for n in names[:25]:
    print('%-40s %8d ' % (n, len(cnt[n])) + ' '.join('%22.4g' % tot.get((n, c), 0) for c in ctrs))
