#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel mean duration, PMC means, HBM bytes per launch
(FETCH_SIZE is doubled for gfx950 as MI355X_MICROARCH.md prescribes; both counters are in KiB... see units note)."""
import csv, sys, collections, os, json
d = sys.argv[1]
def short(n):
    n = n.replace('tfft::', '').replace('void ', '')
    return n.split('(')[0][:48]
stats = {}
for r in csv.DictReader(open(os.path.join(d, 'trace_kernel_stats.csv'))):
    stats[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write'):
    p = os.path.join(d, f + '_counter_collection.csv')
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        pmc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, (calls, us, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    if pct < 0.3: continue
    row = {'calls': calls, 'avg_us': round(us, 1), 'pct': pct}
    for c, v in pmc.get(k, {}).items():
        row[c] = round(sum(v) / len(v), 1)
    out[k] = row
    print(k.ljust(42), row)
json.dump(out, open(os.path.join(d, 'summary.json'), 'w'), indent=1)
