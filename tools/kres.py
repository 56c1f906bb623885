#!/usr/bin/env python3
"""tools/kres.py <remarks.txt> [filter...] -- registers / scratch / occupancy per kernel out of `hipcc -Rpass-analysis=kernel-resource-usage`"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2:]
rows, cur = [], None
for line in txt.split('\n'):
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                     ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)'), ('sgpr', r' SGPRs: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
seen = set()
for r, n in zip(rows, names):
    n = re.sub(r'\(.*', '', n).replace('void tfft::', '').replace('tfft::', '')
    if n in seen or (flt and not any(f in n for f in flt)):
        continue
    seen.add(n)
    print(f"{n:46s} vgpr {r.get('vgpr'):4d} agpr {r.get('agpr', 0):3d} sgpr {r.get('sgpr', 0):3d} scratch {r.get('scratch'):4d} occ {r.get('occ')}")
