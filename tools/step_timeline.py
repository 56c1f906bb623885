#!/usr/bin/env python3
"""tools/step_timeline.py <rocpd results.db> [anchor substring] [occurrence] -- the launches of one timed step out of a rocprofv3 kernel trace, in time order"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_gather_bits"
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = list(db.execute("select name,start,end,stream_id,grid_x,grid_y,grid_z,workgroup_x,workgroup_y,workgroup_z from kernels order by start"))
def short(n):
    n = re.sub(r'^void ', '', n); n = n.replace('tfft::', ''); n = re.sub(r'\(.*', '', n); return n[:58]
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
i = idx[min(occ, len(idx) - 2)]; j = idx[min(occ, len(idx) - 2) + 1]
t0 = rows[i][1]
for n, s, e, st, gx, gy, gz, wx, wy, wz in rows[i:j]:
    print("%9.1f %8.1f s%d %-58s wgs %d x %d thr" % ((s - t0) / 1e3, (e - s) / 1e3, st, short(n), (gx // wx) * (gy // wy) * (gz // wz), wx * wy * wz))
