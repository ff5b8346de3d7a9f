#!/usr/bin/env python3
"""Print the per-dispatch timeline of pass #k from a rocprofv3 rocpd database: timeline.py <dir> [k]"""
import sqlite3, os, sys
d = sys.argv[1]; k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = [os.path.join(d, x) for x in os.listdir(d) if x.endswith(".db")][0]
cur = sqlite3.connect(f).cursor()
rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "rescale_kernel" in r[0]]
i0, i1 = starts[k], (starts[k + 1] if k + 1 < len(starts) else len(rows))
t0 = rows[i0][1]; prev = None
for r in rows[i0:i1]:
    nm = r[0].replace("fpt::", "").split("(")[0][:44]
    if "fillBuffer" in nm: continue
    print("%-46s start %8.1f us  dur %7.1f us  gap %5.1f" % (nm, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev) / 1e3 if prev else 0.0))
    prev = r[2]
print("pass wall %.1f us" % ((rows[i1 - 1][2] - t0) / 1e3))
