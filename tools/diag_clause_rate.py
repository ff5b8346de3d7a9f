#!/usr/bin/env python3
"""Does the intersector's box clause (oracle/o_bvh.h intersect_tri, fpt_trace.hip intersect_record) change the result of any ray of a REAL pass?  It must not: it exists
for "hits" whose computed t is noise, not for true ones.  Bit-exact parity between kernel and oracle cannot answer this -- both apply the same clause; the clause's
first form (ray point against barycentric point) passed every parity test while rejecting 1-2 % of the true hits on the bench scene's sliver triangles.

    python tools/diag_clause_rate.py [bathroom2|standin|water|testball|cornell] [WxH]

Renders one oracle pass at low resolution, captures the closest-hit queues of bounces 0..3 and traces them again with the clause switched off (CPU only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fermat_amd import scene                 # noqa: E402
from oracle import binding as ob             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bathroom2"
res = tuple(int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "400x225").split("x"))
s = {"bathroom2": scene.bathroom2_standin, "standin": scene.bathroom_standin, "water": scene.water_caustic_standin, "testball": scene.testball_room,
     "cornell": lambda: scene.cornell_box("CornellBox-Glossy")}[name]()
table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
pt = ob.OraclePT(s, res[0], res[1], ob.default_options(9), table, scene.DATA_DIR)
L = ob.lib()
total = changed = 0
for b in range(4):
    pt.set_capture(b); pt.render_pass(0)
    cap = pt.captured()
    ray = np.ascontiguousarray(cap["ray"]).view(np.float32).reshape(-1, 8)
    rays = np.zeros(len(ray), ob.RAY_DTYPE)
    rays["origin"] = ray[:, 0:3]; rays["mask"] = ray[:, 3].view(np.uint32); rays["dir"] = ray[:, 4:7]; rays["tmax"] = ray[:, 7]
    with_clause = pt.trace(rays, n_threads=16)
    L.orc_debug_set_box_clause(0)
    without = pt.trace(rays, n_threads=16)
    L.orc_debug_set_box_clause(1)
    diff = (with_clause["triId"] != without["triId"]) | (with_clause["t"].view(np.uint32) != without["t"].view(np.uint32))
    print("%s bounce %d: %7d rays, %7d hits, results the clause changes: %d" % (name, b, len(rays), int((without["triId"] >= 0).sum()), int(diff.sum())))
    total += len(rays); changed += int(diff.sum())
print("%s: %d of %d rays changed by the clause" % (name, changed, total))
