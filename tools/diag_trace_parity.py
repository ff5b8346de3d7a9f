#!/usr/bin/env python3
"""diagnostic (GPU box): fpt_rt_trace / fpt_rt_trace_shadow against the oracle's traces on a bench scene, a few million rays (random, camera, and rays that
start ON surfaces like scattered rays do); prints every disagreement with both answers.  python tools/diag_trace_parity.py [scene function] [n_rays]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob
name = sys.argv[1] if len(sys.argv) > 1 else "water_caustic_standin"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
s = getattr(scene, name)()
table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
r = fa.Renderer(s, 16, 16, fa.default_options(2), table=table)
o = ob.OraclePT(s, 16, 16, ob.default_options(2), table, scene.DATA_DIR)
o.set_trace_threads(16)
rng = np.random.default_rng(3)
lo, hi = s.bbox
rays = np.zeros(n, fa.RAY_DTYPE)
rays["origin"] = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays["dir"] = d.astype(np.float32); rays["tmax"] = 1e34
hg = r.trace(rays); ho = o.trace(rays)
# second generation: start on the surfaces the first rays hit (like scattered rays: tmin 1e-3), random directions
hit = ho["triId"] >= 0
r2 = np.zeros(int(hit.sum()), fa.RAY_DTYPE)
r2["origin"] = (rays["origin"][hit] + rays["dir"][hit] * ho["t"][hit, None]).astype(np.float32)
d2 = rng.normal(size=(len(r2), 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
r2["dir"] = d2.astype(np.float32); r2["tmax"] = 1e8; r2["mask"] = np.float32(1e-3).view(np.uint32)
hg2 = r.trace(r2); ho2 = o.trace(r2)
# shadow rays between surface points
sh = np.zeros(len(r2) - 1, fa.RAY_DTYPE)
sh["origin"] = r2["origin"][:-1]; sh["dir"] = r2["origin"][1:] - r2["origin"][:-1]; sh["tmax"] = 0.9999; sh["mask"] = 2
sg = r.trace(sh, shadow=True); so = o.trace(sh, shadow=True)
for tag, a, b, rr in (("closest", hg, ho, rays), ("closest from surfaces", hg2, ho2, r2)):
    bad = np.where((a["triId"] != b["triId"]) | (a["t"].view(np.uint32) != b["t"].view(np.uint32)) | (a["u"].view(np.uint32) != b["u"].view(np.uint32)) | (a["v"].view(np.uint32) != b["v"].view(np.uint32)))[0]
    print("%s: %d rays, %d disagree" % (tag, len(rr), len(bad)))
    for i in bad[:10]:
        print("   ray", i, rr[i], "hip", a[i], "oracle", b[i])
bad = np.where(sg["t"] != so["t"])[0]
print("any-hit: %d rays, %d disagree" % (len(sh), len(bad)))
for i in bad[:10]:
    print("   ray", i, sh[i], "hip", sg[i], "oracle", so[i])
