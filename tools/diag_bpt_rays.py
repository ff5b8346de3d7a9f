#!/usr/bin/env python3
"""diagnostic (GPU box): every ray the ORACLE's bidirectional tracer traces for ONE pixel / light path of the water_caustic stand-in (-sc 0, L = 9, passes 0 and 1),
traced again by the HIP library (FPT_LIB_PATH chooses the build) as closest-hit and as any-hit rays; prints the rays whose answers differ.
    python tools/diag_bpt_rays.py <pixel> [scene function]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob
pixel = int(sys.argv[1]); W, H, L = 1600, 900, 9
s = getattr(scene, sys.argv[2] if len(sys.argv) > 2 else "water_caustic_standin")()
table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
o.bpt_init(ob.default_bpt_options(L, single_connection=0), scene.DATA_DIR)
OL = ob.lib()
OL.orc_pt_log_rays(o.h, 1)
px = np.array([pixel], np.uint32)
for i in range(2):
    OL.orc_bpt_render_pixels(o.h, C.c_uint32(i), C.c_void_p(px.ctypes.data), C.c_uint32(1))
n = 100000
rays = np.zeros(n, ob.RAY_DTYPE); hits = np.zeros(n, ob.HIT_DTYPE); kind = np.zeros(n, np.uint32)
OL.orc_pt_get_logged_rays.restype = C.c_uint32
n = OL.orc_pt_get_logged_rays(o.h, C.c_void_p(rays.ctypes.data), C.c_void_p(hits.ctypes.data), C.c_void_p(kind.ctypes.data), C.c_uint32(n))
rays, hits = rays[:n], hits[:n]
print("oracle traced", n, "rays for pixel", pixel)
r = fa.Renderer(s, 16, 16, fa.default_options(2), table=table)
g = rays.view(fa.RAY_DTYPE)
hc = r.trace(g)
sh = g.copy()
hs = r.trace(sh, shadow=True)
for i in range(n):
    same = hc["triId"][i] == hits["triId"][i] and hc["t"][i].view(np.uint32) == hits["t"][i].view(np.uint32)
    occl_o = hits["t"][i] > 0; occl_h = hs["t"][i] > 0
    if not same or occl_o != occl_h:
        print("ray %d  origin %s tmin %r dir %s tmax %r\n    oracle hit %s\n    hip closest %s   hip any-hit occluded %s" %
              (i, rays["origin"][i], rays["mask"][i:i + 1].view(np.float32)[0], rays["dir"][i], rays["tmax"][i], hits[i], hc[i], bool(occl_h)))
print("done")
