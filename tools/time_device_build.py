#!/usr/bin/env python3
"""GPU box: fpt_rt_create_geometry in its two modes on the bench scene -- quality (host: binned SAH + re-insertion + collapse, mesh copied both ways) and fast (device:
Morton radix tree + the same collapse, fpt_build_lbvh.hip) -- wall time of the call, the trees' shapes, and what each tree costs to traverse (closest-hit launch over
captured-like random rays: ms and node steps / triangle tests per ray).   python tools/time_device_build.py [bathroom2|standin|testball|water|standin4]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fermat_amd as fa                      # noqa: E402
from fermat_amd import scene                 # noqa: E402
from fermat_amd.api import RAY_DTYPE         # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "bathroom2"
s = {"bathroom2": scene.bathroom2_standin, "standin": scene.bathroom_standin, "testball": scene.testball_room, "water": scene.water_caustic_standin,
     "standin4": lambda: scene.bathroom_standin(4.0)}[which]()          # standin4: 12.5 M triangles
r = fa.Renderer(s, 64, 64, fa.default_options(3))
rng = np.random.default_rng(1)
lo, hi = np.asarray(s.bbox[0]), np.asarray(s.bbox[1])
rays = np.zeros(2000000, RAY_DTYPE)
rays["origin"] = (lo + rng.random((len(rays), 3)) * (hi - lo)).astype(np.float32)
d = rng.standard_normal((len(rays), 3)).astype(np.float32); rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
rays["tmax"] = 1e34
ref = None
print("%s: %d triangles" % (which, s.num_triangles), flush=True)
for mode, name in ((0, "quality (host)"), (1, "fast (device)"), (1, "fast (device)")):
    r.set_build_mode(mode)
    t = time.perf_counter(); r.rebuild_geometry(); dt = time.perf_counter() - t
    st = r.bvh_stats()
    h, cnt = r.trace(rays, counted=True)
    t = time.perf_counter()
    for _ in range(3):
        h2 = r.trace(rays)
    tr = (time.perf_counter() - t) / 3
    if ref is None:
        ref = h
    same = np.array_equal(h["triId"], ref["triId"]) and np.array_equal(h["t"].view(np.uint32), ref["t"].view(np.uint32))
    print("%-16s create_geometry %8.2f ms   %7d wide nodes, depth %2d, stack bound %2d   2 M random rays: %.2f node steps + %.2f triangle tests per ray, trace call (incl. upload) %.1f ms   hits equal the first build's: %s"
          % (name, dt * 1e3, st["nodes"], st["depth"], st["stack_need"], cnt.nodes_visited / cnt.rays, cnt.tris_tested / cnt.rays, tr * 1e3, same), flush=True)
