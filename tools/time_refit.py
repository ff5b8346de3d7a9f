#!/usr/bin/env python3
"""host timing of fpt_rt_refit_geometry's builder half against a fresh build (no GPU): python tools/time_refit.py [scene function]"""
import sys, time, numpy as np, ctypes as C
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
s = getattr(scene, sys.argv[1] if len(sys.argv) > 1 else "bathroom2_standin")()
L = fa.lib()
nn, nr, dp = C.c_uint32(), C.c_uint32(), C.c_uint32()
idx = np.ascontiguousarray(s.vertex_indices, np.int32); v0 = np.ascontiguousarray(s.vertex_data, np.float32)
v1 = v0.copy(); up = v1[:, 1] > np.median(v1[:, 1]); v1[up, 0] += np.float32(0.5)
st = fa.api.BvhStats()
t = time.time()
assert L.fpt_debug_refit_bvh(C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(v0.ctypes.data), C.c_void_p(v1.ctypes.data),
                             C.byref(nn), C.byref(nr), C.byref(dp), None, None, C.byref(st)) == 0
d = st.as_dict()
print("%d triangles, %d threads: build %.3f s (binary %.3f + re-insertion %.3f + collapse %.3f), refit %.3f s" %
      (s.num_triangles, d["build_threads"], d["seconds_binary"] + d["seconds_optimise"] + d["seconds_wide"], d["seconds_binary"], d["seconds_optimise"], d["seconds_wide"], d["seconds_refit"]))
