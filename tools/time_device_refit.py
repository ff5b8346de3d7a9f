#!/usr/bin/env python3
"""GPU box: what fpt_rt_refit_geometry costs on the device (round 6, fpt_build.hip) on the bench scene -- wall time of the call (it returns when the tree is in place: the
error flag is read back) over a few repetitions, vertices already resident -- next to the host refit it replaced (fpt_debug_refit_bvh's seconds_refit + what the two
copies of the mesh and the tree cost then), and the traversal cost of the refitted tree against the built one.
    python tools/time_device_refit.py [bathroom2|standin]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fermat_amd as fa                      # noqa: E402
from fermat_amd import scene                 # noqa: E402
import torch                                 # noqa: E402

s = scene.bathroom2_standin() if (len(sys.argv) < 2 or sys.argv[1] == "bathroom2") else scene.bathroom_standin()
r = fa.Renderer(s, 64, 64, fa.default_options(3))
L = fa.lib()
v0 = np.ascontiguousarray(s.vertex_data, np.float32)
v1 = v0.copy(); up = v1[:, 1] > np.median(v1[:, 1]); v1[up, 0] += np.float32(0.05)
d1 = torch.from_numpy(v1).to(r.dev); d0 = torch.from_numpy(v0).to(r.dev); torch.cuda.synchronize()
args = lambda: (r.ctx, C.c_uint32(s.num_triangles), C.c_void_p(r.d_vi.data_ptr()), C.c_uint32(s.num_vertices), C.c_void_p(r.d_vd.data_ptr()))
times = []
for it in range(6):
    r.d_vd.copy_((d1 if it % 2 == 0 else d0).reshape(r.d_vd.shape)); torch.cuda.synchronize()
    t = time.perf_counter()
    assert L.fpt_rt_refit_geometry(*args()) == 0, L.fpt_last_error(r.ctx)
    times.append((time.perf_counter() - t) * 1e3)
st = r.bvh_stats()
print("%d triangles, %d wide nodes: fpt_rt_refit_geometry on the device, wall ms per call: %s (first call allocates its scratch); seconds_refit of the last %.4f" %
      (s.num_triangles, st["nodes"], " ".join("%.3f" % t for t in times), st["seconds_refit"]))
