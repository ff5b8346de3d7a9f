#!/usr/bin/env python3
"""How long does ONE wave of rays take?  (serial latency of the traversal loop)"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
import torch
s = scene.bathroom_standin(1.0)
r = fa.Renderer(s, 64, 64, fa.default_options(2), gbuffer=False)
rng = np.random.default_rng(1)
lo, hi = s.bbox
def rays(n):
    a = np.zeros(n, fa.RAY_DTYPE)
    a["origin"] = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32); a["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    a["mask"] = np.float32(1e-3).view(np.uint32); a["tmax"] = 1e8
    return a
for n in (1, 64, 4096, 65536, 1 << 20):
    ra = rays(n)
    d_r = torch.from_numpy(ra.view(np.float32).reshape(-1)).cuda(); d_h = torch.zeros(n * 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    hits, cnt = r.trace(ra, counted=True)
    L = r.L
    for _ in range(3):
        L.fpt_rt_trace(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr()))
    r.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        L.fpt_rt_trace(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr()))
    r.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("n=%8d  %.1f us/launch   nodes/ray %.1f tris/ray %.1f   %.2f Grays/s" % (n, dt * 1e6, cnt.nodes_visited / n, cnt.tris_tested / n, n / dt / 1e9))
