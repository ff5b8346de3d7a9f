#!/usr/bin/env python3
"""GPU box: what the fixed part of a traversal launch is (VERDICT r5 task 7).  The rays the path tracer really traces at bounce 3 of a pass of the bench scene (captured from
the product's own queues) are traced again in launches of n = 64 ... all rays, each launch alone on an idle chip: wall time of fpt_rt_trace + synchronise, best of 9.
If a launch of ONE wave of these rays already takes a fifth of a millisecond, the tail of a launch is the dependent chain of its longest rays -- ~90 node steps, each a
fetch from L2 / Infinity Cache -- and no arrangement of launch boundaries (persistent kernels, grid barriers) can shorten it; only overlapping phases could.
    python tools/diag_tail.py [bathroom2|standin]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                 # noqa: E402
import fermat_amd as fa                      # noqa: E402
from fermat_amd import scene                 # noqa: E402
from fermat_amd.api import RAY_DTYPE         # noqa: E402

s = scene.bathroom2_standin() if (len(sys.argv) < 2 or sys.argv[1] == "bathroom2") else scene.bathroom_standin()
r = fa.Renderer(s, 1600, 900, fa.default_options(9), gbuffer=False)
r.set_capture(3); r.render_pass(0); cap = r.captured(); r.set_capture(-1)
rays = np.array(cap["rays"], RAY_DTYPE, copy=True)
rays["mask"] = np.float32(1e-3).view(np.uint32); rays["tmax"] = 1e8          # a renderer's queue keeps bookkeeping in the .w words: the scattered rays' interval is (1e-3, 1e8)
rng = np.random.default_rng(0); rng.shuffle(rays)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1)).to(r.dev); d_h = torch.zeros(len(rays) * 4, dtype=torch.float32, device=r.dev); torch.cuda.synchronize()
L = fa.lib()
_, cnt = r.trace(rays, counted=True)
print("%d closest-hit rays of bounce 3 of one 1600x900 pass; %.2f node steps + %.2f triangle tests per ray" % (len(rays), cnt.nodes_visited / cnt.rays, cnt.tris_tested / cnt.rays))
n = 64
while True:
    n = min(n, len(rays))
    best = 1e9
    for _ in range(9):
        t = time.perf_counter()
        assert L.fpt_rt_trace(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr())) == 0
        r.synchronize()
        best = min(best, time.perf_counter() - t)
    print("  launch of %8d rays (%6d waves): %.3f ms   %.1f ns per ray" % (n, (n + 63) // 64, best * 1e3, best * 1e9 / n), flush=True)
    if n == len(rays):
        break
    n *= 8
# an empty launch for the floor of the measurement itself
best = 1e9
for _ in range(9):
    t = time.perf_counter(); L.fpt_rt_trace(r.ctx, C.c_uint32(1), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr())); r.synchronize(); best = min(best, time.perf_counter() - t)
print("  launch of 1 ray: %.3f ms (launch + synchronise floor of this measurement)" % (best * 1e3))
