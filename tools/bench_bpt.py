#!/usr/bin/env python3
"""Throughput of the bidirectional path tracer at 1600x900 on the bench scene (BASELINE config 5's renderer; water_caustic.obj is
missing from the reference checkout, so the bathroom2 stand-in is used and named)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fermat_amd as fa
from fermat_amd import scene
W, H = 1600, 900
L = int(os.environ.get("BPT_L", "9"))
s = scene.bathroom_standin(float(os.environ.get("DETAIL", "1.0")))
r = fa.Renderer(s, W, H, fa.default_options(L), gbuffer=False, bpt_options=fa.default_bpt_options(L))
for i in range(3):
    r.bpt_render(i)
r.synchronize()
K = int(os.environ.get("STEPS", "16"))
t0 = time.perf_counter()
for i in range(3, 3 + K):
    r.bpt_render(i)
r.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"renderer": "bpt -sc 0", "workload": "bathroom2-standin 1600x900, max path length %d, %d triangles" % (L, s.num_triangles),
                  "ms_per_pass": dt / K * 1e3, "msample_per_s": W * H * K / dt / 1e6}))
