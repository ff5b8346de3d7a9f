#!/usr/bin/env python3
"""Time the post-process kernels (EAW steps, variance filter, whole fpt_filter) at 1600x900 on the bench scene's gbuffer and
report them against the HBM roofline.  Algorithmic bytes per pixel: plain step 16 (img) + 16 (geo) + 4 (var) + 16 (dst) = 52;
weighted steps add the 16-B weight image (+16 B dst read in add mode)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fermat_amd as fa
from fermat_amd import scene

W, H = 1600, 900
s = scene.bathroom_standin(float(os.environ.get("DETAIL", "0.3")))
r = fa.Renderer(s, W, H, fa.default_options(5))
r.clear_gbuffer()
for i in range(4):
    r.render_pass(i)
r.synchronize()
n = W * H
ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
stream = torch.cuda.ExternalStream(r.L.fpt_stream(r.ctx))
res = {}
with torch.cuda.stream(stream):
    for _ in range(3):
        r.filter(3)
    e0, e1 = ev(), ev()
    e0.record(stream)
    K = 20
    for _ in range(K):
        r._check(r.L.fpt_filter(r.ctx, __import__("ctypes").byref(r.view), 3))
    e1.record(stream); e1.synchronize()
    ms = e0.elapsed_time(e1) / K
# per filter(): 2 x (variance 20 B/px + 1 demod step 68 + 5 plain steps 52 + 1 mod/add step 84) + 32 B copy
alg = n * (2 * (20 + 68 + 5 * 52 + 84) + 32)
res["fpt_filter_ms"] = ms
res["alg_bytes"] = alg
res["achieved_gbs"] = alg / (ms * 1e-3) / 1e9
res["frac_of_8TBs"] = res["achieved_gbs"] / 8000.0
res["pixels"] = n
print(json.dumps(res))
