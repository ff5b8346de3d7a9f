#!/usr/bin/env python3
"""batched (passes-in-flight) render vs sequential render: agreement statistics"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
s = scene.cornell_box("CornellBox-Glossy")
a = fa.Renderer(s, 160, 120, fa.default_options(6))
b = fa.Renderer(s, 160, 120, fa.default_options(6))
b.set_batch(4)
for i in range(8):
    a.render_pass(i)
b.render_batch(0, 4); b.render_batch(4, 4)
fa_, fb_ = a.framebuffer(), b.framebuffer()
for c in range(8):
    d = fa_[c] - fb_[c]
    print(c, "xyz rmse %.3e max %.3e | w max %.3e (ref max %.3e)" % (np.sqrt((d[:, :3].astype(np.float64) ** 2).sum(1).mean()), np.abs(d[:, :3]).max(), np.abs(d[:, 3]).max(), np.abs(fa_[c][:, 3]).max()))
print("gbuffer equal:", np.array_equal(a.gb_tri.cpu().numpy(), b.gb_tri.cpu().numpy()), np.array_equal(a.gb_geo.cpu().numpy(), b.gb_geo.cpu().numpy()))
ta, tb = a.gb_tri.cpu().numpy(), b.gb_tri.cpu().numpy()
bad = np.nonzero(ta != tb)[0]
print("tri mismatches", len(bad), bad[:10], ta[bad[:10]], tb[bad[:10]])
ga, gb = a.gb_geo.cpu().numpy(), b.gb_geo.cpu().numpy()
bad = np.nonzero((ga != gb).any(1))[0]
print("geo mismatches", len(bad), ga[bad[:3]], gb[bad[:3]])
c = fa.Renderer(s, 160, 120, fa.default_options(6))
c.render_pass(7)
print("pass-7-only gbuffer == batched:", np.array_equal(c.gb_tri.cpu().numpy(), tb), np.array_equal(c.gb_geo.cpu().numpy(), gb))
