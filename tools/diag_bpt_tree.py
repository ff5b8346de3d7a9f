#!/usr/bin/env python3
"""diagnostic (GPU box): the BPT frame (-sc 0, 1600x900, 2 passes, water_caustic stand-in) of the library given by FPT_LIB_PATH, saved to gpurun_out/<tag>.npy"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
tag = sys.argv[1]
s = scene.water_caustic_standin()
table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
r = fa.Renderer(s, 1600, 900, fa.default_options(9), table=table, gbuffer=False, bpt_options=fa.default_bpt_options(9, single_connection=0))
for i in range(2):
    r.bpt_render(i)
fb = r.framebuffer()
np.save("gpurun_out/%s.npy" % tag, fb[:6])
print(tag, r.bvh_stats()["nodes"], [float(fb[c][:, :3].sum()) for c in range(6)])
