#!/usr/bin/env python3
"""Run a few passes of a named scene (for rocprofv3 timelines): run_passes.py <scene> <W> <H> <L> <passes>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
name, W, H, L, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
s = scene.bathroom_standin(float(name.split(":")[1])) if name.startswith("standin") else scene.cornell_box(name)
r = fa.Renderer(s, W, H, fa.default_options(L), gbuffer=False)
for i in range(n):
    r.render_pass(i)
r.synchronize()
print("done", s.num_triangles)
