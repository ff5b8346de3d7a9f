#!/usr/bin/env python3
"""What RenderingContext::update_model costs end to end on the bench scene (bathroom2 stand-in, 1.82 M triangles, 1600x900) through the C++ mirror
(fpt_host_context_update_model): refit = 0 -- the acceleration structure is built again -- and refit = 1 -- boxes and triangle records follow the vertices --,
each followed by HipPathTracer::update_scene (flush of pending passes, emitter tables).  GPU box."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fermat_amd as fa                      # noqa: E402
from fermat_amd import scene                 # noqa: E402

L = fa.lib()
L.fpt_host_context_create.restype = C.c_void_p
L.fpt_host_last_error.restype = C.c_char_p


class SceneArrays(C.Structure):
    _fields_ = [("mesh", fa.api.MeshView), ("textures", C.c_void_p), ("num_textures", C.c_uint32), ("dir_lights", C.c_void_p),
                ("dir_lights_count", C.c_uint32), ("glossy_reflectance", C.c_void_p), ("camera", fa.api.Camera), ("samples_dir", C.c_char_p)]


s = scene.bathroom2_standin() if (len(sys.argv) < 2 or sys.argv[1] == "bathroom2") else scene.bathroom_standin()
table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
sa = SceneArrays()
sa.mesh.num_triangles = s.num_triangles; sa.mesh.num_vertices = s.num_vertices; sa.mesh.num_materials = len(s.materials)
sa.mesh.vertex_indices = s.vertex_indices.ctypes.data; sa.mesh.vertex_data = s.vertex_data.ctypes.data
sa.mesh.material_indices = s.material_indices.ctypes.data; sa.mesh.materials = s.materials.ctypes.data
if getattr(s, "texture_indices_comp", None) is not None:
    sa.mesh.texture_indices_comp = s.texture_indices_comp.ctypes.data
sa.mesh.tex_bias = (C.c_float * 2)(*s.tex_bias); sa.mesh.tex_scale = (C.c_float * 2)(*s.tex_scale)
sa.glossy_reflectance = table.ctypes.data
cam = s.camera
sa.camera.eye = (C.c_float * 3)(*cam[0:3]); sa.camera.aim = (C.c_float * 3)(*cam[3:6]); sa.camera.up = (C.c_float * 3)(*cam[6:9])
sa.camera.dx = (C.c_float * 3)(*cam[9:12]); sa.camera.fov = float(cam[12])
sa.samples_dir = scene.DATA_DIR.encode()
args = [b"fermat", b"-pt", b"-r", b"1600", b"900", b"-bounces", b"8"]
argv = (C.c_char_p * len(args))(*args)
t0 = time.time()
h = L.fpt_host_context_create(C.c_int(len(args)), argv, C.byref(sa))
assert h, L.fpt_host_last_error()
h = C.c_void_p(h)
print("%d triangles: context created (scene upload, build, emitter tables, first init) in %.3f s" % (s.num_triangles, time.time() - t0))
out = np.zeros((1600 * 900, 4), np.float32)
inst = 0
def passes(n):
    global inst
    for _ in range(n):
        assert L.fpt_host_context_render(h, C.c_uint32(inst)) == 0, L.fpt_host_last_error()
        inst += 1
    assert L.fpt_host_context_download(h, C.c_uint32(5), C.c_void_p(out.ctypes.data)) == 0
passes(4)
moved = np.array(s.vertex_data, np.float32, copy=True)
# which vertices belong to emitting triangles (the emitter tables follow only those: fpt_mesh_lights_update)
emissive = np.array([np.any(np.asarray(m["emissive"][:3]) > 0) for m in s.materials])
lit = np.zeros(len(moved), bool); lit[np.unique(s.vertex_indices[emissive[s.material_indices], :3])] = True
for what in ("everything moves (emitters too: the emitter tables are rebuilt)", "everything but the emitters moves (the emitter tables are kept)"):
  print(what)
  for rep in range(2):
    for refit in (1, 0):
        moved[~lit if "but" in what else slice(None), 0] += np.float32(0.001)
        t0 = time.time()
        assert L.fpt_host_context_update_model(h, C.c_void_p(moved.ctypes.data), C.c_int(refit)) == 0, L.fpt_host_last_error()
        t1 = time.time()
        passes(2)
        print("  update_model(refit = %d): %.4f s; the two passes after it + download: %.3f s" % (refit, t1 - t0, time.time() - t1))
L.fpt_host_context_destroy(h)
