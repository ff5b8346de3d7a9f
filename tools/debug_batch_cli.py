import os, subprocess, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "fermat_amd", "bin", "fermat_hip")
d = os.path.join(scene.DATA_DIR, "scenes", "CornellBox")
table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
s = scene.cornell_box("CornellBox-Glossy")
o = ob.OraclePT(s, 64, 48, ob.default_options(5), table, scene.DATA_DIR)
for i in range(5): o.render_pass(i)
want = o.to_rgba().reshape(48, 64, 4)[..., :3].astype(np.int32)
for args in (["-passes", "4", "-batch", "3"], ["-passes", "4", "-batch", "5"], ["-passes", "4", "-batch", "2"], ["-passes", "4"]):
    out = "/tmp/dbg_" + "_".join(a.strip("-") for a in args)
    r = subprocess.run([exe, "-i", os.path.join(d, "CornellBox-Glossy.obj"), "-c", os.path.join(d, "camera-frontal.txt"), "-r", "64", "48", "-pt", "-bounces", "4"] + args + ["-o", out], capture_output=True, text=True)
    got = (scene.load_tga(out + ".tga")[..., :3] * 255.0 + 0.5).astype(np.int32)
    print(args, r.returncode, "max diff", np.abs(got - want).max(), "mean", got.mean(), want.mean())
r = fa.Renderer(s, 64, 48, fa.default_options(5), table=table)
r.set_batch(3); r.render_batch(0, 3); r.render_batch(3, 2)
fb = r.framebuffer()[5]
print("python 3+2: rmse", float(np.sqrt(((fb[:, :3] - o.fb[5][:, :3]) ** 2).sum(1).mean())), fb[:, :3].mean(), o.fb[5][:, :3].mean())

# the host mirror driven call by call (fpt_host_context_*): frame mean after every render(i)
import ctypes as C
L = fa.lib()
L.fpt_host_scene_load.restype = C.c_void_p; L.fpt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p]
L.fpt_host_context_create.restype = C.c_void_p; L.fpt_host_last_error.restype = C.c_char_p
class SA(C.Structure):
    _fields_ = [("mesh", fa.api.MeshView), ("textures", C.c_void_p), ("num_textures", C.c_uint32), ("dir_lights", C.c_void_p),
                ("dir_lights_count", C.c_uint32), ("glossy_reflectance", C.c_void_p), ("camera", fa.api.Camera), ("samples_dir", C.c_char_p)]
h = L.fpt_host_scene_load(os.path.join(d, "CornellBox-Glossy.obj").encode(), scene.DATA_DIR.encode())
cam = fa.api.Camera()
L.fpt_host_load_camera(os.path.join(d, "camera-frontal.txt").encode(), C.byref(cam))
sa = SA(); L.fpt_host_scene_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
L.fpt_host_scene_arrays(h, C.byref(cam), C.byref(sa))
for extra in (["-batch", "3"], ["-batch", "2"], []):
    args = ["fermat", "-r", "64", "48", "-pt", "-bounces", "4", "-passes", "4"] + extra
    argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
    L.fpt_host_context_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    c = L.fpt_host_context_create(len(args), argv, C.byref(sa))
    assert c, L.fpt_host_last_error()
    L.fpt_host_context_render.argtypes = [C.c_void_p, C.c_uint32]; L.fpt_host_context_download.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    buf = np.zeros((64 * 48, 4), np.float32)
    o2 = ob.OraclePT(s, 64, 48, ob.default_options(5), table, scene.DATA_DIR)
    for i in range(5):
        assert L.fpt_host_context_render(c, i) == 0, L.fpt_host_last_error()
        L.fpt_host_context_download(c, 5, C.c_void_p(buf.ctypes.data))
        o2.render_pass(i)
        print(extra, "after render(%d): mean %.6f (oracle %.6f) nan %d" % (i, buf[:, :3].mean(), o2.fb[5][:, :3].mean(), np.isnan(buf).sum()))
    L.fpt_host_context_destroy.argtypes = [C.c_void_p]; L.fpt_host_context_destroy(c)
