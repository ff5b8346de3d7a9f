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
