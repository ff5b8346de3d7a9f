import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
import fermat_amd as fa
from fermat_amd import scene
s = scene.bathroom_standin(1.0)
r = fa.Renderer(s, 1600, 900, fa.default_options(9), gbuffer=False); r.set_batch(64)
t = time.time()
for i in range(0, 4096, 64):
    r.render_batch(i, 64)
r.synchronize(); dt = time.time() - t
fb = r.framebuffer()
print("4096 passes in %.2f s (%.0f Msample/s); finite %s; mean %.5f; max %.3f; min %.3g" % (dt, 1600*900*4096/dt/1e6, bool(np.isfinite(fb).all()), fb[5][:, :3].mean(), fb[5][:, :3].max(), fb[5][:, :3].min()))
a = fb[5][:, :3].copy()
for i in range(4096, 4096 + 64, 64):
    r.render_batch(i, 64)
b = r.framebuffer()[5][:, :3]
print("rmse(4160 vs 4096 spp) = %.3e" % float(np.sqrt(((a - b) ** 2).sum(1).mean())))
r.close()
