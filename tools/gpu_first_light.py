#!/usr/bin/env python3
"""Bring-up script: render CornellBox with the HIP path and the oracle, print agreement statistics."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob

res = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
name = sys.argv[5] if len(sys.argv) > 5 else "CornellBox-JP"
s = scene.cornell_box(name)
table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
t = time.time()
r = fa.Renderer(s, res[0], res[1], fa.default_options(L))
print("renderer init %.2fs" % (time.time() - t), r.bvh_info())
o = ob.OraclePT(s, res[0], res[1], ob.default_options(L), table, scene.DATA_DIR)

# sequence
sh_g, sa_g = r.sequence(0); sh_o, sa_o = o.sequence(0)
print("shifts equal:", np.array_equal(sh_g, sh_o), " samples equal:", np.array_equal(sa_g, sa_o))
lg, lo = r.lights(), o.lights()
print("vpls equal:", np.array_equal(lg["vpls"], lo["vpls"]), "norm", lg["norm"], lo["norm"], "cdf eq", np.array_equal(lg["mesh_cdf"], lo["mesh_cdf"]))

r.set_profiling(True)
for b in range(0, min(L, 3)):
    r.set_capture(b); o.set_capture(b)
    r.clear_framebuffer(); o.fb[:] = 0
    r.render_pass(0, sync=True); o.render_pass(0)
    cg = r.captured(); co = o.captured()
    print("bounce", b, "queue sizes", len(cg["rays"]), len(co))
    if len(cg["rays"]) == len(co) and len(co):
        ig = np.argsort(cg["pixel_info"] & 0x7FFFFFF, kind="stable"); io = np.argsort(co["pixel_info"] & 0x7FFFFFF, kind="stable")
        rays_eq = np.array_equal(cg["rays"][ig].view(np.uint32), co["ray"][io].view(np.uint32))
        hits_eq = np.array_equal(cg["hits"][ig].view(np.uint32), co["hit"][io].view(np.uint32))
        w_eq = np.array_equal(cg["weights"][ig].view(np.uint32), co["weight"][io].view(np.uint32))
        pi_eq = np.array_equal(cg["pixel_info"][ig], co["pixel_info"][io])
        cone_eq = np.array_equal(cg["cones"][ig].view(np.uint32), co["cone"][io].view(np.uint32))
        print("   rays", rays_eq, "hits", hits_eq, "weights", w_eq, "pixel_info", pi_eq, "cones", cone_eq)
        if not hits_eq:
            hg = cg["hits"][ig]; ho = co["hit"][io]
            bad = np.nonzero((hg["triId"] != ho["triId"]) | (hg["t"] != ho["t"]) | (hg["u"] != ho["u"]) | (hg["v"] != ho["v"]))[0]
            print("   hit mismatches:", len(bad), [(hg[k].tolist(), ho[k].tolist()) for k in bad[:5]])
        if not w_eq:
            wg = cg["weights"][ig]; wo = co["weight"][io]
            bad = np.nonzero((wg != wo).any(1))[0]
            print("   weight mismatches:", len(bad), [(wg[k].tolist(), wo[k].tolist()) for k in bad[:5]])
        if not rays_eq:
            rg = cg["rays"][ig]; ro = co["ray"][io]
            bad = np.nonzero((rg.view(np.uint32).reshape(-1, 8) != ro.view(np.uint32).reshape(-1, 8)).any(1))[0]
            print("   ray mismatches:", len(bad), [(rg[k].tolist(), ro[k].tolist()) for k in bad[:3]])
r.set_capture(-1); o.set_capture(-1)
r.clear_framebuffer(); o.fb[:] = 0
for i in range(passes):
    r.render_pass(i, sync=True); o.render_pass(i)
    st = r.stats()
    print("pass", i, "gpu in", list(st.in_size[:st.n_bounces]), "sh", list(st.shadow_size[:st.n_bounces]))
    print("         orc", o.stats().tolist())
fg = r.framebuffer(); fo = o.fb
for c, nm in enumerate(["DIFFUSE_C", "DIFFUSE_A", "SPECULAR_C", "SPECULAR_A", "DIRECT_C", "COMPOSITED_C", "FILTERED_C", "LUMINANCE"]):
    d = fg[c] - fo[c]
    rmse = float(np.sqrt((d[:, :3].astype(np.float64) ** 2).sum(1).mean()))
    print("%-13s bit-equal %-5s  rmse %.3e  max|d| %.3e  mean %.4f" % (nm, np.array_equal(fg[c].view(np.uint32), fo[c].view(np.uint32)), rmse, float(np.abs(d).max()), float(fo[c][:, :3].mean())))
print("rgba equal:", np.array_equal(r.to_rgba(), o.to_rgba()))
st = r.stats()
print("timers ms: primary %.3f path %.3f shadow %.3f shade %.3f" % (st.primary_rt_ms, st.path_rt_ms, st.shadow_rt_ms, st.path_shade_ms))
