"""Builder evaluation without a GPU: builds the 8-wide BVH of a bench scene with the product's host builder (fpt_debug_build_bvh), prints the
node-occupancy histogram and walks the rays of an oracle-rendered low-resolution pass (closest-hit queues of bounces 0..3, in queue order)
through tools/bvh_walk.cpp, a CPU model of the kernel's traversal order.  Reports node steps / triangle tests per ray and the number of
64-lane lock-step iterations (the VALU cost model: a wave pays for an iteration while any lane is busy).

    python tools/bvh_stats.py [--workload standin|testball-room|cornell] [--res 400x225]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fermat_amd as fa                      # noqa: E402
from fermat_amd import scene                 # noqa: E402


def walker():
    out = os.path.join(ROOT, "tools", "_build"); os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libbvhwalk.so"); src = os.path.join(ROOT, "tools", "bvh_walk.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


def load_scene(name):
    if name == "standin":
        return scene.bathroom_standin()
    if name == "testball-room":
        return scene.testball_room()
    return scene.cornell_box("CornellBox-Glossy")


def capture_rays(s, name, res, bounces=4):
    cache = "/tmp/bvh_stats_rays_%s_%dx%d.npz" % (name, res[0], res[1])
    if os.path.exists(cache):
        z = np.load(cache); return [z["b%d" % b] for b in range(bounces)]
    from oracle import binding as ob
    table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
    out = []
    for b in range(bounces):
        o = ob.OraclePT(s, res[0], res[1], ob.default_options(9), table, scene.DATA_DIR)
        o.set_trace_threads(8)
        o.set_capture(b); o.render_pass(0)
        out.append(np.ascontiguousarray(o.captured()["ray"]))
        del o
    np.savez(cache, **{"b%d" % b: r for b, r in enumerate(out)})
    return out


def build(s):
    L = fa.lib()
    nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    idx = np.ascontiguousarray(s.vertex_indices, np.int32); vtx = np.ascontiguousarray(s.vertex_data, np.float32)
    args = (C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(vtx.ctypes.data))
    t = time.time()
    st = fa.api.BvhStats()
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), None, None, C.byref(st)) == 0, L.fpt_last_error(None)
    dt = time.time() - t
    print("  builder:", st.as_dict())
    nodes = np.zeros((nn.value, nw.value), np.uint32); recs = np.zeros((nr.value, 12), np.float32)
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), None) == 0
    return nodes, recs, dp.value, dt


def occupancy(nodes):
    meta = nodes.view(np.uint8).reshape(len(nodes), 80)[:, 24:32]
    used = (meta != 0).sum(1)
    inner = ((meta >> 5) == 1) & ((meta & 0x1F) >= 24)
    leaf = (meta != 0) & ~inner
    ntri = np.where(leaf, np.array([0, 1, 0, 2, 0, 0, 0, 3])[meta >> 5], 0).sum()
    return used, int(inner.sum()), int(leaf.sum()), int(ntri)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="standin")
    ap.add_argument("--res", default="400x225")
    a = ap.parse_args()
    res = tuple(int(x) for x in a.res.split("x"))
    s = load_scene(a.workload)
    rays = capture_rays(s, a.workload, res)
    nodes, recs, depth, dt = build(s)
    used, n_inner, n_leaf, n_tri = occupancy(nodes)
    hist = np.bincount(used, minlength=9)
    print("%s: %d triangles, %d wide nodes, %d records, depth %d, build %.2f s" % (a.workload, s.num_triangles, len(nodes), len(recs), depth, dt))
    print("  used slots / 8: avg %.2f   histogram 0..8: %s   inner %d leaf %d (%.2f tris/leaf)" % (used.mean(), hist.tolist(), n_inner, n_leaf, n_tri / max(1, n_leaf)))
    W = walker()
    tot = np.zeros(9, np.float64); nr = 0
    for b, r in enumerate(rays):
        r = np.ascontiguousarray(r)
        out = (C.c_uint64 * 9)()
        W.bvh8_walk(C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), C.c_void_p(r.ctypes.data), C.c_uint32(len(r)), 0, out, None, None)
        o = np.array(list(out), np.float64); n = len(r)
        print("  bounce %d: %7d rays  nodes/ray %6.2f  tris/ray %5.2f  wave-iters/ray %6.3f (lane util %.2f)  with refill %6.3f  max stack %d" %
              (b, n, o[0] / n, o[1] / n, o[2] / n, o[3] / (64 * o[2]), o[5] / n, int(o[4])))
        tot[:4] += o[:4]; tot[5:] += o[5:]; nr += n
    print("  all     : %7d rays  nodes/ray %6.2f  tris/ray %5.2f  wave-iters/ray %6.3f (lane util %.2f)  with refill %6.3f" %
          (nr, tot[0] / nr, tot[1] / nr, tot[2] / nr, tot[3] / (64 * tot[2]), tot[5] / nr))
    print("  triangle tests an fp32 per-triangle box would have culled: %.2f per ray (%.0f %%)" % (tot[8] / nr, 100.0 * tot[8] / max(1.0, tot[1])))
    print("  model   : wave node-iterations/ray %.4f, triangle-iterations/ray %.4f -> VALU instructions per ray (228 / 100 / 30 per iteration) %.1f" %
          (tot[6] / nr, tot[7] / nr, (228 * tot[6] + 100 * tot[7] + 30 * tot[5]) / nr))


if __name__ == "__main__":
    main()
