"""Builder evaluation without a GPU: builds the 8-wide BVH of a bench scene with the product's host builder (fpt_debug_build_bvh), prints the
node-occupancy histogram and walks the rays of an oracle-rendered low-resolution pass (closest-hit queues of bounces 0..3, in queue order)
through tools/bvh_walk.cpp, a CPU model of the kernel's traversal order.  Reports node steps / triangle tests per ray and the number of
64-lane lock-step iterations (the VALU cost model: a wave pays for an iteration while any lane is busy).

    python tools/bvh_stats.py [--workload bathroom2|standin|testball-room|cornell] [--res 400x225] [--what-if]

--what-if also prices the alternatives DESIGN.md 5 quotes, all on the same tree and rays: the stack policies (bvh8_walk_policy), two rays per lane
(bvh8_walk_pairs), a pool of rays per wave with its state in LDS (bvh8_walk_pool, bvh8_walk_pool_pipelined), fp32 child boxes instead of the 8-bit grid (bvh8_walk_set_exact), a strictly nearest-first walk (bvh8_walk_sorted) and distances kept
with the stacked groups (bvh8_walk_cull).
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
    if name == "bathroom2":
        return scene.bathroom2_standin()
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
    return with_slot_bytes(nodes), recs, dp.value, dt


def with_slot_bytes(nodes):
    """tools/bvh_walk.cpp models the traversal ORDER and reads a per-slot byte for it (inner child: 0x20 | 24 + slot; leaf: unary triangle count << 5 | offset of its first
    record behind tri_base -- the node's words 6..7 until round 5).  Since round 6 the node carries two `valid` bits per slot instead (fpt_bvh.h BvhNode8, word 6): the bytes
    are derived from them here, in a copy, so that the model walks the tree the kernel walks."""
    out = nodes.copy()
    by = out.view(np.uint8).reshape(len(out), 80)
    imask = (nodes[:, 3] >> 24).astype(np.int64); valid = nodes[:, 6].astype(np.int64)
    offset = np.zeros(len(nodes), np.int64)
    for sl in range(8):
        pair = (valid >> (2 * sl)) & 3
        inner = ((imask >> sl) & 1) == 1
        by[:, 24 + sl] = np.where(inner, 0x20 | (24 + sl), np.where(pair != 0, (pair << 5) | offset, 0)).astype(np.uint8)
        offset += (pair & 1) + (pair >> 1)
    return out


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
    ap.add_argument("--what-if", action="store_true")
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
    if a.what_if:
        what_if(W, nodes, recs, rays)


def exact_child_boxes(nodes, recs):
    """fp32 boxes of every child slot (bottom-up over the breadth-first node array): leaves from their triangle records, inner children from their own children"""
    N = len(nodes)
    by = nodes.view(np.uint8).reshape(N, 80)
    meta = by[:, 24:32]
    v0 = recs[:, 0:3]; v1 = v0 + recs[:, 3:6]; v2 = v0 + recs[:, 6:9]
    tlo = np.minimum(np.minimum(v0, v1), v2); thi = np.maximum(np.maximum(v0, v1), v2)
    mag = np.abs(np.concatenate([v0, v1, v2])).max()
    pad = (2.0e-6 * (np.maximum(np.abs(tlo), np.abs(thi)).max(1) + mag))[:, None]          # the builder's padding
    tlo = tlo - pad; thi = thi + pad
    exact = np.zeros((N, 8, 6), np.float32); exact[:, :, :3] = 3e38; exact[:, :, 3:] = -3e38
    own_lo = np.full((N, 3), 3e38, np.float32); own_hi = np.full((N, 3), -3e38, np.float32)
    cnt = {1: 1, 3: 2, 7: 3}
    for i in range(N - 1, -1, -1):
        rel = 0
        cb = int(nodes[i, 4]); tb = int(nodes[i, 5])
        for sl in range(8):
            m = int(meta[i, sl])
            if m == 0:
                continue
            if (m >> 5) == 1 and (m & 0x1F) >= 24:
                c = cb + rel; rel += 1
                exact[i, sl, :3] = own_lo[c]; exact[i, sl, 3:] = own_hi[c]
            else:
                a = tb + (m & 0x1F); n = cnt[m >> 5]
                exact[i, sl, :3] = tlo[a:a + n].min(0); exact[i, sl, 3:] = thi[a:a + n].max(0)
            own_lo[i] = np.minimum(own_lo[i], exact[i, sl, :3]); own_hi[i] = np.maximum(own_hi[i], exact[i, sl, 3:])
    return np.ascontiguousarray(exact)


def what_if(W, nodes, recs, rays):
    nr = sum(len(r) for r in rays)
    pn, pr = C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data)

    def walk():
        tot = np.zeros(9)
        for r in rays:
            out = (C.c_uint64 * 9)()
            W.bvh8_walk(pn, pr, C.c_void_p(r.ctypes.data), C.c_uint32(len(r)), 0, out, None, None)
            tot += np.array(list(out), np.float64)
        return tot, (228 * tot[6] + 100 * tot[7] + 30 * tot[5]) / nr

    rays = [np.ascontiguousarray(r) for r in rays]
    print("  what-if (node steps / triangle tests / wave iterations per ray, modelled wave instructions per ray):")
    for pol, name in ((0, "pop only with nothing in hand (round 2)"), (1, "the kernel: a node group is taken while triangles are in hand"), (2, "... and parked triangles while only nodes are in hand")):
        W.bvh8_walk_policy(pol)
        t, c = walk()
        ls = (C.c_uint64 * 5)(); W.bvh8_walk_lane_stats(ls); ls = np.array(list(ls), np.float64); ls /= ls.sum()
        print("    stack policy %d  %5.2f %5.2f %.4f  %6.1f   lanes of the last bounce: no ray %.2f, node only %.2f, triangle only %.2f, both %.2f   (%s)" %
              (pol, t[0] / nr, t[1] / nr, t[5] / nr, c, ls[0], ls[1], ls[2], ls[3], name))
    W.bvh8_walk_policy(1)
    for cull, name in ((1, "one entry distance per stacked group"), (2, "one per child"), (3, "one per group + leaves behind the nearest inner child parked")):
        W.bvh8_walk_cull(cull)
        t, c = walk()
        print("    distances %d     %5.2f %5.2f %.4f  %6.1f   (%s)" % (cull, t[0] / nr, t[1] / nr, t[5] / nr, c, name))
    W.bvh8_walk_cull(0)
    exact = exact_child_boxes(nodes, recs)
    W.bvh8_walk_set_exact(C.c_void_p(exact.ctypes.data), pn)
    t, c = walk()
    W.bvh8_walk_set_exact(None, pn)
    print("    fp32 child boxes %5.2f %5.2f %.4f  %6.1f" % (t[0] / nr, t[1] / nr, t[5] / nr, c))
    ts = np.zeros(2)
    for r in rays:
        o2 = (C.c_uint64 * 2)()
        W.bvh8_walk_sorted(pn, pr, C.c_void_p(r.ctypes.data), C.c_uint32(len(r)), o2)
        ts += np.array(list(o2), np.float64)
    print("    nearest-first    %5.2f %5.2f   (every hit child, leaf or inner, by entry distance; stacked entries behind the hit dropped)" % (ts[0] / nr, ts[1] / nr))
    t8 = np.zeros(7)
    for r in rays:
        o7 = (C.c_uint64 * 7)()
        W.bvh8_walk_lanes8(pn, pr, C.c_void_p(r.ctypes.data), C.c_uint32(len(r)), o7)
        t8 += np.array(list(o7), np.float64)
    # prices (VALU instructions per half, from the node step's ISA and tools/micro/issue_model.hip): node half = set-up 27 + one child per lane 21 + 3-stage sorting
    # network over the row (6 compare-exchanges x ~5: DPP move, compare, two selects) 30 + push of the sorted children and pop 22 = 100; triangle half = the 100 of
    # fpt-MT + a 3-step row reduction of the best hit 12 + group bookkeeping 10 = 122.  The kernel as built pays 228 / 100 per half for 64 rays at the lane utilisation it reaches.
    print("    8 lanes per ray, 8 rays per wave: %.2f node steps %.2f triangle tests per ray; per ray %.3f node halves + %.3f triangle halves of a wave (ray slots active %.2f / %.2f of 8)"
          " -> %.1f wave instructions per ray at 100 / 122 per half; rays in flight per SIMD at 8 waves: 64 instead of 512" %
          (t8[3] / nr, t8[4] / nr, t8[1] / nr, t8[2] / nr, t8[5] / (8 * max(t8[1], 1)), t8[6] / (8 * max(t8[2], 1)), (100 * t8[1] + 122 * t8[2]) / nr))
    for refill in (32, 64):
        t = np.zeros(5)
        for r in rays:
            out = (C.c_uint64 * 5)()
            W.bvh8_walk_pairs(pn, pr, C.c_void_p(r.ctypes.data), C.c_uint32(len(r)), 0, refill, out)
            t += np.array(list(out), np.float64)
        print("    two rays per lane, refill at %d idle slots: %.4f iterations per ray, %.1f / %.1f wave instructions with 30 / 80 extra per iteration" %
              (refill, t[0] / nr, (228 * t[1] + 100 * t[2] + 60 * t[0]) / nr, (228 * t[1] + 100 * t[2] + 110 * t[0]) / nr))

    for pool, refill in ((64, 16), (96, 16), (128, 32), (192, 64), (256, 64)):
        t = np.zeros(6)
        for r in rays:
            out = (C.c_uint64 * 6)()
            W.bvh8_walk_pool(pn, pr, C.c_void_p(r.ctypes.data), C.c_uint32(len(r)), 0, pool, refill, out)
            t += np.array(list(out), np.float64)
        print("    pool of %3d rays per wave in LDS, refill at %d empty slots: node halves %.4f / triangle halves %.4f per ray (lanes filled %.2f / %.2f); wave instructions per ray with "
              "40 / 80 / 120 extra per half (state in and out of LDS, compaction): %.1f / %.1f / %.1f" %
              (pool, refill, t[0] / nr, t[1] / nr, t[4] / (64 * t[0]), t[5] / (64 * t[1]), *[((228 + x) * t[0] + (100 + x) * t[1]) / nr for x in (40, 80, 120)]))
        tp = np.zeros(6)
        for r in rays:
            out = (C.c_uint64 * 6)()
            W.bvh8_walk_pool_pipelined(pn, pr, C.c_void_p(r.ctypes.data), C.c_uint32(len(r)), 0, pool, refill, out)
            tp += np.array(list(out), np.float64)
        print("      pipelined (next batch chosen from the rays outside the current one): node halves %.4f / triangle halves %.4f per ray (lanes filled %.2f / %.2f): %.1f / %.1f / %.1f" %
              (tp[0] / nr, tp[1] / nr, tp[4] / (64 * max(tp[0], 1)), tp[5] / (64 * max(tp[1], 1)), *[((228 + x) * tp[0] + (100 + x) * tp[1]) / nr for x in (40, 80, 120)]))


if __name__ == "__main__":
    main()
