"""The 8-wide compressed BVH the traversal kernels walk (fermat_amd/csrc/fpt_bvh.{h,cpp} build_wide8, fpt_trace8.hip), checked on the CPU:
an independent numpy walker that decodes the 80-byte nodes exactly as the header documents them must reach, for every ray, the triangle
the oracle's own (different) BVH reports as the closest hit -- i.e. the layout, the valid / imask / slot encodings and the outward
quantisation of the child boxes are right, whatever the kernel does with them.  (The kernel itself is checked against the oracle bit for bit
in the -m gpu tests.)"""
import ctypes as C

import numpy as np
import pytest

import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob


def build(s):
    L = fa.lib()
    nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    idx = np.ascontiguousarray(s.vertex_indices, np.int32); vtx = np.ascontiguousarray(s.vertex_data, np.float32)
    args = (C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(vtx.ctypes.data))
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), None, None, None) == 0
    nodes = np.zeros((nn.value, nw.value), np.uint32); recs = np.zeros((nr.value, 12), np.float32)
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), None) == 0
    return nodes, recs, dp.value


def decode(node):
    b = node.view(np.uint8)
    p = node[:3].view(np.float32).astype(np.float64)
    e = b[12:15].astype(np.int64) - 127
    imask = int(b[15]); child_base = int(node[4]); tri_base = int(node[5])
    q = b[32:80].reshape(6, 8).astype(np.float64)
    cell = np.ldexp(1.0, e)
    lo = p[None, :] + q[0:3].T * cell[None, :]; hi = p[None, :] + q[3:6].T * cell[None, :]
    return p, imask, child_base, tri_base, slots(node), lo, hi


def slots(node):
    """per slot (kind, first, count): kind 0 = empty, 1 = inner child (first = its node index), 2 = leaf (first = its first record).  Word 6 holds two `valid` bits per
    slot (bit 2s: a first triangle, bit 2s + 1: a second one); a node's records are packed in slot order, so a leaf's first record is tri_base + the number of valid bits
    below its pair; inner children are packed in slot order behind child_base (fpt_bvh.h BvhNode8)."""
    imask = int(node[3]) >> 24; valid = int(node[6]); child_base = int(node[4]); tri_base = int(node[5])
    assert valid >> 16 == 0 and int(node[7]) == 0
    out = []
    for s in range(8):
        pair = (valid >> (2 * s)) & 3
        assert pair in (0, 1, 3), "a leaf's valid bits are unary"
        if (imask >> s) & 1:
            assert pair == 0
            out.append((1, child_base + bin(imask & ((1 << s) - 1)).count("1"), 1))
        elif pair:
            out.append((2, tri_base + bin(valid & ((1 << (2 * s)) - 1)).count("1"), 1 if pair == 1 else 2))
        else:
            out.append((0, 0, 0))
    return out


def walk(nodes, o, d, tmin, tmax):
    """all triangle records in leaves whose (decoded) box the ray segment touches; also checks the structural invariants on the way"""
    out = []
    stack = [0]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d.astype(np.float64)
    while stack:
        ni = stack.pop()
        p, imask, child_base, tri_base, sl, lo, hi = decode(nodes[ni])
        q = nodes[ni].view(np.uint8)[32:80].reshape(6, 8)
        for s in range(8):
            kind, first, cnt = sl[s]
            if kind == 0:
                assert (q[0:3, s] == 255).all() and (q[3:6, s] == 0).all(), "an empty slot holds the inverted box no ray hits"
                continue
            t0 = (lo[s] - o) * inv; t1 = (hi[s] - o) * inv
            tn = max(np.nanmax(np.minimum(t0, t1)), tmin); tf = min(np.nanmin(np.maximum(t0, t1)), tmax)
            if tn <= tf:
                if kind == 1:
                    stack.append(first)
                else:
                    out.extend(range(first, first + cnt))
    return out


def _rays(s, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = s.bbox
    r = np.zeros(n, ob.RAY_DTYPE)
    r["origin"] = lo + rng.random((n, 3), dtype=np.float32) * (hi - lo)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::7, 0] = 0.0; d[3::11, 1] = 0.0                                 # axis-parallel components
    r["dir"] = d; r["tmax"] = 1.0e34; r["mask"] = 0
    return r


def check_tree(s, nodes, recs, depth, table, n_rays, seed):
    """the structure is a tree over all records, every triangle is referenced, and the walker reaches the oracle's closest hit for every ray"""
    ids = recs[:, 9].view(np.int32)
    n_rec = len(ids)
    assert sorted(ids.tolist()) == list(range(s.num_triangles)) and 1 <= depth <= 48
    seen_nodes = set(); seen_tris = []
    stack = [0]
    while stack:
        ni = stack.pop(); assert ni not in seen_nodes; seen_nodes.add(ni)
        p, imask, child_base, tri_base, sl, lo, hi = decode(nodes[ni])
        stack.extend(child_base + k for k in range(bin(imask).count("1")))
        for kind, first, cnt in sl:
            if kind == 2:
                seen_tris.extend(range(first, first + cnt))
    assert len(seen_nodes) == len(nodes) and sorted(seen_tris) == list(range(n_rec if s.num_triangles else 0))
    # child boxes contain their triangles (v0, v0 + e1, v0 + e2), through every level
    o = ob.OraclePT(s, 16, 16, ob.default_options(2), table, scene.DATA_DIR)
    rays = _rays(s, n_rays, seed)
    hits = o.trace(rays)
    assert (hits["triId"] >= 0).mean() > 0.5
    rec_of = {int(t): i for i, t in enumerate(ids)}
    for r, h in zip(rays, hits):
        if h["triId"] < 0:
            continue
        cand = walk(nodes, r["origin"].astype(np.float64), r["dir"], 0.0, float(h["t"]) * (1.0 + 1e-6) + 1e-9)
        assert rec_of[int(h["triId"])] in cand


@pytest.mark.parametrize("name", ["CornellBox-JP", "CornellBox-Glossy"])
def test_wide_bvh_reaches_every_closest_hit(table, name):
    s = scene.cornell_box(name)
    nodes, recs, depth = build(s)
    check_tree(s, nodes, recs, depth, table, 400, 3)


def test_reinsertion_pass_lowers_the_cost_and_keeps_the_tree_valid(table):
    """optimize_bvh2 (between the binary build and the collapse) on a scene with the mix it is there for -- room-sized triangles among small ones:
    the summed inner-node area of the binary tree falls, and the collapsed tree is still a tree over every triangle whose boxes hide no hit"""
    s = scene.bathroom_standin(0.12)
    L = fa.lib()
    st = fa.api.BvhStats(); nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    idx = np.ascontiguousarray(s.vertex_indices, np.int32); vtx = np.ascontiguousarray(s.vertex_data, np.float32)
    assert L.fpt_debug_build_bvh(C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(vtx.ctypes.data), C.byref(nn), C.byref(nr),
                                 C.byref(dp), C.byref(nw), None, None, C.byref(st)) == 0
    d = st.as_dict()
    assert s.num_triangles > 5000 and d["records"] == s.num_triangles
    assert d["optimise_iterations"] >= 1 and 0.0 < d["inner_area_after"] < 0.95 * d["inner_area_before"], d
    assert d["depth_binary"] >= d["depth"] and d["stack_need"] <= 48 and d["avg_used_slots"] >= 6.5
    nodes, recs, depth = build(s)
    assert len(nodes) == d["nodes"]            # the builder is deterministic
    check_tree(s, nodes, recs, depth, table, 250, 11)


def test_builder_bounds_the_traversal_stack_on_a_deep_chain():
    """the builder's own bound of the traversal stack (fpt_bvh_stats.stack_need, what fpt_rt_create_geometry checks against the kernel's 48 entries)
    on a mesh that makes a SAH builder peel one primitive per level; and the occupancy the SAH-optimal collapse reaches on a real scene"""
    L = fa.lib()
    n = 1500
    k = np.arange(n, dtype=np.float64); sz = 1.03 ** k; ang = k * 0.7
    P = [np.stack([np.cos(ang + d), np.sin(ang + d), k * 1e-3], 1) * np.stack([sz, sz, np.ones(n)], 1) for d in (0.0, 2.1, 4.2)]
    vtx = np.zeros((3 * n, 4), np.float32); vtx[:, :3] = np.concatenate(P)
    idx = np.zeros((n, 4), np.int32); idx[:, 0] = np.arange(n); idx[:, 1] = n + np.arange(n); idx[:, 2] = 2 * n + np.arange(n)
    st = fa.api.BvhStats(); nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert L.fpt_debug_build_bvh(C.c_uint32(n), C.c_void_p(idx.ctypes.data), C.c_uint32(3 * n), C.c_void_p(vtx.ctypes.data), C.byref(nn), C.byref(nr), C.byref(dp),
                                 C.byref(nw), None, None, C.byref(st)) == 0
    d = st.as_dict()
    assert d["records"] == n and 1 <= d["stack_need"] <= 48 and d["stack_need"] <= 2 * d["depth"] and sum(d["slot_hist"]) == d["nodes"]
    s = scene.cornell_box("CornellBox-Glossy")
    idx = np.ascontiguousarray(s.vertex_indices, np.int32); vtx = np.ascontiguousarray(s.vertex_data, np.float32)
    assert L.fpt_debug_build_bvh(C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(vtx.ctypes.data), C.byref(nn), C.byref(nr),
                                 C.byref(dp), C.byref(nw), None, None, C.byref(st)) == 0
    d = st.as_dict()
    assert d["avg_used_slots"] >= 6.0 and d["inner_children"] == d["nodes"] - 1 and d["records"] == s.num_triangles


def test_coincident_triangles_do_not_degrade_the_tree():
    """where every position costs the same the re-insertion search would string subtrees into a chain (binary depth in the hundreds): nodes that are no
    larger than their children are left alone, and a tree that still fails the kernel's stack bound is built again without the pass -- the tree handed to
    the kernel is shallow and holds every triangle once; a cluster of duplicates inside an ordinary scene does not deepen it either"""
    L = fa.lib()
    n = 60000
    vtx = np.zeros((3, 4), np.float32); vtx[1, 0] = 1.0; vtx[2, 1] = 1.0
    idx = np.zeros((n, 4), np.int32); idx[:, 1] = 1; idx[:, 2] = 2
    st = fa.api.BvhStats(); nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    args = (C.c_uint32(n), C.c_void_p(idx.ctypes.data), C.c_uint32(3), C.c_void_p(vtx.ctypes.data))
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), None, None, C.byref(st)) == 0
    d = st.as_dict()
    assert d["records"] == n and d["stack_need"] <= 24 and d["depth"] <= 12 and d["depth_binary"] <= 40, d
    recs = np.zeros((nr.value, 12), np.float32); nodes = np.zeros((nn.value, nw.value), np.uint32)
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), None) == 0
    assert sorted(recs[:, 9].view(np.int32).tolist()) == list(range(n))
    # the Cornell box with one of its triangles repeated 20 000 times
    s = scene.cornell_box("CornellBox-Glossy")
    idx = np.ascontiguousarray(np.concatenate([s.vertex_indices, np.repeat(s.vertex_indices[5:6], 20000, 0)]), np.int32); vtx = np.ascontiguousarray(s.vertex_data, np.float32)
    assert L.fpt_debug_build_bvh(C.c_uint32(len(idx)), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(vtx.ctypes.data), C.byref(nn), C.byref(nr),
                                 C.byref(dp), C.byref(nw), None, None, C.byref(st)) == 0
    d = st.as_dict()
    assert d["records"] == len(idx) and d["stack_need"] <= 30 and d["depth"] <= 16, d


@pytest.mark.parametrize("which,settings", [("bathroom_standin(0.5)", ("1", "3", "8")), ("bathroom2_standin()", ("2", "7"))])
def test_the_tree_does_not_depend_on_the_number_of_builder_threads(tmp_path, which, settings):
    """the binary build hands subtrees to worker threads and stitches them in task order, the re-insertion pass searches serially (its candidate measures are computed
    on all threads) and numbers the nodes by their pre-order rank, the collapse solves disjoint subtrees and emits the wide nodes of a level on all threads: the node
    and record arrays must be byte-identical for any number of threads (FPT_BUILD_THREADS is read when the builder starts; one process per setting).  The 1.8 M-triangle
    bench scene is the case that was NOT thread-independent until round 5: the task cut moves with the thread count there, and the pass used to break ties by the array
    position build_bvh2 left a node at"""
    import hashlib
    import os
    import subprocess
    import sys
    code = (
        "import sys, hashlib, ctypes as C, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import fermat_amd as fa\n"
        "from fermat_amd import scene\n"
        "s = scene.%s\n"
        "L = fa.lib(); nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()\n"
        "idx = np.ascontiguousarray(s.vertex_indices, np.int32); vtx = np.ascontiguousarray(s.vertex_data, np.float32)\n"
        "a = (C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(vtx.ctypes.data))\n"
        "st = fa.api.BvhStats()\n"
        "assert L.fpt_debug_build_bvh(*a, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), None, None, C.byref(st)) == 0\n"
        "nodes = np.zeros((nn.value, nw.value), np.uint32); recs = np.zeros((nr.value, 12), np.float32)\n"
        "assert L.fpt_debug_build_bvh(*a, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), None) == 0\n"
        "print(s.num_triangles, st.build_threads, hashlib.sha256(nodes.tobytes() + recs.tobytes()).hexdigest())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), which)
    seen = {}
    for threads in settings:
        env = dict(os.environ, FPT_BUILD_THREADS=threads)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        n_tri, used, digest = r.stdout.split()
        assert int(n_tri) >= 100000 and used == threads         # large enough for the sliced top nodes (>= 65 536 references) and the subtree tasks
        seen[threads] = digest
    assert len(set(seen.values())) == 1, seen


def test_refit_keeps_the_tree_valid_after_the_vertices_moved(table):
    """fpt_rt_refit_geometry's host half (fpt_debug_refit_bvh): the tree is built over a scene, a third of its vertices then move (a rigid shift of every vertex in the
    upper part of the room, a shear of the rest), and the refitted structure -- same topology, boxes and triangle records recomputed bottom-up -- is checked by the
    independent walker against the oracle's closest hits on the MOVED scene: still a tree over every triangle, and no box hides a hit.  The records hold the new edges."""
    s = scene.bathroom_standin(0.12)
    moved = scene.bathroom_standin(0.12)
    v = moved.vertex_data
    up = v[:, 1] > 0.5 * (v[:, 1].min() + v[:, 1].max())
    v[up, 0] += np.float32(1.75); v[up, 2] -= np.float32(0.5)
    v[~up, 0] += (v[~up, 1] * np.float32(0.2)).astype(np.float32)
    moved.bbox = (v[:, :3].min(0), v[:, :3].max(0))
    L = fa.lib()
    nn, nr, dp = C.c_uint32(), C.c_uint32(), C.c_uint32()
    idx = np.ascontiguousarray(s.vertex_indices, np.int32); v0 = np.ascontiguousarray(s.vertex_data, np.float32); v1 = np.ascontiguousarray(moved.vertex_data, np.float32)
    args = (C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(v0.ctypes.data), C.c_void_p(v1.ctypes.data))
    st = fa.api.BvhStats()
    assert L.fpt_debug_refit_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), None, None, C.byref(st)) == 0, L.fpt_last_error(None)
    nodes = np.zeros((nn.value, 20), np.uint32); recs = np.zeros((nr.value, 12), np.float32)
    assert L.fpt_debug_refit_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), None) == 0
    d = st.as_dict()
    assert d["records"] == s.num_triangles and 0.0 < d["seconds_refit"] < d["seconds_binary"] + d["seconds_optimise"] + d["seconds_wide"]
    # the records follow the vertices
    ids = recs[:, 9].view(np.int32)
    tri = moved.vertex_indices[ids][:, :3]
    assert np.array_equal(recs[:, 0:3], v1[tri[:, 0], :3]) and np.array_equal(recs[:, 3:6], v1[tri[:, 1], :3] - v1[tri[:, 0], :3])
    # same topology as the tree built over the unmoved scene
    n0, r0, depth0 = build(s)
    assert n0.shape == nodes.shape and np.array_equal(n0[:, 4:8], nodes[:, 4:8]) and np.array_equal(r0[:, 9].view(np.int32), ids)
    check_tree(moved, nodes, recs, dp.value, table, 300, 5)


def _soup(n, rng, spread=1.0, size=0.01):
    c = rng.random((n, 3)) * spread
    v = (c[:, None, :] + rng.standard_normal((n, 3, 3)) * size).reshape(-1, 3)
    vtx = np.zeros((3 * n, 4), np.float32); vtx[:, :3] = v
    idx = np.zeros((n, 4), np.int32); idx[:, :3] = np.arange(3 * n).reshape(n, 3)
    return idx, vtx


def _build_raw(idx, vtx, want_arrays=False):
    L = fa.lib()
    nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = fa.api.BvhStats()
    idx = np.ascontiguousarray(idx, np.int32); vtx = np.ascontiguousarray(vtx, np.float32)
    args = (C.c_uint32(len(idx)), C.c_void_p(idx.ctypes.data), C.c_uint32(len(vtx)), C.c_void_p(vtx.ctypes.data))
    rc = L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), None, None, C.byref(st))
    if rc != 0 or not want_arrays:
        return rc, nn.value, nr.value, st.as_dict(), None, None
    nodes = np.zeros((nn.value, nw.value), np.uint32); recs = np.zeros((nr.value, 12), np.float32)
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), None) == 0
    return rc, nn.value, nr.value, st.as_dict(), nodes, recs


def test_builder_on_edge_sizes_and_degenerate_inputs():
    """The in-place builder of round 5's second half at its seams: the empty scene, one triangle, the sizes around the 32-reference switch between the two split
    finders, the sizes around the hand-out of subtrees to threads and around the ranges that all threads partition together; coincident triangles (no axis is
    live: object medians), a scene of extent 1e19 (areas overflow fp32), half the triangles in a point-like cluster (deep SAH peeling), a vertex index out of range.
    Every triangle must come out in exactly one record, and the stack bound must fit the kernel's."""
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 3, 5, 31, 32, 33, 34, 65, 100, 4097, 20000, 32768, 70000):
        idx, vtx = _soup(n, rng) if n else (np.zeros((0, 4), np.int32), np.zeros((1, 4), np.float32))
        rc, nn, nr, st, nodes, recs = _build_raw(idx, vtx, want_arrays=True)
        assert rc == 0 and nr == max(n, 1) and st["stack_need"] <= 48
        if n:
            assert np.array_equal(np.sort(recs[:, 9].view(np.int32)), np.arange(n))          # tri_id of every record: a permutation
            assert (recs[:, 11] > 0).all()                                                    # the box clause's tolerance is in every record
    for n in (7, 1000, 70000):
        idx, vtx = _soup(n, rng, spread=0.0, size=0.0); vtx[:, :3] = np.tile(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), (n, 1))
        rc, nn, nr, st, _, _ = _build_raw(idx, vtx)
        assert rc == 0 and nr == n and st["stack_need"] <= 48
    idx, vtx = _soup(50000, rng, spread=1e19, size=1e15)
    assert _build_raw(idx, vtx)[0] == 0
    idx, vtx = _soup(100000, rng, spread=1.0, size=1e-3); vtx[:150000, :3] *= 1e-6
    rc, nn, nr, st, _, _ = _build_raw(idx, vtx)
    assert rc == 0 and nr == 100000 and st["stack_need"] <= 48
    idx, vtx = _soup(70000, rng); idx[69999, 1] = 10 ** 7
    assert _build_raw(idx, vtx)[0] != 0


def check_containment(nodes, recs):
    """The whole structure, not only what some rays meet: every child's decoded box (node origin + 8-bit grid) contains everything below it -- for a leaf the boxes of
    its triangles padded by (nearly) the builder's 4 x the record's tolerance word, for an inner child the content of its whole subtree -- and every node's inner
    children sit where child_base + rank says.  Nodes are numbered breadth-first (children behind their parent), so one backward sweep does it."""
    N = len(nodes)
    b = nodes.view(np.uint8).reshape(N, 80)
    p = nodes[:, :3].view(np.float32).astype(np.float64)                         # (N, 3)
    cell = np.ldexp(1.0, b[:, 12:15].astype(np.int64) - 127)                      # (N, 3)
    q = b[:, 32:80].reshape(N, 6, 8).astype(np.float64)                          # qlo.xyz, qhi.xyz per slot
    lo = p[:, None, :] + np.transpose(q[:, 0:3, :], (0, 2, 1)) * cell[:, None, :]  # (N, 8, 3)
    hi = p[:, None, :] + np.transpose(q[:, 3:6, :], (0, 2, 1)) * cell[:, None, :]
    imask = b[:, 15].astype(np.int64)
    v0 = recs[:, 0:3].astype(np.float64); v1 = v0 + recs[:, 3:6].astype(np.float64); v2 = v0 + recs[:, 6:9].astype(np.float64)
    pad = 3.9 * recs[:, 11].astype(np.float64)                                    # the builder pads by 2e-6 (...) = 4 x this word, up to an ulp
    tlo = np.minimum(np.minimum(v0, v1), v2) - pad[:, None]; thi = np.maximum(np.maximum(v0, v1), v2) + pad[:, None]
    clo = np.full((N, 3), np.inf); chi = np.full((N, 3), -np.inf)                # content of each node's subtree
    n_checked = 0
    for n in range(N - 1, -1, -1):
        for s, (kind, first, cnt) in enumerate(slots(nodes[n])):
            if kind == 0:
                continue
            if kind == 1:
                c = first
                assert n < c < N, "inner children come behind their parent"
                a, z = clo[c], chi[c]
            else:
                a, z = tlo[first:first + cnt].min(0), thi[first:first + cnt].max(0)
            assert (lo[n, s] <= a).all() and (hi[n, s] >= z).all(), "node %d slot %d: the decoded box does not contain its content" % (n, s)
            clo[n] = np.minimum(clo[n], a); chi[n] = np.maximum(chi[n], z)
            n_checked += 1
    return n_checked


def test_every_box_of_the_structure_contains_what_is_below_it(table):
    """conservative quantisation and padding checked over the WHOLE tree (check_containment), for a built tree, for a refitted one (vertices moved by up to 5 % of the
    scene: the topology is stale, the boxes must not be), and for a triangle soup with coincident and degenerate triangles"""
    for s in (scene.cornell_box("CornellBox-Glossy"), scene.bathroom_standin(0.2)):
        nodes, recs, depth = build(s)
        assert check_containment(nodes, recs) >= len(nodes)
        rng = np.random.default_rng(5)
        moved = s.vertex_data.copy()
        ext = float(np.max(np.asarray(s.bbox[1]) - np.asarray(s.bbox[0])))
        moved[:, :3] += (rng.standard_normal((len(moved), 3)) * 0.05 * ext).astype(np.float32)
        L = fa.lib()
        nn, nr, dp = C.c_uint32(), C.c_uint32(), C.c_uint32()
        idx = np.ascontiguousarray(s.vertex_indices, np.int32); v0 = np.ascontiguousarray(s.vertex_data, np.float32); v1 = np.ascontiguousarray(moved, np.float32)
        a = (C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(v0.ctypes.data), C.c_void_p(v1.ctypes.data))
        rn = np.zeros((len(nodes), 20), np.uint32); rr = np.zeros((len(recs), 12), np.float32)
        assert L.fpt_debug_refit_bvh(*a, C.byref(nn), C.byref(nr), C.byref(dp), C.c_void_p(rn.ctypes.data), C.c_void_p(rr.ctypes.data), None) == 0
        assert nn.value == len(nodes) and nr.value == len(recs)
        assert np.array_equal(rn[:, 4:8], nodes[:, 4:8])                              # topology, slots and leaf layout untouched
        assert check_containment(rn, rr) >= len(nodes)
    rng = np.random.default_rng(9)
    idx, vtx = _soup(3000, rng, spread=1.0, size=0.05)
    vtx[:300, :3] = np.tile(np.float32([[0.5, 0.5, 0.5], [0.6, 0.5, 0.5], [0.5, 0.6, 0.5]]), (100, 1))          # a hundred coincident triangles
    vtx[300:330, :3] = np.float32([0.25, 0.25, 0.25])                                                         # ten triangles collapsed to a point
    vtx[330:360, :3] = np.tile(np.float32([[0.1, 0.1, 0.1], [0.9, 0.9, 0.9], [0.5, 0.5, 0.5]]), (10, 1))         # ten collinear ones
    rc, nn, nr, st, nodes, recs = _build_raw(idx, vtx, want_arrays=True)
    assert rc == 0 and check_containment(nodes, recs) >= len(nodes)
