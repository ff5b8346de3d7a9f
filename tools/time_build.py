import sys, time, numpy as np, ctypes as C, hashlib
sys.path.insert(0,'/root/repo')
import fermat_amd as fa
from fermat_amd import scene
s = getattr(scene, sys.argv[1] if len(sys.argv)>1 else 'bathroom2_standin')()
L = fa.lib()
nn, nr, dp, nw = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
idx = np.ascontiguousarray(s.vertex_indices, np.int32); vtx = np.ascontiguousarray(s.vertex_data, np.float32)
args = (C.c_uint32(s.num_triangles), C.c_void_p(idx.ctypes.data), C.c_uint32(s.num_vertices), C.c_void_p(vtx.ctypes.data))
st = fa.api.BvhStats()
for rep in range(2):
    t=time.time()
    assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), None, None, C.byref(st)) == 0
    dt=time.time()-t
    d=st.as_dict()
    print("total %.3f  binary %.3f opt %.3f wide %.3f  its %d area %.3f->%.3f nodes %d sah_wide %.3f threads %d" % (dt, d['seconds_binary'], d['seconds_optimise'], d['seconds_wide'], d['optimise_iterations'], d['inner_area_before'], d['inner_area_after'], d['nodes'], d['sah_cost_wide'], d['build_threads']))
nodes = np.zeros((nn.value, nw.value), np.uint32); recs = np.zeros((nr.value, 12), np.float32)
assert L.fpt_debug_build_bvh(*args, C.byref(nn), C.byref(nr), C.byref(dp), C.byref(nw), C.c_void_p(nodes.ctypes.data), C.c_void_p(recs.ctypes.data), None) == 0
print("tree hash", hashlib.sha256(nodes.tobytes()+recs.tobytes()).hexdigest()[:16])
