#!/usr/bin/env python3
"""Traversal kernel in isolation: times fpt_rt_trace (closest hit) and fpt_rt_trace_shadow (any hit) on the REAL ray populations of the bench
workload -- the in-queues of bounces 0, 1 and 3 of a 4-pass batch, captured through fpt_pt_set_capture -- for whichever product library
FPT_LIB_PATH selects, and prints one JSON line with ms per launch, Mray/s, nodes / triangles per ray and a checksum of the hits (equal
across variants: results do not depend on the acceleration structure).  Kernel experiments compare variants with this before bench.py."""
import argparse, json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="standin"); ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--passes", type=int, default=4); ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--bounces", default="0,1,3")
    a = ap.parse_args()
    import torch
    import fermat_amd as fa
    from fermat_amd import scene
    s = scene.testball_room() if a.workload == "testball-room" else scene.bathroom_standin(a.detail)
    W, H, L = 1600, 900, 9
    r = fa.Renderer(s, W, H, fa.default_options(L), gbuffer=False)
    r.set_batch(a.passes)
    out = {"lib": os.path.basename(fa.lib_path()), "workload": a.workload, "triangles": int(s.num_triangles), "bvh": r.bvh_info(), "bounces": {}}
    for b in [int(x) for x in a.bounces.split(",")]:
        r.clear_framebuffer()
        r.set_capture(b)
        r.render_batch(0, a.passes, sync=True)
        cap = r.captured()
        rays = np.ascontiguousarray(cap["rays"])
        n = len(rays)
        d_r = torch.from_numpy(rays.view(np.float32).reshape(-1)).to(r.dev)
        d_h = torch.zeros(n * 4, dtype=torch.float32, device=r.dev)
        torch.cuda.synchronize(r.dev)
        res = {"rays": n}
        for shadow in (False, True):
            if shadow:
                # the same population as shadow rays: segments from the origin to 90 % of the hit distance... keep it simple: same rays, any-hit
                pass
            fn = r.L.fpt_rt_trace_shadow if shadow else r.L.fpt_rt_trace
            r._check(fn(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr()))); r.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                r._check(fn(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr())))
            r.synchronize()
            ms = (time.perf_counter() - t0) / a.reps * 1e3
            hits = d_h.cpu().numpy()
            cnt = fa.api.TraceCounters()
            r._check(r.L.fpt_rt_trace_counted(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr()), C.c_int(1 if shadow else 0), C.byref(cnt)))
            res["any" if shadow else "closest"] = {"ms": ms, "mray_s": n / ms / 1e3, "nodes_per_ray": cnt.nodes_visited / max(1, n), "tris_per_ray": cnt.tris_tested / max(1, n),
                                                  "crc": zlib.crc32(hits.tobytes()) & 0xFFFFFFFF}
        out["bounces"][str(b)] = res
    r.set_capture(-1)
    print(json.dumps(out))
    r.close()


if __name__ == "__main__":
    main()
