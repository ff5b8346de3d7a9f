#!/usr/bin/env python3
"""Launch time of the traversal kernel against the number of rays (fpt_rt_trace on prefixes of one captured ray population, bounce 1 of the
bench frame): t(n) = a + b n separates the fixed cost of a launch (ramp + tail of the persistent waves) from the per-ray cost."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=16); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--bounce", type=int, default=1)
    a = ap.parse_args()
    import torch
    import fermat_amd as fa
    from fermat_amd import scene
    s = scene.bathroom_standin(1.0)
    r = fa.Renderer(s, 1600, 900, fa.default_options(9), gbuffer=False)
    r.set_batch(a.passes)
    r.set_capture(a.bounce)
    r.render_batch(0, a.passes, sync=True)
    rays = np.ascontiguousarray(r.captured()["rays"])
    r.set_capture(-1)
    n_all = len(rays)
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1)).to(r.dev)
    d_h = torch.zeros(n_all * 4, dtype=torch.float32, device=r.dev)
    out = {"rays_captured": n_all, "points": []}
    n = 1 << 12
    sizes = []
    while n < n_all:
        sizes.append(n); n *= 2
    sizes.append(n_all)
    for n in sizes:
        fn = r.L.fpt_rt_trace
        r._check(fn(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr()))); r.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            r._check(fn(r.ctx, C.c_uint32(n), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_h.data_ptr())))
        r.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        out["points"].append({"n": n, "ms": round(ms, 4), "ns_per_ray": round(ms * 1e6 / n, 3)})
    print(json.dumps(out))
    r.close()


if __name__ == "__main__":
    main()
