#!/usr/bin/env python3
"""Per-launch times of one batch of a bench workload (profiling level 2: HIP events around every launch, fpt_pt_launch_list): where a launch chain's time
goes, launch by launch, and how it changes with the passes in flight.
    python tools/diag_launches.py [--batch 20] [--workload standin|testball-room]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import fermat_amd as fa
from fermat_amd import scene
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, nargs="+", default=[20]); ap.add_argument("--workload", default="standin")
a = ap.parse_args()
s = {"standin": lambda: scene.bathroom_standin(1.0), "bathroom2": scene.bathroom2_standin, "testball-room": scene.testball_room}[a.workload]()
names = {0: "trace0", 1: "mixed", 2: "shadow", 3: "shade"}
for batch in a.batch:
    r = fa.Renderer(s, 1600, 900, fa.default_options(9), gbuffer=False)
    if batch > 1:
        r.set_batch(batch)
    run = (lambda i: r.render_batch(i, batch)) if batch > 1 else (lambda i: r.render_pass(i))
    run(0); r.synchronize()
    r.set_profiling(2)
    run(batch); r.synchronize()
    ll = r.launch_list()
    r.set_profiling(0)
    r.set_counting(True); run(2 * batch); r.synchronize(); c, sh = r.trace_counters(); r.set_counting(False)
    print("batch %d: %d launches, trace %.3f ms shade %.3f ms; rays traced closest %d shadow %d" % (batch, len(ll), sum(m for b, m in ll if b != 3), sum(m for b, m in ll if b == 3), c.rays, sh.rays))
    print("   " + "  ".join("%s %.3f" % (names[b], m) for b, m in ll))
    r.close()
