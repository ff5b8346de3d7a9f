#!/usr/bin/env python3
"""Per-launch times of one batch of the headline workload (profiling level 2), with and without straggler carry-over: where a chain's time goes, launch by launch.
    python tools/diag_launches.py [--batch 20] [--handoff 16] [--delay 2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fermat_amd as fa
from fermat_amd import scene
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=20); ap.add_argument("--handoff", type=int, default=16); ap.add_argument("--delay", type=int, default=2)
ap.add_argument("--workload", default="standin")
a = ap.parse_args()
s = scene.bathroom_standin(1.0) if a.workload == "standin" else scene.testball_room()
names = {0: "trace0", 1: "mixed", 2: "shadow", 3: "shade"}
for handoff in (0, a.handoff):
    r = fa.Renderer(s, 1600, 900, fa.default_options(9), gbuffer=False)
    r.set_batch(a.batch)
    r.set_carry_over(handoff, a.delay)
    r.render_batch(0, a.batch); r.synchronize()
    r.set_profiling(2)
    r.render_batch(a.batch, a.batch); r.synchronize()
    ll = r.launch_list()
    r.set_counting(True); r.render_batch(2 * a.batch, a.batch); r.synchronize(); c, sh = r.trace_counters(); r.set_counting(False)
    print("handoff %d delay %d batch %d: %d launches, trace %.3f ms shade %.3f ms; rays traced closest %d shadow %d" % (handoff, a.delay, a.batch, len(ll), sum(m for b, m in ll if b != 3), sum(m for b, m in ll if b == 3), c.rays, sh.rays))
    print("   " + "  ".join("%s %.3f" % (names[b], m) for b, m in ll))
    r.close()
