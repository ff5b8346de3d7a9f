#!/usr/bin/env python3
"""A/B of experimental library builds (tools/build_variant.sh) on the GPU box: runs bench.py once per variant (FPT_LIB_PATH) and prints one compact line each.
    python tools/ab_variants.py [--args "--steps 20 --warmup 5"] [--repeat 2] base vote16 drop8 ...        ('base' = the product library)"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--args", default="--steps 20 --warmup 5")
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
for rep in range(a.repeat):
    for v in a.variants:
        env = dict(os.environ)
        if v != "base":
            env["FPT_LIB_PATH"] = os.path.join(ROOT, "fermat_amd", "variants", "libfermat_%s.so" % v)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-cpu-baseline"] + a.args.split(), env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("%-12s FAILED rc %d: %s" % (v, p.returncode, p.stderr[-300:])); continue
        o = json.loads(line[-1]); k = o["kernel_ms_per_step"]
        print("%-12s [%s] %8.1f Msample/s  %.4f ms/step  trace %.4f shadow-only %.4f shade %.4f  nodes/ray %.2f tris/ray %.2f  in flight %d" %
              (v, a.args, o["value"], o["ms_per_step"], k["trace_primary+mixed"], k["trace_shadow_only"], k["shade"], o["roofline"]["nodes_per_ray"], o["roofline"]["tris_per_ray"], o["config"]["passes_in_flight"]), flush=True)
