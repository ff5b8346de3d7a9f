#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats rocpd database -> profiles/<tag>.md (per-kernel calls / total / average duration)
usage: tools/summarize_stats.py <dir with the .db> <tag> "<command line that was profiled>" """
import os, sqlite3, sys
src, tag, what = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fs = [os.path.join(r, x) for r, _, f in os.walk(src) for x in f if x.endswith(".db")]
cur = sqlite3.connect(fs[0]).cursor()
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
with open(os.path.join(root, "profiles", tag + ".md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats : `%s` (1x MI355X)\n\n" % what)
    f.write("Durations in microseconds. `trace_kernel<MODE, COUNTED>`: MODE 0 = closest hit (the RT boundary's rays; until round 4 also the primary rays), 3 = MIXED (closest-hit rays of bounce b+1 +\n"
            "any-hit shadow rays of bounce b fused with solve_occlusion), 2 = any-hit fused only, 1 = any-hit with written results, 4 = MIXED with the PSFPT resolve, 5 = MIXED with written any-hit results (BPT), 6 / 7 = closest hit over a renderer's path queue (primary / scattered rays: round 5's queue layout), 8 = any-hit over a renderer's shadow queue with written results; COUNTED=true rows are the instrumented re-run bench.py\n"
            "does after the timed region (same passes, counts node steps / triangles), not part of the timed region.\n\n")
    f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows:
        f.write("| `%s` | %d | %.1f | %.2f | %.2f |\n" % (r[0], r[1], r[2], r[3], r[4]))
print(open(os.path.join(root, "profiles", tag + ".md")).read()[:3000])
