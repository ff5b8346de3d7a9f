#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (gpurun_out/prof/{stats,pmc_fetch,pmc_write}/*.db) into the committed summaries under profiles/.

  profiles/<tag>_kernel_stats.md      per-kernel calls / total / average duration (rocprofv3 --kernel-trace --stats)
  profiles/<tag>_pmc_traversal.json   HBM traffic per launch of the traversal kernels from FETCH_SIZE / WRITE_SIZE, collected in
                                      separate --pmc passes and corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
                                      counters are KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> doubled.
"""
import json, os, sqlite3, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "prof")
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def db(path):
    f = [x for x in os.listdir(path) if x.endswith(".db")][0]
    return sqlite3.connect(os.path.join(path, f)).cursor()


cur = db(os.path.join(src, "stats"))
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(os.path.join(out, tag + "_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats : `python bench.py --warmup 64 --no-cpu-baseline` (1x MI355X, 256 steps + 64 warm-up steps, 64 passes in flight: every batch is 64 passes)\n\n")
    f.write("Durations in microseconds. `trace_kernel<MODE, COUNTED>`: MODE 0 = closest hit (primary rays), 3 = MIXED (closest-hit rays of bounce b+1 +\n"
            "any-hit shadow rays of bounce b fused with solve_occlusion), 2 = any-hit fused only; COUNTED=true rows are the instrumented re-run bench.py\n"
            "does after the timed region (same passes, counts nodes/triangles), not part of the timed region.\n\n")
    f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows:
        f.write("| `%s` | %d | %.1f | %.2f | %.2f |\n" % (r[0], r[1], r[2], r[3], r[4]))

res = {}
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cur = db(os.path.join(src, name))
    q = "select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=? group by kernel_name"
    for kn, n, v, d in cur.execute(q, (key,)):
        res.setdefault(kn, {})[key] = {"launches": n, "avg_kib": v, "avg_duration_ns": d}
summary = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 64 --warmup 0 (one 64-pass batch, as in the timed runs)",
           "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024  (MI355X_MICROARCH.md: counters in KiB; gfx950 FETCH_SIZE tallies 128-B requests as 64 B)", "kernels": {}}
for kn, v in res.items():
    if "fpt::" not in kn:
        continue
    fetch = v.get("FETCH_SIZE", {}).get("avg_kib", 0.0) or 0.0
    write = v.get("WRITE_SIZE", {}).get("avg_kib", 0.0) or 0.0
    summary["kernels"][kn] = {"fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
                              "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                              "launches_sampled": v.get("FETCH_SIZE", {}).get("launches", 0),
                              "avg_duration_us_profiled": (v.get("FETCH_SIZE", {}).get("avg_duration_ns", 0) or 0) / 1e3}
# the timed traversal launches: MODE 0 (primary) and MODE 3 (mixed), uninstrumented; launch-weighted mean
ks = [x for x in summary["kernels"] if ("trace_kernel<0, false>" in x or "trace_kernel<3, false>" in x or "trace_kernel<2, false>" in x)]
if ks:
    n = sum(summary["kernels"][x]["launches_sampled"] for x in ks)
    summary["hbm_bytes_per_launch"] = sum(summary["kernels"][x]["hbm_bytes_per_launch"] * summary["kernels"][x]["launches_sampled"] for x in ks) / max(1, n)
    summary["kernel"] = "trace_kernel<MODE 0|3, false> (launch-weighted mean over %d launches)" % n
json.dump(summary, open(os.path.join(out, tag + "_pmc_traversal.json"), "w"), indent=1)
print(open(os.path.join(out, tag + "_kernel_stats.md")).read())
print(json.dumps(summary, indent=1)[:3000])


# widened rows: HBM traffic per traversal launch of the BPT / PSFPT bench lines (same correction), when their PMC passes were collected
for kind, modes in (("bpt", ("trace_kernel<0, false>", "trace_kernel<1, false>")), ("psfpt", ("trace_kernel<0, false>", "trace_kernel<4, false>", "trace_kernel<1, false>"))):
    if not (os.path.isdir(os.path.join(src, "pmc_fetch_" + kind)) and os.path.isdir(os.path.join(src, "pmc_write_" + kind))):
        continue
    res = {}
    for name, key in (("pmc_fetch_" + kind, "FETCH_SIZE"), ("pmc_write_" + kind, "WRITE_SIZE")):
        cur = db(os.path.join(src, name))
        for kn, n, v, d in cur.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=? group by kernel_name", (key,)):
            res.setdefault(kn, {})[key] = {"launches": n, "avg_kib": v, "avg_duration_ns": d}
    ks = [x for x in res if any(m in x for m in modes)]
    n = sum(res[x].get("FETCH_SIZE", {}).get("launches", 0) for x in ks)
    tot = sum((2.0 * (res[x].get("FETCH_SIZE", {}).get("avg_kib", 0.0) or 0.0) + (res[x].get("WRITE_SIZE", {}).get("avg_kib", 0.0) or 0.0)) * 1024.0 * res[x].get("FETCH_SIZE", {}).get("launches", 0) for x in ks)
    w = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --renderer %s --steps 16 --warmup 0" % kind,
         "correction": summary["correction"], "kernel": "traversal launches of the %s pass (launch-weighted mean over %d launches)" % (kind.upper(), n),
         "hbm_bytes_per_launch": tot / max(1, n)}
    json.dump(w, open(os.path.join(out, "%s_pmc_traversal_%s.json" % (tag, kind)), "w"), indent=1)
    print(kind, w["hbm_bytes_per_launch"])
