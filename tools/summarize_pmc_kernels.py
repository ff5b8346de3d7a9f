#!/usr/bin/env python3
"""rocprofv3 PMC collections (fetch/ write/ valu/ sub-directories of rocpd .db files) of ANY command -> profiles/<tag>.json: per kernel, HBM bytes per
launch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, MI355X_MICROARCH.md), the achieved HBM rate over the profiled launch duration, VALU busy fraction
(SQ_INSTS_VALU x 4 issue cycles over 1024 SIMDs at 2.4 GHz) and lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)).
usage: tools/summarize_pmc_kernels.py <dir> <tag> "<command that was profiled>" [kernel-name filter]"""
import json, os, sqlite3, sys
src, tag, what = sys.argv[1], sys.argv[2], sys.argv[3]
flt = sys.argv[4] if len(sys.argv) > 4 else "fpt::"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(sub):
    d = os.path.join(src, sub)
    fs = [os.path.join(r, x) for r, _, f in os.walk(d) for x in f if x.endswith(".db")] if os.path.isdir(d) else []
    if not fs:
        return []
    cur = sqlite3.connect(fs[0]).cursor()
    return list(cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"))

per = {}
for sub in ("fetch", "write", "valu"):
    for kn, cn, n, v, du in rows(sub):
        if flt in kn:
            per.setdefault(kn, {})[cn] = {"launches": n, "avg": v, "avg_duration_us": du / 1e3}
out = {"command": what, "source": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES} (three separate passes)",
       "correction": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", "kernels": {}}
for kn, c in sorted(per.items()):
    f = c.get("FETCH_SIZE", {}).get("avg", 0.0) or 0.0; w = c.get("WRITE_SIZE", {}).get("avg", 0.0) or 0.0
    du = c.get("FETCH_SIZE", {}).get("avg_duration_us")
    k = {"launches": c.get("FETCH_SIZE", {}).get("launches", 0), "avg_duration_us_profiled": du, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    if du:
        k["hbm_gbs"] = k["hbm_bytes_per_launch"] / (du * 1e-6) / 1e9; k["hbm_frac_of_8TBs"] = k["hbm_gbs"] / 8000.0
    if "SQ_INSTS_VALU" in c:
        iv = c["SQ_INSTS_VALU"]["avg"]; dv = c["SQ_INSTS_VALU"]["avg_duration_us"]
        k["valu_wave_instructions_per_launch"] = iv
        k["valu_busy_frac_at_2.4GHz"] = iv * 4.0 / 1024.0 / (dv * 1e-6 * 2.4e9)
        if c.get("SQ_ACTIVE_INST_VALU", {}).get("avg"):
            k["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"]["avg"] / (c["SQ_ACTIVE_INST_VALU"]["avg"] * 64.0)
    out["kernels"][kn] = k
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", tag + ".json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
