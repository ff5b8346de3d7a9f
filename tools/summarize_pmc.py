#!/usr/bin/env python3
"""rocprofv3 PMC collections of one bench.py command line -> profiles/<tag>.json, the file bench.py's roofline object reads back for the
SAME configuration (matched by `config_key`, which bench.py prints in config.config_key).

  tools/summarize_pmc.py <dir with fetch/ write/ valu/ sub-directories of rocpd .db files> <bench json line file> <tag>

HBM bytes per traversal launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM: counters in KiB; gfx950's FETCH_SIZE tallies
128-B requests as 64 B), launch-weighted over the uninstrumented trace_kernel<MODE, false> launches (the COUNTED=true launches belong to
bench.py's instrumented re-run after the timed region).  VALU: SQ_INSTS_VALU wave-instructions x 4 issue cycles over 1024 SIMDs against the launch duration at the 2.4 GHz peak clock; lane
utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) (both calibrated on fully converged kernels)."""
import json, os, sqlite3, sys

src, line_file, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = json.loads([l for l in open(line_file).read().splitlines() if l.startswith("{")][-1])


def rows(sub):
    d = os.path.join(src, sub)
    fs = [os.path.join(r, x) for r, _, f in os.walk(d) for x in f if x.endswith(".db")] if os.path.isdir(d) else []
    if not fs:
        return []
    cur = sqlite3.connect(fs[0]).cursor()
    return list(cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"))


def is_timed_trace(kn):
    return "trace_kernel<" in kn and "false>" in kn

per = {}
for sub in ("fetch", "write", "valu"):
    for kn, cn, n, v, du in rows(sub):
        if "fpt::" in kn:
            per.setdefault(kn, {})[cn] = {"launches": n, "avg": v, "avg_duration_us": du / 1e3}
out = {"config_key": line["config"]["config_key"], "bench_line": {k: line[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step")},
       "source": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES} "
                 "(three separate passes) over the bench.py command line of this configuration",
       "correction": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", "kernels": {}}
tot_b = tot_n = 0.0
vi = va = vt = vd = vn = 0.0
for kn, c in per.items():
    f = c.get("FETCH_SIZE", {}).get("avg", 0.0) or 0.0; w = c.get("WRITE_SIZE", {}).get("avg", 0.0) or 0.0
    n = c.get("FETCH_SIZE", {}).get("launches", 0)
    k = {"hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "launches": n, "avg_duration_us_profiled": c.get("FETCH_SIZE", {}).get("avg_duration_us")}
    if "SQ_INSTS_VALU" in c:
        iv = c["SQ_INSTS_VALU"]["avg"]; du = c["SQ_INSTS_VALU"]["avg_duration_us"]
        k["valu_wave_instructions_per_launch"] = iv
        k["valu_busy_frac_at_2.4GHz"] = iv * 4.0 / 1024.0 / (du * 1e-6 * 2.4e9)
        if c.get("SQ_ACTIVE_INST_VALU", {}).get("avg"):
            k["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"]["avg"] / (c["SQ_ACTIVE_INST_VALU"]["avg"] * 64.0)
    out["kernels"][kn] = k
    if is_timed_trace(kn):
        tot_b += k["hbm_bytes_per_launch"] * n; tot_n += n
        if "SQ_INSTS_VALU" in c:
            m = c["SQ_INSTS_VALU"]["launches"]
            vi += c["SQ_INSTS_VALU"]["avg"] * m; vd += c["SQ_INSTS_VALU"]["avg_duration_us"] * m; vn += m
            va += (c.get("SQ_ACTIVE_INST_VALU", {}).get("avg") or 0.0) * m; vt += (c.get("SQ_THREAD_CYCLES_VALU", {}).get("avg") or 0.0) * m
if tot_n:
    out["hbm_bytes_per_launch"] = tot_b / tot_n
    out["kernel"] = "trace_kernel<MODE, false> (launch-weighted mean over %d launches)" % tot_n
if vn:
    out["valu"] = {"bound": "valu", "unit": "wave-instructions/s", "achieved": vi / (vd * 1e-6), "peak": 1024 * 2.4e9 / 4.0,
                   "frac": (vi / (vd * 1e-6)) / (1024 * 2.4e9 / 4.0), "lane_utilisation": (vt / (va * 64.0)) if va else None,
                   "note": "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 issue cycles per wave64 VALU instruction (SQ_ACTIVE_INST_VALU reads 1.00 quad-cycle per instruction); lane utilisation calibrated on fully converged kernels (= 1.00); the effective clock under load is lower (DVFS)"}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", tag + ".json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
