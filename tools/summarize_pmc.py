#!/usr/bin/env python3
"""rocprofv3 PMC collections of one bench.py command line -> profiles/<tag>.json, the file bench.py's roofline object reads back for the
SAME configuration (matched by `config_key`, which bench.py prints in config.config_key) and the SAME kernel sources (`source_hash`).

  tools/summarize_pmc.py <dir with fetch/ write/ valu/ sub-directories of rocpd .db files> <bench json line file> <tag>

Round 5 (VERDICT r4 task 2): bytes and durations come from the SAME launches.  The timed region's traversal launches are the last N uninstrumented
trace_kernel<MODE, false> launches of the run, N = the bench line's roofline.launches (warm-up launches, which have another batch size, come before them; the COUNTED
re-run after the timed region uses trace_kernel<MODE, true>); the three passes run the same command, so launch i of one pass is launch i of another.
  hbm bytes of launch i  = (FETCH_FACTOR x FETCH_SIZE_i + WRITE_SIZE_i) x 1024
  counter_gbs_profiled   = sum of those bytes / sum of the SAME launches' durations (as rocprofv3 timed them in the FETCH_SIZE pass)
  hbm_bytes_per_launch   = sum / N: what bench.py divides by ITS live average launch duration of the same N launches of an unprofiled run
FETCH_FACTOR: MI355X_MICROARCH.md (HBM) -- gfx950's FETCH_SIZE tallies a 128-byte request as 64 bytes on coalesced 16-B/lane streams: x 2; calibrated on the traversal's own
pattern (every lane at another 80-byte record: profiles/r05_fetch_size_calibration.json, tools/calibrate_fetch_size.py) it is x 2.008 (x 2.006 on whole lines: a 128-byte line is tallied as 64 bytes whatever part of it is read), so 2 it stays.
VALU: SQ_INSTS_VALU wave-instructions x 4 issue cycles over 1024 SIMDs against the launch duration at the 2.4 GHz peak clock; lane utilisation =
SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) (both calibrated on fully converged kernels)."""
import json, os, sqlite3, sys

src, line_file, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
line = json.loads([l for l in open(line_file).read().splitlines() if l.startswith("{")][-1])
FETCH_FACTOR = 2.0


def launches(sub):
    """{counter: [(kernel_name, value, duration_ns) in dispatch order]} of one pass"""
    d = os.path.join(src, sub)
    fs = [os.path.join(r, x) for r, _, f in os.walk(d) for x in f if x.endswith(".db")] if os.path.isdir(d) else []
    out = {}
    if fs:
        cur = sqlite3.connect(fs[0]).cursor()
        for kn, cn, v, du in cur.execute("select kernel_name, counter_name, value, duration from counters_collection order by dispatch_id"):
            if "fpt::" in kn:
                out.setdefault(cn, []).append((kn, v, du))
    return out


def is_timed_trace(kn):
    return "trace_kernel<" in kn and "false>" in kn


fetch, write, valu = launches("fetch"), launches("write"), launches("valu")
N = int(line["roofline"]["launches"])
out = {"config_key": line["config"]["config_key"], "bench_line": {k: line[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step")},
       "source": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES} "
                 "(three separate passes) over the bench.py command line of this configuration",
       "correction": "HBM bytes = (%g x FETCH_SIZE + WRITE_SIZE) x 1024 (factor calibrated on the traversal's access pattern: profiles/r05_fetch_size_calibration.json)" % FETCH_FACTOR,
       "kernels": {}}
try:
    from fermat_amd.api import kernel_source_hash
    out["source_hash"] = kernel_source_hash()
except Exception as e:          # noqa: BLE001
    out["source_hash"] = None; out["source_hash_error"] = str(e)

# per kernel name, over ALL its launches of the run (warm-up included): the per-kernel table of DESIGN 7
per = {}
for cn, rows in list(fetch.items()) + list(write.items()) + list(valu.items()):
    for kn, v, du in rows:
        e = per.setdefault(kn, {}).setdefault(cn, [0, 0.0, 0.0])
        e[0] += 1; e[1] += v; e[2] += du
for kn, c in per.items():
    f = c.get("FETCH_SIZE", [0, 0.0, 0.0]); w = c.get("WRITE_SIZE", [0, 0.0, 0.0])
    k = {"launches": f[0], "hbm_bytes_per_launch": ((FETCH_FACTOR * f[1] + (w[1] * f[0] / w[0] if w[0] else 0.0)) * 1024.0 / f[0]) if f[0] else None,
         "avg_duration_us_profiled": (f[2] / f[0] / 1e3) if f[0] else None}
    if k["hbm_bytes_per_launch"] and f[2]:
        k["hbm_gbs_profiled"] = k["hbm_bytes_per_launch"] * f[0] / (f[2] * 1e-9) / 1e9
    if "SQ_INSTS_VALU" in c and c["SQ_INSTS_VALU"][2]:
        n, iv, du = c["SQ_INSTS_VALU"]
        k["valu_wave_instructions_per_launch"] = iv / n
        k["valu_busy_frac_at_2.4GHz"] = iv * 4.0 / 1024.0 / (du * 1e-9 * 2.4e9)
        if c.get("SQ_ACTIVE_INST_VALU", [0, 0.0])[1]:
            k["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"][1] / (c["SQ_ACTIVE_INST_VALU"][1] * 64.0)
    out["kernels"][kn] = k

# the timed region's traversal launches: the last N uninstrumented ones, the same launches in every pass
tf = [r for r in fetch.get("FETCH_SIZE", []) if is_timed_trace(r[0])][-N:]
tw = [r for r in write.get("WRITE_SIZE", []) if is_timed_trace(r[0])][-N:]
if len(tf) == N and len(tw) == N and N:
    assert [r[0] for r in tf] == [r[0] for r in tw], "the FETCH_SIZE and WRITE_SIZE passes did not run the same launch sequence"
    b = [(FETCH_FACTOR * f[1] + w[1]) * 1024.0 for f, w in zip(tf, tw)]
    dur = [f[2] * 1e-9 for f in tf]
    out["timed_launches"] = N
    out["hbm_bytes_total"] = sum(b); out["duration_total_ms_profiled"] = sum(dur) * 1e3
    out["hbm_bytes_per_launch"] = sum(b) / N
    out["counter_gbs_profiled"] = sum(b) / sum(dur) / 1e9
    out["kernel"] = "trace_kernel<MODE, false>: the %d launches of the timed region (bytes and durations of the same launches)" % N
    out["per_launch"] = [{"kernel": f[0].split("<")[1].split(">")[0] if "<" in f[0] else f[0], "hbm_bytes": x, "ms_profiled": d * 1e3} for f, x, d in zip(tf, b, dur)]
else:
    out["error"] = "expected %d timed traversal launches, found %d (FETCH_SIZE pass) / %d (WRITE_SIZE pass)" % (N, len(tf), len(tw))
vs = {cn: [r for r in rows if is_timed_trace(r[0])][-N:] for cn, rows in valu.items()}
if vs.get("SQ_INSTS_VALU") and len(vs["SQ_INSTS_VALU"]) == N:
    vi = sum(r[1] for r in vs["SQ_INSTS_VALU"]); vd = sum(r[2] for r in vs["SQ_INSTS_VALU"]) * 1e-9
    va = sum(r[1] for r in vs.get("SQ_ACTIVE_INST_VALU", [])); vt = sum(r[1] for r in vs.get("SQ_THREAD_CYCLES_VALU", []))
    # the VALU roof of the kernel's OWN instruction mix (VERDICT r5 task 1a): gfx950 issues two classes of wave64 VALU instruction (tools/micro/issue_model2.hip,
    # profiles/r06_micro_issue_model2.txt: ~2.7 and ~4.4 cycles per SIMD); tools/isa_classes.py counts the classes in the node step and the triangle step of the kernel
    # these counters were collected on and weights them with the line's node steps / triangle tests per ray -> issue cycles per counted wave-instruction
    import subprocess
    rf = line["roofline"]
    try:
        census = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "tools", "isa_classes.py"), "--build", "--json", "--nodes-per-ray", "%.4f" % rf["nodes_per_ray"],
                                                     "--tris-per-ray", "%.4f" % rf["tris_per_ray"]], text=True))
        cyc = census["issue_cycles_per_wave_instruction"]
    except Exception as e:          # noqa: BLE001
        census = {"error": str(e)}; cyc = 4.0
    lane = (vt / (va * 64.0)) if va else None
    peak = 1024 * 2.4e9 / cyc
    out["valu"] = {"bound": "valu", "unit": "wave-instructions/s", "achieved": vi / vd, "peak": peak,
                   "frac": (vi / vd) / peak, "lane_utilisation": lane, "useful_frac": ((vi / vd) / peak * lane) if lane else None,
                   "issue_cycles_per_wave_instruction": cyc, "frac_at_4_cycles": (vi / vd) / (1024 * 2.4e9 / 4.0),
                   "wave_instructions_per_launch": vi / N, "mix": census,
                   "note": "the timed region's launches only.  peak = 256 CUs x 4 SIMDs x 2.4 GHz / the issue time of the kernel's own mix: (fast-class instructions x 2.7 + slow-class x 4.4 "
                           "cycles) / instructions over a ray's node steps and triangle tests (tools/isa_classes.py on the kernel's ISA; classes measured by tools/micro/issue_model2.hip) -- "
                           "until round 5 the peak assumed 4 cycles per instruction and frac read 1.06 (kept as frac_at_4_cycles).  useful_frac = frac x lane utilisation: the share of the "
                           "chip's VALU lane-slots that do a ray's work.  The class cycles are measured at the nominal 2.4 GHz; the effective clock under load is lower (DVFS)"}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", tag + ".json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("kernels", "per_launch")}, indent=1))
